"""Static mesh plan: everything about the (fixed) topology that the reference recomputes in every
forward with device->host syncs is computed here once and cached.

  * int32 copies of the int64 index tensors of the `Graph` layout,
  * CSR-by-destination permutation + segment offsets for every aggregation
    (replaces the index handling inside `torch_geometric.utils.scatter`, nn/blocks.py:183,231,330,378),
  * the topology part of `pool_edge` (nn/blocks.py:63-67): coarse edge_index, fine->coarse edge
    map as a segmented permutation.

Plans are keyed on tensor identity (data_ptr, shape, version counter, device) so the block API
keeps the reference's signatures (`GNBlock.forward(v, e, edge_index)` gets only the tensor).
Building a plan copies the index tensor to the host once (the C-ABI plan builders are host code)
and uploads int32 results; a rollout step itself never syncs.
"""
from __future__ import annotations

import collections
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib


@dataclass
class CsrPlan:
    perm: Optional[torch.Tensor]   # int32 [n] on device; None when the input is already grouped in order
    off: torch.Tensor              # int32 [n_seg+1] on device
    n: int
    n_seg: int
    max_deg: int
    uniform_deg: int = 0           # k when EVERY segment has exactly k rows (kNN meshes), else 0
    _tiles: object = False         # cache of tiles(): False = not built yet, None = not tileable

    def tiles(self, max_rows: int = 32):
        """(tile_rows int32 [T+1], tile_seg int32 [T+1], T) on the device: tiles of whole segments of at most `max_rows`
        rows (g4c_plan_tiles), for the edge-MLP launch that also aggregates (ops.mlp_forward(agg=...)); None when the rows
        are not in segment order or a segment is longer than a tile."""
        if self._tiles is False:
            self._tiles = None
            if self.perm is None and self.n > 0 and 0 < self.max_deg <= max_rows:
                lib = _lib.load()
                off = np.ascontiguousarray(_host_i64(self.off).astype(np.int32))
                cap = self.n_seg + 2
                rows, seg = np.empty(cap, np.int32), np.empty(cap, np.int32)
                nt = int(lib.g4c_plan_tiles(off.ctypes.data, self.n_seg, max_rows, rows.ctypes.data, seg.ctypes.data, cap))
                if nt > 0:
                    dev = self.off.device
                    self._tiles = (_upload(rows[: nt + 1].copy(), dev), _upload(seg[: nt + 1].copy(), dev), nt)
        return self._tiles


@dataclass
class EdgePlan:
    """Plan of one `edge_index` / `angle_index` ([2, E] int64, row = sender, col = receiver)."""
    row: torch.Tensor              # int32 [E]
    col: torch.Tensor              # int32 [E]
    n_edges: int
    csr: dict                      # n_targets -> CsrPlan grouped by col


@dataclass
class PoolEdgePlan:
    edge_index: torch.Tensor       # int64 [2, E_l], ordered like torch_geometric coalesce output
    csr: CsrPlan                   # fine edges grouped by coarse edge
    n_coarse: int


def _host_i64(t: torch.Tensor) -> np.ndarray:
    """Host int64 copy of an index tensor.  A device tensor whose host image is known (`remember_host`: the graph was moved
    with `Graph.to`, or the tensor was produced by a host-side plan builder) is NOT read back: a device->host copy waits for
    every launch queued so far, which in a training loop over fresh batches serialises the host and the GPU once per plan."""
    if t.device.type != "cpu":
        hit = _host_image(t)
        if hit is not None:
            return np.ascontiguousarray(hit, dtype=np.int64)
    return np.ascontiguousarray(t.detach().to("cpu", torch.int64).numpy())


def _host_image(t: torch.Tensor) -> Optional[np.ndarray]:
    """The registered host image of a device tensor, unless its source CPU tensor was modified in place since."""
    hit = _host_copies.get(_Cache.key(t))
    if hit is None:
        return None
    arr, src, version = hit
    if src is not None and src._version != version:
        return None
    return arr


def _upload(arr: np.ndarray, device) -> torch.Tensor:
    """Host array -> device tensor.  (Through pageable memory: staging through `pin_memory()` was measured 2.5x slower per
    training iteration over fresh batches — every new array size costs a pinned allocation of several ms.)"""
    return torch.from_numpy(arr).to(device)


def remember_host(dev_tensor: torch.Tensor, host) -> None:
    """Register the host image (numpy array or CPU tensor, not to be modified afterwards) of a device index tensor."""
    if dev_tensor.device.type == "cpu" or dev_tensor.dtype.is_floating_point or not _REMEMBER_HOST:
        return
    src = host if torch.is_tensor(host) else None
    arr = host.detach().numpy() if src is not None else np.asarray(host)
    _host_copies.put(_Cache.key(dev_tensor), (dev_tensor,), (arr, src, src._version if src is not None else 0))


def _uniform(deg: np.ndarray) -> int:
    """k if every segment has exactly k rows (1 <= k <= 32), else 0."""
    if deg.size == 0:
        return 0
    k = int(deg.max())
    return k if 1 <= k <= 32 and int(deg.min()) == k else 0


def build_csr(keys: torch.Tensor, n_seg: int, device: torch.device, drop_last_segment: bool = False) -> CsrPlan:
    """Group positions by key (stable). `drop_last_segment`: keys == n_seg-1 are a trash bin."""
    lib = _lib.load()
    k = _host_i64(keys)
    n = int(k.shape[0])
    perm = np.empty(n, dtype=np.int32)
    off = np.empty(n_seg + 1, dtype=np.int32)
    _lib.check(lib.g4c_plan_csr(k.ctypes.data, n, n_seg, perm.ctypes.data, off.ctypes.data))
    if drop_last_segment:
        n_seg -= 1
        off = off[: n_seg + 1]
        perm = perm[: int(off[-1])]
    n_kept = int(perm.shape[0])
    identity = n_kept == n and bool(np.array_equal(perm, np.arange(n, dtype=np.int32)))
    deg = np.diff(off)
    off_h = off.copy()
    out = CsrPlan(
        perm=None if identity else _upload(perm.copy(), device),
        off=_upload(off_h, device),
        n=n_kept, n_seg=n_seg, max_deg=int(deg.max()) if deg.size else 0, uniform_deg=_uniform(deg))
    remember_host(out.off, off_h)
    return out


def _nbytes(obj, depth: int = 0) -> int:
    """Bytes pinned by a cache entry (device index tensors, host images), walked through tuples / dataclasses."""
    if torch.is_tensor(obj):
        return obj.numel() * obj.element_size()
    if isinstance(obj, np.ndarray):
        return int(obj.nbytes)
    if depth > 3 or obj is None or isinstance(obj, (int, float, str, bool)):
        return 0
    if isinstance(obj, (tuple, list)):
        return sum(_nbytes(o, depth + 1) for o in obj)
    if isinstance(obj, dict):
        return sum(_nbytes(o, depth + 1) for o in obj.values())
    if hasattr(obj, "__dict__"):
        return sum(_nbytes(o, depth + 1) for o in vars(obj).values())
    return 0


class _Cache:
    """LRU over plans, bounded both by entry count and by the bytes its entries pin (key tensors are kept alive so their
    data_ptr cannot be recycled under a cached plan; over fresh training batches nothing is ever reused, so an unbounded
    cache would just retain the last N batches' index data in HBM).  G4C_PLAN_CACHE_MB: per-cache byte bound."""

    def __init__(self, capacity: int = 256, max_bytes: Optional[int] = None):
        self.capacity = capacity
        self.max_bytes = int(__import__("os").environ.get("G4C_PLAN_CACHE_MB", "1024")) << 20 if max_bytes is None else max_bytes
        self.bytes = 0
        self.data = collections.OrderedDict()

    @staticmethod
    def key(*tensors: torch.Tensor):
        return tuple((t.data_ptr(), tuple(t.shape), t._version, str(t.device), t.dtype) for t in tensors)

    def get(self, key):
        hit = self.data.get(key)
        if hit is not None:
            self.data.move_to_end(key)
            return hit[1]
        return None

    def put(self, key, tensors, value):
        old = self.data.pop(key, None)
        if old is not None:
            self.bytes -= old[2]
        size = _nbytes(tensors) + _nbytes(value)
        self.data[key] = (tensors, value, size)
        self.bytes += size
        # the newest entry always stays (a single mesh larger than the bound must still be planned once per rollout)
        while len(self.data) > 1 and (len(self.data) > self.capacity or self.bytes > self.max_bytes):
            _, (_, _, freed) = self.data.popitem(last=False)
            self.bytes -= freed
        return value

    def clear(self) -> None:
        self.data.clear()
        self.bytes = 0


_edge_plans = _Cache()
_host_copies = _Cache(capacity=256)
_REMEMBER_HOST = True      # (False: always read index tensors back; A/B only — scripts/bench_fit_batches.py flips it)
_pool_plans = _Cache()
_index_plans = _Cache()
_cluster_plans = _Cache()


def snapshot() -> list:
    """References to everything the caches hold right now.  A captured hipGraph bakes the plans' device pointers in, so its
    owner (nn.model.Rollout, partition.DistributedRollout) keeps this list for as long as it may replay: eviction from the
    caches then cannot free memory the graph still reads."""
    return [entry for c in (_edge_plans, _pool_plans, _index_plans, _cluster_plans) for entry in c.data.values()]


def clear_caches() -> None:
    for c in (_edge_plans, _pool_plans, _index_plans, _cluster_plans, _host_copies):
        c.clear()


def edge_plan(edge_index: torch.Tensor) -> EdgePlan:
    key = _Cache.key(edge_index)
    plan = _edge_plans.get(key)
    if plan is None:
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
        dev = _lib.require_hip(edge_index)
        plan = EdgePlan(row=edge_index[0].to(torch.int32).contiguous(), col=edge_index[1].to(torch.int32).contiguous(),
                        n_edges=int(edge_index.size(1)), csr={})
        host = _host_image(edge_index)
        if host is not None:
            remember_host(plan.row, np.ascontiguousarray(host[0]))
            remember_host(plan.col, np.ascontiguousarray(host[1]))
        _edge_plans.put(key, (edge_index,), plan)
    return plan


def edge_csr(edge_index: torch.Tensor, n_targets: int) -> Tuple[EdgePlan, CsrPlan]:
    plan = edge_plan(edge_index)
    csr = plan.csr.get(n_targets)
    if csr is None:
        csr = build_csr(plan.col, n_targets, edge_index.device)
        plan.csr[n_targets] = csr
    return plan, csr


def grouped_by_target(edge_index: torch.Tensor, n_targets: int) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """(edge_index with its columns stably grouped by target, the int64 permutation that does it) — cached like the plans, so that
    the grouped tensor has a stable identity of its own (its EdgePlan / CsrPlan are built once).  (edge_index, None) when the columns
    are grouped already.  For launches that want their rows in receiver order (the row-split kernels' fused aggregation) where the
    caller can reorder the row tensor once: REMuS-GNN's static inter-level angle latents (DownEdgeMP)."""
    _, csr = edge_csr(edge_index, n_targets)
    if csr.perm is None:
        return edge_index, None
    key = _Cache.key(edge_index) + (("grouped", n_targets),)
    out = _index_plans.get(key)
    if out is None:
        perm = csr.perm.to(torch.int64)
        out = _index_plans.put(key, (edge_index,), (edge_index.index_select(1, perm).contiguous(), perm))
    return out


def index32(index: torch.Tensor) -> torch.Tensor:
    """Cached int32 copy of an int64 gather index (idx{h}_to_idx{l}, x_idx, col, ...)."""
    if index.dtype == torch.int32:
        return index
    key = _Cache.key(index)
    out = _index_plans.get(key)
    if out is None:
        _lib.require_hip(index)
        out = _index_plans.put(key, (index,), index.to(torch.int32).contiguous())
        host = _host_image(index)
        if host is not None:
            remember_host(out, host.copy())
    return out


def gather_csr(index: torch.Tensor, n_rows: int) -> CsrPlan:
    """Cached CSR plan grouping the positions of a gather index by the row they read (adjoint of `x[index]`: the
    training path sums the gradient rows of each group in order, autograd.py)."""
    key = _Cache.key(index) + ("gather_csr", n_rows)
    out = _index_plans.get(key)
    if out is None:
        out = _index_plans.put(key, (index,), build_csr(index, n_rows, index.device))
    return out


def mask_index32(mask: torch.Tensor) -> torch.Tensor:
    """Cached int32 positions of the True entries of a boolean node mask (coarse_mask{l})."""
    key = _Cache.key(mask) + ("nonzero",)
    out = _index_plans.get(key)
    if out is None:
        _lib.require_hip(mask)
        out = _index_plans.put(key, (mask,), mask.nonzero().reshape(-1).to(torch.int32).contiguous())
    return out


def restricted_level(mask_l: torch.Tensor, edge_index_l: torch.Tensor, mask_prev: Optional[torch.Tensor]):
    """Static part of a gMuS-GNN down-sampling (nn/mugs_gnn.py:101-104,251-255 + `restriction`, nn/blocks.py:9-32):
    `mask_l` / `mask_prev` are the boolean masks of the new / current level over LEVEL-1 node ids, `edge_index_l` the new
    level's edges in level-1 ids.  Returns (keep32: rows of the current level that survive, edge_index in compact ids)."""
    key = _Cache.key(mask_l, edge_index_l) + (None if mask_prev is None else _Cache.key(mask_prev),)
    out = _index_plans.get(key)
    if out is None:
        _lib.require_hip(mask_l, edge_index_l)
        n1 = int(mask_l.size(0))
        lut = torch.full((n1,), -1, dtype=torch.long, device=mask_l.device)
        lut[mask_l] = torch.arange(int(mask_l.sum()), dtype=torch.long, device=mask_l.device)
        keep = mask_l if mask_prev is None else mask_l[mask_prev]
        keep32 = keep.nonzero().reshape(-1).to(torch.int32).contiguous()
        ei = lut[edge_index_l].contiguous()
        if bool((ei < 0).any()):
            raise ValueError("edge_index of a coarse level references nodes outside its coarse_mask")
        held = (mask_l, edge_index_l) + (() if mask_prev is None else (mask_prev,))
        out = _index_plans.put(key, held, (keep32, ei))
    return out


def segments_of_sorted(index: torch.Tensor, n_seg: Optional[int] = None) -> CsrPlan:
    """CSR plan for an index vector (knn_interpolate's y_idx; any `scatter` index)."""
    key = _Cache.key(index) + (n_seg,)
    plan = _index_plans.get(key)
    if plan is None:
        if n_seg is None:
            n_seg = int(index.max()) + 1 if index.numel() else 0
        plan = _index_plans.put(key, (index,), build_csr(index, n_seg, index.device))
    return plan


def cluster_plan(cluster: torch.Tensor, mask: torch.Tensor) -> CsrPlan:
    """Plan for `scatter(x, cluster, reduce='mean')[mask]` (DownMP, nn/blocks.py:231): output row j
    averages the rows i with cluster[i] == mask[j]; cluster ids absent from `mask` are dropped."""
    key = _Cache.key(cluster, mask)
    plan = _cluster_plans.get(key)
    if plan is None:
        dev = _lib.require_hip(cluster, mask)
        c, m = _host_i64(cluster), _host_i64(mask)
        if m.dtype == np.bool_ or mask.dtype == torch.bool:
            raise NotImplementedError("boolean mask_l is not part of the GridClustering layout (int64 ids expected)")
        n_out = int(m.shape[0])
        if np.unique(m).shape[0] != n_out:
            raise NotImplementedError("mask_l with repeated cluster ids is not supported")
        sorter = np.argsort(m, kind="stable")
        ms = m[sorter]
        pos = np.searchsorted(ms, c)
        pos_c = np.minimum(pos, max(n_out - 1, 0))
        valid = (ms[pos_c] == c) if n_out else np.zeros_like(c, dtype=bool)
        keys = np.where(valid, sorter[pos_c] if n_out else 0, n_out).astype(np.int64)
        plan = build_csr(torch.from_numpy(keys), n_out + 1, dev, drop_last_segment=True)
        _cluster_plans.put(key, (cluster, mask), plan)
    return plan


def pool_edge_plan(idx_hr_to_lr: torch.Tensor, edge_index: torch.Tensor, target_major: bool = False) -> PoolEdgePlan:
    """Static part of `pool_edge` (nn/blocks.py:51-68).  `target_major=False`: coarse edges in torch_geometric `coalesce`
    order (sorted by (row, col)) — the public `pool_edge`.  `target_major=True` (the models' internal use): the same
    edges grouped by TARGET (stable, i.e. by (col, row)): the reference never exposes the coarse edge order
    (SURVEY.md appendix A.2), and in this order the coarse MP layers need no permutation and can aggregate on load."""
    key = _Cache.key(idx_hr_to_lr, edge_index) + (bool(target_major),)
    plan = _pool_plans.get(key)
    if plan is None:
        lib = _lib.load()
        dev = _lib.require_hip(idx_hr_to_lr, edge_index)
        idx, ei = _host_i64(idx_hr_to_lr), _host_i64(edge_index)
        n_hr, n_edges = int(idx.shape[0]), int(ei.shape[1])
        coarse = np.empty((2, max(n_edges, 1)), dtype=np.int64)
        perm = np.empty(max(n_edges, 1), dtype=np.int32)
        off = np.empty(n_edges + 1, dtype=np.int32)
        kept = C.c_int64(0)
        n_coarse = lib.g4c_plan_pool_edge_ordered(idx.ctypes.data, n_hr, ei.ctypes.data, n_edges, 1 if target_major else 0,
                                                  coarse.ctypes.data, perm.ctypes.data, off.ctypes.data, C.byref(kept))
        if n_coarse < 0:
            _lib.check(int(n_coarse))
        n_coarse, n_kept = int(n_coarse), int(kept.value)
        # the builder writes the two rows back to back at stride n_coarse
        flat = coarse.reshape(-1)[: 2 * n_coarse].reshape(2, n_coarse) if n_coarse else np.empty((2, 0), dtype=np.int64)
        off = off[: n_coarse + 1].copy()
        deg = np.diff(off)
        csr = CsrPlan(perm=_upload(perm[:n_kept].copy(), dev), off=_upload(off, dev),
                      n=n_kept, n_seg=n_coarse, max_deg=int(deg.max()) if deg.size else 0, uniform_deg=_uniform(deg))
        flat_h = flat.copy()
        plan = PoolEdgePlan(edge_index=_upload(flat_h, dev), csr=csr, n_coarse=n_coarse)
        remember_host(plan.edge_index, flat_h)
        remember_host(csr.off, off)
        _pool_plans.put(key, (idx_hr_to_lr, edge_index), plan)
    return plan

"""Node-partitioned multi-GPU execution of the MuS-GNN hot path (new functionality: the reference is
single-device).

One process per GPU.  The fixed mesh is cut into `world` spatially compact parts:

  * ownership is decided on the COARSEST level (recursive coordinate bisection over the coarsest nodes, balanced by
    the number of level-1 nodes underneath; `assign_owners`) and inherited downwards through `idx{l-1}_to_idx{l}`, so
    every cluster of every level is wholly owned by one rank: DownMP node pooling, `pool_edge`
    (a coarse edge I->J only collects fine edges whose target lies in J) and UpMP are rank-local;
  * an edge is owned by the rank of its TARGET node, so aggregation (`scatter` onto `col`) and the
    CSR-by-destination layout need no edge communication and edge latents never move;
  * the only exchange is the halo of SOURCE-node latents: before every MP layer each rank sends the
    rows of its owned nodes that appear as edge sources on other ranks.  Halo rows live behind the
    owned rows of the same tensor ([n_own + n_halo, H]), grouped by owner rank, so the receive side of
    one `all_to_all_single` (RCCL over xGMI; gloo in the CPU tests) lands in place.

`MusPartitionedForward` interprets the same per-class program as `nn/mus_gnn.py` on the local
sub-mesh; the arithmetic is delegated to an `impl` object: `HipImpl` (the HIP kernels, product path) or,
in tests only, an oracle-backed implementation that exercises the partition / halo logic on CPU + gloo.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, ops, plan
from .graph import Graph
from .ops import Source

SELU, TANH, NONE = _lib.ACT_SELU, _lib.ACT_TANH, _lib.ACT_NONE


# ------------------------------------------------------------------------------------- host-side partitioner
@dataclass
class LevelPart:
    """One level of one rank's sub-mesh (all index arrays are numpy int64 on the host)."""
    owned: np.ndarray              # global ids of owned nodes, ascending
    halo: np.ndarray               # global ids of halo nodes, grouped by owner rank, ascending inside a group
    halo_owner: np.ndarray         # owner rank of every halo node
    edge_ids: np.ndarray           # global ids (positions in the level's edge list) of owned edges, ascending
    edge_index: np.ndarray         # [2, E_own] in LOCAL node ids (owned first, then halo)
    send_idx: List[np.ndarray] = field(default_factory=list)   # per peer: local owned ids to send, in the peer's halo order
    recv_counts: List[int] = field(default_factory=list)       # per peer: number of halo rows received

    @property
    def n_own(self) -> int:
        return int(self.owned.shape[0])

    @property
    def n_halo(self) -> int:
        return int(self.halo.shape[0])


def coarse_topology(graph: Graph, levels: int):
    """Global edge lists of every level: level 1 from the Graph, level l from the topology part of
    `pool_edge` (nn/blocks.py:51-68) applied to level l-1."""
    lib = _lib.load()
    import ctypes as C
    edges = [np.ascontiguousarray(graph.edge_index.cpu().numpy().astype(np.int64))]
    for l in range(2, levels + 1):
        idx = np.ascontiguousarray(getattr(graph, f"idx{l - 1}_to_idx{l}").cpu().numpy().astype(np.int64))
        ei = edges[-1]
        n_edges = int(ei.shape[1])
        coarse = np.empty((2, max(n_edges, 1)), dtype=np.int64)
        perm = np.empty(max(n_edges, 1), dtype=np.int32)
        off = np.empty(n_edges + 1, dtype=np.int32)
        kept = C.c_int64(0)
        nc = lib.g4c_plan_pool_edge(idx.ctypes.data, int(idx.shape[0]), ei.ctypes.data, n_edges, coarse.ctypes.data,
                                    perm.ctypes.data, off.ctypes.data, C.byref(kept))
        if nc < 0:
            _lib.check(int(nc))
        edges.append(coarse.reshape(-1)[: 2 * nc].reshape(2, nc).copy())
    return edges


def _rcb(pos: np.ndarray, weight: np.ndarray, ids: np.ndarray, r0: int, nr: int, out: np.ndarray) -> None:
    """Recursive coordinate bisection: ranks [r0, r0 + nr) share the nodes `ids`; cut across the longer axis of their
    bounding box where the weight splits like floor(nr/2) : ceil(nr/2)."""
    if nr == 1 or ids.size == 0:
        out[ids] = r0
        return
    p = pos[ids]
    axis = int(np.argmax(p.max(0) - p.min(0)))
    order = np.lexsort((ids, p[:, axis]))              # coordinate, ties by id: deterministic on every rank
    cum = np.cumsum(weight[ids][order])
    n_lo = nr // 2
    cut = int(np.searchsorted(cum, cum[-1] * n_lo / nr, side="right"))
    cut = min(max(cut, 1), ids.size - 1) if ids.size > 1 else ids.size
    _rcb(pos, weight, ids[order[:cut]], r0, n_lo, out)
    _rcb(pos, weight, ids[order[cut:]], r0 + n_lo, nr - n_lo, out)


def assign_owners(graph: Graph, levels: int, world: int, method: Optional[str] = None) -> List[np.ndarray]:
    """Owner rank of every node of every level, decided on the coarsest level and inherited by the finer ones through their
    parents, balanced by the number of level-1 nodes below each coarsest node.  `method` (default: environment variable
    G4C_PARTITION, else "rcb"): "rcb" = recursive coordinate bisection (compact blocks: the halo of a part grows with its
    perimeter, 2-3x fewer halo rows than strips at 8 ranks on a square / cubic domain), "strips" = slabs along x."""
    import os
    method = method or os.environ.get("G4C_PARTITION", "rcb")
    if method not in ("rcb", "strips"):
        raise ValueError(f"unknown partition method {method!r} (rcb | strips)")
    n1 = int(graph.pos.size(0))
    weight = np.ones(n1, dtype=np.int64)
    maps = []
    for l in range(2, levels + 1):
        idx = getattr(graph, f"idx{l - 1}_to_idx{l}").cpu().numpy().astype(np.int64)
        maps.append(idx)
        weight = np.bincount(idx, weights=weight, minlength=int(idx.max()) + 1).astype(np.int64)
    pos_top = (graph.pos if levels == 1 else getattr(graph, f"pos_{levels}")).cpu().numpy().astype(np.float64)
    owner_top = np.empty(pos_top.shape[0], dtype=np.int64)
    if method == "strips":
        order = np.argsort(pos_top[:, 0], kind="stable")
        cum = np.cumsum(weight[order])
        owner_top[order] = np.minimum((cum - 1) * world // int(cum[-1]), world - 1)
    else:
        _rcb(pos_top, weight, np.arange(pos_top.shape[0]), 0, world, owner_top)
    owners = [None] * levels
    owners[levels - 1] = owner_top
    for l in range(levels - 1, 0, -1):
        owners[l - 1] = owners[l][maps[l - 1]]
    return owners


def uniform_edge_counts(parts: List[List[LevelPart]]) -> List[int]:
    """Per level, the smallest number of owned edges over the ranks.  Whether an MP layer hoists its first layer — and with it
    WHAT the halo exchange before it carries (the latents, or the first-layer products of the same size) — is decided on the
    edge count; every rank must decide alike or one rank's products land in another rank's latents.  Every rank builds the
    same partition table, so the minimum over the table is the same number everywhere."""
    return [min(int(p[l].edge_index.shape[1]) for p in parts) for l in range(len(parts[0]))]


def build_partition(graph: Graph, levels: int, world: int) -> List[List[LevelPart]]:
    """parts[rank][level-1] for every rank (the whole table is cheap and lets tests check consistency)."""
    edges = coarse_topology(graph, levels)
    owners = assign_owners(graph, levels, world)
    parts: List[List[LevelPart]] = [[] for _ in range(world)]
    for l in range(levels):
        ei, own = edges[l], owners[l]
        n = int(own.shape[0])
        tgt_owner = own[ei[1]]
        for r in range(world):
            owned = np.nonzero(own == r)[0]
            e_ids = np.nonzero(tgt_owner == r)[0]
            src = ei[0, e_ids]
            halo = np.unique(src[own[src] != r])
            h_owner = own[halo]
            order = np.lexsort((halo, h_owner))          # group by owner rank, ascending id inside
            halo, h_owner = halo[order], h_owner[order]
            g2l = np.full(n, -1, dtype=np.int64)
            g2l[owned] = np.arange(owned.shape[0])
            g2l[halo] = owned.shape[0] + np.arange(halo.shape[0])
            local_ei = np.stack([g2l[ei[0, e_ids]], g2l[ei[1, e_ids]]], 0)
            assert (local_ei >= 0).all()
            parts[r].append(LevelPart(owned=owned, halo=halo, halo_owner=h_owner, edge_ids=e_ids, edge_index=local_ei))
        # send lists: what q receives from r, in q's halo order
        for r in range(world):
            pr = parts[r][l]
            g2l_own = {}
            lut = np.full(n, -1, dtype=np.int64)
            lut[pr.owned] = np.arange(pr.n_own)
            pr.send_idx = []
            for q in range(world):
                pq = parts[q][l]
                need = pq.halo[pq.halo_owner == r]
                pr.send_idx.append(lut[need])
                assert (pr.send_idx[-1] >= 0).all()
            pr.recv_counts = [int((pr.halo_owner == q).sum()) for q in range(world)]
    return parts


# ------------------------------------------------------------------------------------- local sub-mesh on a device
class LocalMesh:
    """Rank-local tensors of the partitioned Graph, in local numbering, on `device`."""

    def __init__(self, graph: Graph, levels: int, parts: List[LevelPart], device: torch.device, rank: int, world: int):
        self.levels, self.device, self.rank, self.world = levels, device, rank, world
        self.parts = parts
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)   # noqa: E731
        p1 = parts[0]
        own1 = torch.from_numpy(p1.owned)
        self.n_own = [p.n_own for p in parts]
        self.n_halo = [p.n_halo for p in parts]
        self.owned_global = [t(p.owned) for p in parts]
        # node inputs of owned level-1 nodes (the reference's concat order: field, loc, glob, omega)
        self.inputs = {k: getattr(graph, k)[own1].contiguous().to(device) for k in ("field", "loc", "glob", "omega")
                       if hasattr(graph, k)}
        # per level: owned edges in local ids (+ the input edge_attr at level 1).  The coarse levels' edges (whose order nothing
        # observes: their latents are produced by this mesh's own pool_edge plan) are grouped by target, so that the coarse MP
        # layers need no CSR permutation and can aggregate on load; level 1 keeps the Graph's order (already target-grouped
        # for kNN meshes, and aligned with edge_attr).
        self._edge_index_np = [p.edge_index for p in parts]
        for l in range(1, levels):
            ei = parts[l].edge_index
            self._edge_index_np[l] = np.ascontiguousarray(ei[:, np.argsort(ei[1], kind="stable")])
        self.edge_index = [t(a) for a in self._edge_index_np]
        # edge count per level that rank-uniform decisions are taken on (see `uniform_edge_counts`); the local counts unless the
        # caller knows the whole partition table
        self.decision_edges = [int(a.shape[1]) for a in self._edge_index_np]
        # owned edges split by where their SENDER lives: interior (owned sender: computable before the halo exchange has
        # landed) / boundary (halo sender).  int32 row lists + the matching sender / target lists, for the overlapped MP layer.
        self.sub = []
        for l in range(levels):
            ei = self._edge_index_np[l]
            interior = ei[0] < parts[l].n_own
            ent = {}
            for tag, mask in (("int", interior), ("bnd", ~interior)):
                ids = np.nonzero(mask)[0].astype(np.int32)
                ent[tag] = (t(ids), t(ei[0][ids].astype(np.int32)), t(ei[1][ids].astype(np.int32)))
            self.sub.append(ent)
        self.edge_attr = graph.edge_attr[torch.from_numpy(p1.edge_ids)].contiguous().to(device)
        # inter-level maps for owned fine nodes -> local coarse id, relative positions, pooling of edges
        self.parent, self.rel, self.parent_full = [], [], []
        for l in range(1, levels):
            fine, coarse = parts[l - 1], parts[l]
            idx = getattr(graph, f"idx{l}_to_idx{l + 1}").cpu().numpy().astype(np.int64)
            n_c = int(idx.max()) + 1
            g2l_c = np.full(n_c, -1, dtype=np.int64)
            g2l_c[coarse.owned] = np.arange(coarse.n_own)
            g2l_c[coarse.halo] = coarse.n_own + np.arange(coarse.n_halo)
            par_own = g2l_c[idx[fine.owned]]
            assert (par_own >= 0).all() and (par_own < coarse.n_own).all(), "cluster not wholly owned"
            par_halo = g2l_c[idx[fine.halo]]
            # a fine halo whose parent is neither owned nor a coarse halo only feeds intra-cluster edges elsewhere;
            # point it at its own (unused) slot so the coarse-edge plan drops nothing it should keep
            self.parent.append(t(par_own))
            self.parent_full.append((par_own, par_halo))
            self.rel.append(getattr(graph, f"e_{l}{l + 1}")[torch.from_numpy(fine.owned)].contiguous().to(device))
        self.send_idx32 = [[t(ix.astype(np.int32)) for ix in p.send_idx] for p in parts]
        self.send_counts = [[int(ix.shape[0]) for ix in p.send_idx] for p in parts]
        self.recv_counts = [list(p.recv_counts) for p in parts]
        self._pool_plans: Dict[int, object] = {}

    def pool_edge_csr(self, l: int):
        """Plan that averages owned fine edges of level l into the owned coarse edges of level l+1 (rows in
        the local coarse edge order).  Built on the host from the global coarse edge ids (static)."""
        if l not in self._pool_plans:
            fine, coarse = self.parts[l - 1], self.parts[l]
            par_own, par_halo = self.parent_full[l - 1]
            par = np.concatenate([par_own, par_halo])
            fe = self._edge_index_np[l - 1]          # (the order the latents of that level are stored in)
            cr, cc = par[fe[0]], par[fe[1]]
            keep = np.nonzero((cr != cc) & (cr >= 0))[0]
            # local coarse edges, keyed like the local coarse edge list
            ce = self._edge_index_np[l]
            n_loc = coarse.n_own + coarse.n_halo
            key_c = ce[0] * n_loc + ce[1]
            order_c = np.argsort(key_c, kind="stable")
            key_f = cr[keep] * n_loc + cc[keep]
            pos = np.searchsorted(key_c[order_c], key_f)
            assert (key_c[order_c][np.minimum(pos, len(order_c) - 1)] == key_f).all(), "coarse edge missing on this rank"
            seg = order_c[pos]                                   # local coarse edge id of every kept fine edge
            csr = plan.build_csr(torch.from_numpy(seg), int(ce.shape[1]), self.device)
            perm = csr.perm.cpu().numpy() if csr.perm is not None else np.arange(len(seg))
            csr.perm = torch.from_numpy(keep[perm].astype(np.int32)).to(self.device)   # positions in the fine edge list
            self._pool_plans[l] = csr
        return self._pool_plans[l]


# ------------------------------------------------------------------------------------- halo exchange
class HaloExchanger:
    """v[n_own:] <- owned rows of the peers, one collective per call (`all_to_all_single` with split sizes:
    RCCL grouped send/recv over xGMI on GPUs, gloo in the CPU tests).  world == 1: no-op.

    A rank does not wait on peers it shares no edge with: with unequal splits torch's NCCL back-end issues the exchange as ONE
    group of ncclSend / ncclRecv calls and skips every peer whose count is zero (torch/csrc/cuda/nccl.cpp,
    all2all_single_unequal_split), i.e. it already is the neighbour-only point-to-point exchange — a hand-written
    batch_isend_irecv would enqueue the same calls.  The pack launch in front of it (`g4c_copy_cols` through the concatenated send
    lists) stays: a boundary row can go to two or three peers, the kernels' output index scatters a row to one place; where the
    exchange is overlapped with the interior edges (HipImpl.mp) the pack runs on the side stream with the collective.

    `force=True` (scripts/dist_check.py --force-exchange): enter the collective with world == 1 as well (zero-length splits) — on a
    single-GPU box that is the only way to execute "hipGraph capture with an RCCL collective inside" at all."""

    def __init__(self, mesh: LocalMesh, group=None, force: bool = False):
        self.mesh, self.group, self.force = mesh, group, force
        self._send_buf: Dict[tuple, torch.Tensor] = {}
        self._send_cat: Dict[int, Optional[torch.Tensor]] = {}
        self._side = None
        # bookkeeping for bench.py: exchanges entered and bytes moved since reset_stats(); with `timing` on, an event pair
        # around every collective (on the stream it is enqueued on; eager steps only)
        self.n_exchanges, self.bytes_sent, self.bytes_recv = 0, 0, 0
        self.timing, self.events = False, []

    def reset_stats(self) -> None:
        self.n_exchanges, self.bytes_sent, self.bytes_recv, self.events = 0, 0, 0, []

    def exchange_async(self, v: torch.Tensor, level: int):
        """`exchange` on a side stream (ordered after everything enqueued so far on the current stream); returns a handle for
        `wait`.  Lets the caller enqueue work that does not read the halo rows in between (HipImpl.mp: the interior edges)."""
        if not v.is_cuda:
            self.exchange(v, level)
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=v.device)
        cur = torch.cuda.current_stream(v.device)
        start = torch.cuda.Event()
        start.record(cur)
        with torch.cuda.stream(self._side):
            self._side.wait_event(start)
            self.exchange(v, level)
            done = torch.cuda.Event()
            done.record(self._side)
        if not torch.cuda.is_current_stream_capturing():
            v.record_stream(self._side)
        return done

    def wait(self, handle) -> None:
        if handle is not None:
            torch.cuda.current_stream().wait_event(handle)

    def exchange(self, v: torch.Tensor, level: int) -> None:
        m = self.mesh
        if m.world == 1 and not self.force:
            return
        # (no per-rank shortcut for an empty halo: the exchange is a collective, every rank of the group has to enter it, with
        # zero-length splits if it has nothing to send or receive at this level)
        import torch.distributed as dist
        width = int(v.size(1))
        n_send = sum(m.send_counts[level - 1])
        key = (level, width)
        buf = self._send_buf.get(key)
        if buf is None:
            buf = self._send_buf[key] = torch.empty((max(n_send, 1), width), dtype=v.dtype, device=v.device)
        # one pack launch: the per-peer send lists are concatenated in peer order (= the order all_to_all_single splits)
        cat = self._send_cat.get(level)
        if cat is None:
            parts = [idx for q, idx in enumerate(m.send_idx32[level - 1]) if m.send_counts[level - 1][q]]
            cat = self._send_cat[level] = torch.cat(parts) if parts else None
        if n_send:
            if v.is_cuda:
                ops.copy_cols(v, buf[:n_send], 0, scol0=0, width=width, idx32=cat, n_rows=n_send)
            else:   # CPU tests (gloo): host logic only
                buf[:n_send] = v[cat.long()]
        recv = v[m.n_own[level - 1]:]
        self.n_exchanges += 1
        self.bytes_sent += n_send * width * v.element_size()
        self.bytes_recv += int(sum(m.recv_counts[level - 1])) * width * v.element_size()
        ev = None
        if self.timing and v.is_cuda and not torch.cuda.is_current_stream_capturing():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_to_all_single(recv, buf[:n_send], output_split_sizes=m.recv_counts[level - 1],
                               input_split_sizes=m.send_counts[level - 1], group=self.group)
        if ev is not None:
            ev[1].record()
            self.events.append(ev)


# ------------------------------------------------------------------------------------- compute back-ends
class HipImpl:
    """Arithmetic of the partitioned forward on the HIP kernels (the product path)."""

    overlaps = True     # mp() takes `overlap=` (interior edges while the halo exchange is in flight)

    def __init__(self, model):
        self.m = model

    def new(self, rows: int, width: int, device) -> torch.Tensor:
        return torch.empty((rows, width), dtype=torch.float32, device=device)

    def _node_launch(self, mlp, srcs, n_own: int, act: int, v_out: torch.Tensor, next_name: Optional[str],
                     pr_out: Optional[torch.Tensor]):
        """`mlp` -> v_out (own rows); when `next_name` is the MP layer that consumes v_out and hoists its first layer, the
        same launch also emits its products (own rows of `pr_out`, and W1c v as a fresh tensor).  Returns them or None."""
        if next_name is not None and pr_out is not None:
            nxt = getattr(self.m, next_name).edge_mlp
            w = int(v_out.size(1))
            res = mlp.run_with_heads(srcs, n_own, act, nxt, nxt.input_size - 2 * w, [w, w], out=v_out,
                                     head_outs=[pr_out[:n_own], None])
            if res is not None:
                return pr_out, res[1][1]
        mlp.run_coded(srcs, n_own, act, out=v_out)
        return None

    def encode(self, mesh: LocalMesh, v_out: torch.Tensor, next_name: Optional[str] = None, pr_out: Optional[torch.Tensor] = None):
        m = self.m
        e = ops.static_launch("edge_encoder", [mesh.edge_attr],       # (static inside a rollout: ops.StaticCache)
                              lambda: m.edge_encoder.run_coded([Source(mesh.edge_attr)], int(mesh.edge_attr.size(0)), SELU))
        srcs = [Source(mesh.inputs[k]) for k in ("field", "loc", "glob", "omega") if k in mesh.inputs]
        return e, self._node_launch(m.node_encoder, srcs, mesh.n_own[0], SELU, v_out, next_name, pr_out)

    def hoists(self, n_edges: int) -> bool:
        """Whether an MP layer with this many local edges multiplies its node-side first-layer terms per node
        (MLP.run_hoisted) — then the halo exchange can carry those products instead of the latents."""
        from .nn import blocks as _blocks
        return n_edges >= _blocks.HOIST_MIN_ROWS and ops.mlp_precision() != "bf16"

    def mp(self, name: str, v: torch.Tensor, e: torch.Tensor, e_pending: int, edge_index: torch.Tensor, n_own: int,
           v_out: torch.Tensor, products=None, next_name: Optional[str] = None, pr_out: Optional[torch.Tensor] = None,
           overlap=None):
        """One MP layer on the local sub-mesh.  `products` = (W1r v over own + halo rows, W1c v over own rows) when the
        previous layer's node launch made them (and the halo rows of the first were exchanged): then `v`'s halo rows are
        not read at all.  `next_name` / `pr_out`: also emit the NEXT layer's products from this layer's node launch
        (own rows of `pr_out`, and a fresh tensor).  Returns (e', next products or None)."""
        blk = getattr(self.m, name)
        ep, csr = plan.edge_csr(edge_index, n_own)
        mean = blk.aggr == "mean"
        if overlap is not None:
            # `overlap` = (start, wait, subsets): the halo exchange of products[0] runs on a side stream while the edges whose
            # sender is owned are computed; the edges with a halo sender follow once it has landed.  Same arithmetic per
            # edge as the single launch (each edge row is independent), written into one tensor through out_idx.
            start, wait, sub = overlap
            handle = start()
            e_new = torch.empty((ep.n_edges, blk.edge_mlp.output_size), dtype=torch.float32, device=e.device)
            for tag in ("int", "bnd"):
                ids, row, col = sub[tag]
                if tag == "bnd":
                    wait(handle)
                if int(ids.numel()):
                    blk.edge_mlp.run_hoisted([Source(e, index=ids, pre_act=e_pending)], [(v, row), (v, col)], int(ids.numel()),
                                             products=products, out=e_new, out_idx32=ids)
            if ops.can_aggregate_on_load(csr, blk.edge_mlp.output_size, [blk.edge_mlp.output_size, int(v.size(1))]):
                agg_src = Source(e_new, segments=csr, seg_mean=mean)
            else:
                agg_src = Source(ops.segment_reduce(e_new, csr, mean))
            return e_new, self._node_launch(blk.node_mlp, [agg_src, Source(v[:n_own])], n_own, SELU, v_out, next_name, pr_out)
        if (not ops.can_fuse_aggregation(csr, blk.edge_mlp.output_size)
                and ops.can_aggregate_on_load(csr, blk.edge_mlp.output_size, [blk.edge_mlp.output_size, int(v.size(1))])):
            e_new = blk.edge_mlp.run_hoisted([Source(e, pre_act=e_pending)], [(v, ep.row), (v, ep.col)], ep.n_edges, products=products)
            agg_src = Source(e_new, segments=csr, seg_mean=mean)
        else:
            # (the edge launch reduces its own rows where it can, ops.can_fuse_aggregation; otherwise a separate reduction)
            agg = torch.empty((csr.n_seg, blk.edge_mlp.output_size), dtype=torch.float32, device=e.device)
            e_new = blk.edge_mlp.run_hoisted([Source(e, pre_act=e_pending)], [(v, ep.row), (v, ep.col)], ep.n_edges,
                                             products=products, agg=(csr, agg, mean))
            agg_src = Source(agg)
        return e_new, self._node_launch(blk.node_mlp, [agg_src, Source(v[:n_own])], n_own, SELU, v_out, next_name, pr_out)

    def down(self, name: str, v_own: torch.Tensor, rel: torch.Tensor, parent: torch.Tensor, n_coarse: int, e: torch.Tensor,
             e_pending: int, pool_csr, v_out: torch.Tensor):
        blk = getattr(self.m, name)
        msg = blk.down_mlp.run_coded([Source(rel), Source(v_own)], int(v_own.size(0)))
        csr = plan.segments_of_sorted(parent, n_coarse)
        ops.segment_reduce(msg, csr, True, TANH, out=v_out)
        return ops.segment_reduce(e, pool_csr, True, src_act=e_pending)

    def up(self, name: str, v_coarse: torch.Tensor, v_old_own: torch.Tensor, rel: torch.Tensor, parent: torch.Tensor,
           v_out: torch.Tensor, next_name: Optional[str] = None, pr_out: Optional[torch.Tensor] = None):
        blk = getattr(self.m, name)
        srcs = [Source(rel, negate=True), Source(v_coarse, plan.index32(parent)), Source(v_old_own)]
        return self._node_launch(blk.up_mlp, srcs, int(v_old_own.size(0)), TANH, v_out, next_name, pr_out)

    def decode(self, v_own: torch.Tensor, field: torch.Tensor, nf: int) -> torch.Tensor:
        return self.m.node_decoder.run_coded([Source(v_own)], int(v_own.size(0)), NONE, resid=field,
                                             resid_col0=int(field.size(1)) - nf)


class MusPartitionedForward:
    """The MuS-GNN V-cycle of `nn/mus_gnn.py` on one rank's sub-mesh, with a halo exchange of the node latents
    before every MP layer."""

    def __init__(self, program: Sequence[str], mesh: LocalMesh, impl, exchanger: HaloExchanger, width: int, nf: int):
        self.program, self.mesh, self.impl, self.xch, self.width, self.nf = tuple(program), mesh, impl, exchanger, width, nf
        import os
        # overlap every product exchange with the interior edges of the layer that consumes it (G4C_DIST_OVERLAP=0: off)
        self.overlap = os.environ.get("G4C_DIST_OVERLAP", "1") != "0"

    def _buf(self, level: int) -> torch.Tensor:
        m = self.mesh
        return self.impl.new(m.n_own[level - 1] + m.n_halo[level - 1], self.width, m.device)

    def forward(self) -> torch.Tensor:
        m, impl = self.mesh, self.impl
        level = 1
        v = self._buf(1)

        def wants(k: int, lvl: int) -> bool:      # program entry k is an MP layer that will hoist its first layer
            return (k < len(self.program) and self.program[k].startswith("mp")
                    and impl.hoists(m.decision_edges[lvl - 1]))

        # (W1r v [own + halo rows], W1c v [own rows]) of the next MP layer, when the launch producing v already made them
        w0 = wants(0, 1)
        e, prod = impl.encode(m, v[: m.n_own[0]], self.program[0] if w0 else None, self._buf(1) if w0 else None)
        e_pending = NONE
        stash = []
        for k, name in enumerate(self.program):
            n_own = m.n_own[level - 1]
            if name.startswith("down_mp"):
                prod = None
                stash.append((v, e, e_pending))
                v_c = self._buf(level + 1)
                e = impl.down(name, v[:n_own], m.rel[level - 1], m.parent[level - 1], m.n_own[level], e, e_pending,
                              m.pool_edge_csr(level), v_c[: m.n_own[level]])
                v, e_pending = v_c, NONE
                level += 1
            elif name.startswith("up_mp"):
                v_old, e, e_pending = stash.pop()
                level -= 1
                v_f = self._buf(level)
                wn = wants(k + 1, level)
                prod = impl.up(name, v, v_old[: m.n_own[level - 1]], m.rel[level - 1], m.parent[level - 1], v_f[: m.n_own[level - 1]],
                               self.program[k + 1] if wn else None, self._buf(level) if wn else None)
                v = v_f
            else:
                # products ride on the previous layer's node launch when this layer hoists its first layer: then the halo
                # exchange carries W1r v (same size) and the latents of the halo rows are never needed
                overlap = None
                if prod is not None and self.overlap and m.world > 1 and getattr(impl, "overlaps", False) and hasattr(self.xch, "exchange_async"):
                    pr = prod[0]
                    overlap = (lambda pr=pr, level=level: self.xch.exchange_async(pr, level), self.xch.wait, m.sub[level - 1])
                elif prod is not None:
                    self.xch.exchange(prod[0], level)
                else:
                    self.xch.exchange(v, level)
                v_new = self._buf(level)
                nxt = self.program[k + 1] if k + 1 < len(self.program) else ""
                want_next = nxt.startswith("mp") and impl.hoists(m.decision_edges[level - 1])
                kw = {"overlap": overlap} if overlap is not None else {}
                e, prod = impl.mp(name, v, e, e_pending, m.edge_index[level - 1], n_own, v_new[:n_own], products=prod,
                                  next_name=nxt if want_next else None, pr_out=self._buf(level) if want_next else None, **kw)
                v, e_pending = v_new, SELU
        return impl.decode(v[: m.n_own[0]], m.inputs["field"], self.nf)


class DistributedRollout:
    """`Rollout` over a node-partitioned mesh: every rank advances its owned nodes; `outputs` holds the owned
    rows ([n_own, nf*steps]); `gather_outputs()` assembles the global tensor on every rank (validation / I/O)."""

    def __init__(self, model, graph_cpu: Graph, max_steps: int, rank: int, world: int, device: torch.device, group=None,
                 capture: bool = True):
        import os
        capture = capture and os.environ.get("G4C_DIST_HIPGRAPH", "1") != "0"
        self.model, self.rank, self.world, self.device = model, rank, world, device
        program = model._PROGRAM
        self.n_global = int(graph_cpu.pos.size(0))
        self._perm = None
        self.nf = int(model.num_fields)
        if hasattr(model, "_ENCODERS"):          # REMuS-GNN: latents on edges / angles, edge-latent halo (partition_remus.py)
            from . import partition_remus as PR
            parts = PR.build_remus_partition(graph_cpu, world)
            self.mesh = PR.RemusLocalMesh(graph_cpu, parts[rank], device, rank, world)
            self.fwd = PR.RemusPartitionedForward(program, self.mesh, PR.RemusHipImpl(model, self.mesh), HaloExchanger(self.mesh, group))
        else:
            # level-1 nodes numbered along a Morton curve first (reorder.py): a part's local numbering is its owned nodes in
            # ascending global order, so it inherits the locality; gather_outputs maps the rows back
            if self.n_global >= 50_000 and os.environ.get("G4C_REORDER", "1") != "0":
                from .reorder import reorder_nodes
                re = reorder_nodes(graph_cpu)
                if re is not None:
                    graph_cpu, self._perm = re[0], re[1].to(device)
            levels = 1 + sum(1 for n in program if n.startswith("down_mp"))
            parts = build_partition(graph_cpu, levels, world)
            self.mesh = LocalMesh(graph_cpu, levels, parts[rank], device, rank, world)
            self.mesh.decision_edges = uniform_edge_counts(parts)
            width = int(model.node_encoder.output_size)
            self.fwd = MusPartitionedForward(program, self.mesh, HipImpl(model), HaloExchanger(self.mesh, group), width, self.nf)
        self.max_steps = max_steps
        self.field = self.mesh.inputs["field"] = self.mesh.inputs["field"].clone()
        self._out_steps = torch.zeros((max_steps, int(self.mesh.owned_global[0].numel()), self.nf), dtype=torch.float32, device=device)
        self.step_counter = torch.zeros(2, dtype=torch.int32, device=device)          # [step index, g4c_rollout_advance's ticket]
        self.steps_done = 0
        self.capture = capture and device.type == "cuda"
        self.capture_error = None if self.capture else "capture not requested"
        self._hipgraph, self._epoch = None, -1
        self.static = ops.StaticCache()        # per-mesh constants (edge / angle encoders), as nn.model.Rollout
        self._sites = getattr(model, "_range_sites", None)
        self._watch = ops.RangeWatch(device, self._sites) if ops.mlp_precision() == "f16x3" and device.type == "cuda" else None
        # optimistic "f16x3" (nn.model.Rollout): the input window the rollout started from, for the exact-range recomputation
        self._field0 = self.field.clone()
        self.exact_range = False

    def _one(self) -> None:
        with self.static:
            pred = self.fwd.forward()
        ops.rollout_advance(self.field, pred, self._out_steps, self.step_counter, self.nf)

    def step(self) -> None:
        """Step 1 eager (plans, packing), step 2 captured into a hipGraph together with its RCCL halo
        exchanges (every rank captures the same sequence), later steps replayed."""
        if self.steps_done >= self.max_steps:
            raise RuntimeError(f"rollout buffer holds {self.max_steps} steps")
        if self.exact_range and ops.mlp_precision() == "f16x3":
            old = ops.set_mlp_precision("bf16x6")
            try:
                self._step()
            finally:
                ops.set_mlp_precision(old)
        else:
            self._step()

    def validate(self) -> bool:
        """As nn.model.Rollout.validate, decided jointly: if ANY rank's launches clipped a value at the end of the fp16 range,
        every rank recomputes its steps in "bf16x6" (the halo exchanges pair up again) and stays in that arithmetic."""
        if ops.mlp_precision() != "f16x3" or self.exact_range or self.device.type != "cuda":
            return False
        if self._watch is None:
            self._watch = ops.RangeWatch(self.device, self._sites, drain=False)
        hit = self._watch.take()
        clipped = bool(hit)
        import torch.distributed as dist
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            flag = torch.tensor([1 if clipped else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=getattr(self.fwd.xch, "group", None))
            clipped = int(flag.item()) != 0
        if not clipped:
            return False
        import warnings
        n = self.steps_done
        warnings.warn(f"DistributedRollout (rank {self.rank}): the default 'f16x3' MLP arithmetic reached the end of the fp16 range "
                      f"(|x| >= 65504){' in ' + ', '.join(hit[:8]) if hit else ' on another rank'}: the {n} step(s) were recomputed in "
                      "'bf16x6' (fp32's exponent range) and this rollout continues in it.", RuntimeWarning, stacklevel=3)
        self.exact_range = True
        self.field.copy_(self._field0)
        self.step_counter.zero_()
        self._hipgraph, self._epoch = None, -1
        self.steps_done = 0
        self.run(n)
        return True

    def _step(self) -> None:
        with torch.no_grad():
            if self._epoch != -1 and ops.weights_epoch() != self._epoch:
                # (as nn.model.Rollout: the weights changed — load_state_dict, fit, invalidate_packed — so the captured step's packed
                # images are stale or freed; one eager step repacks, then the step is captured again.  Every rank sees the same
                # epoch sequence as long as every rank updates its replica of the model, which a partitioned rollout requires anyway)
                self._hipgraph, self._epoch = None, -1
            elif self.static.stale():
                # (a per-mesh constant edited in place, or another arithmetic — also just BEFORE the capture: as nn.model.Rollout — and, like new weights, something
                # every rank has to do alike, or a replaying rank would pair its captured collectives with an eager rank's)
                self._hipgraph, self._epoch = None, -1
            if self.steps_done == 0 or not self.capture or self._epoch == -1:
                self._one()
                self._epoch = ops.weights_epoch()
            elif self._hipgraph is None:
                torch.cuda.synchronize(self.device)
                hg, err = None, None
                self._pins = plan.snapshot()       # the graph bakes the plans' pointers in: keep them past cache eviction
                try:
                    hg = torch.cuda.CUDAGraph()
                    # (thread_local: the process group's watchdog thread polls its events while this thread captures; under
                    # the default global mode a call from that thread can invalidate the capture)
                    import torch.distributed as _dist
                    pg_live = _dist.is_available() and _dist.is_initialized()
                    with torch.cuda.graph(hg, capture_error_mode="thread_local" if (self.world > 1 or pg_live) else "global"):
                        self._one()
                except Exception as exc:   # keep the rollout alive on stacks where the collective cannot be captured
                    hg, err = None, f"{type(exc).__name__}: {exc}"
                # every rank takes the same path: one rank replaying while another launches eagerly would pair a captured
                # collective with an eager one
                ok = hg is not None
                import torch.distributed as dist
                if self.world > 1 and dist.is_available() and dist.is_initialized():     # (in-process test transports have no group)
                    torch.cuda.synchronize(self.device)
                    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=getattr(self.fwd.xch, "group", None))
                    if ok and int(flag.item()) == 0:
                        ok, err = False, "capture failed on another rank"
                if ok:
                    self._hipgraph = hg
                    self._hipgraph.replay()
                else:
                    import sys
                    self.capture_error = err
                    print(f"[graphs4cfd_amd] rank {self.rank}: hipGraph capture of the partitioned step failed "
                          f"({err}); continuing with eager launches", file=sys.stderr)
                    self.capture, self._hipgraph = False, None
                    torch.cuda.synchronize(self.device)
                    self._one()
            else:
                self._hipgraph.replay()
        self.steps_done += 1

    @property
    def captured(self) -> bool:
        return self._hipgraph is not None

    def run(self, n: int) -> None:
        for _ in range(n):
            self.step()

    @property
    def outputs(self) -> torch.Tensor:
        """The owned rows as [n_own, nf * max_steps] (a transposed copy of the step-major buffer)."""
        return ops.steps_to_columns(self._out_steps)

    def gather_outputs(self) -> torch.Tensor:
        import torch.distributed as dist
        self.validate()          # (a clipped rollout is recomputed in "bf16x6" on every rank before anything is gathered)
        mine = self.outputs
        full = torch.zeros((self.n_global, mine.size(1)), dtype=torch.float32, device=self.device)
        full[self.mesh.owned_global[0]] = mine
        if self.world > 1:
            dist.all_reduce(full)
        if self._perm is not None:        # rows back in the caller's numbering
            out = torch.empty_like(full)
            out[self._perm] = full
            return out
        return full

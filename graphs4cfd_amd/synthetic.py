"""Synthetic meshes in the reference's `Graph` attribute layout, and the published arch dicts.

The benchmark and the parity tests need inputs shaped exactly like what the reference's
pre-processing transforms emit (SURVEY.md §8(b) "Graph data layout"), at sizes (100k nodes) where
the reference's own Python-loop transforms are infeasible and on a box without PyG / torch_cluster.
These builders are vectorised CPU restatements of
  * `connect_knn` + `ScaleEdgeAttr`      (transforms/connect.py:9-92, transforms/scale.py:29),
  * `grid_clustering` / `GridClustering` (transforms/mus.py:9-65),
  * `guillard_coarsening`                (transforms/mugs.py:8-29),
  * `extend_graph`, `BuildRemusGraph`, `angleIndexDownMP` (transforms/remus.py:9-175),
  * `get_knn_interpolate_weights` / `BuildKnnInterpWeights` (transforms/interpolate.py:110-155),
checked against outputs of the reference transforms in tests/test_synthetic.py.  They are one-off
pre-processing (out of the hot path's HIP scope).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from scipy.spatial import cKDTree

from .graph import Graph


# ------------------------------------------------------------------------------ connectivity
def knn_neighbours(points: np.ndarray, queries: np.ndarray, k: int) -> np.ndarray:
    """Indices [len(queries), k] of the k nearest `points` of every query, by ascending distance."""
    _, nbr = cKDTree(points).query(queries, k=k)
    return np.asarray(nbr).reshape(len(queries), k)


def _bin_cloud(pos: torch.Tensor, k: int) -> dict:
    """Uniform cell grid over a device point cloud for `g4c_knn_grid[_query]`: cells of ~2k/pi (2-D) or ~2k/4.2 (3-D)
    points, so that one ring of cells almost always holds the k neighbours; cell coordinates in fp64 so that a point lies
    geometrically inside its cell; points stable-sorted by cell id."""
    import ctypes as C
    from . import _lib
    dev = _lib.require_hip(pos)
    n, dim = int(pos.size(0)), int(pos.size(1))
    p32 = pos.detach().float().contiguous()
    lo, hi = p32.min(0)[0].cpu(), p32.max(0)[0].cpu()
    ext = (hi.double() - lo.double())
    live = ext[ext > 0]
    per_cell = 2.0 * k / (np.pi if dim == 2 else 4.0 * np.pi / 3.0)
    h = float(np.float32((float(live.prod()) * per_cell / n) ** (1.0 / live.numel()))) if live.numel() else 1.0
    n_cells = [int(np.floor(float(e) / h)) + 1 for e in ext] + [1] * (3 - dim)
    grid = dict(dev=dev, n=n, dim=dim, h=h, n_cells=n_cells, lo=lo.double().to(dev),
                nc=(C.c_int32 * 3)(*n_cells), org=(C.c_float * 3)(*([float(v) for v in lo] + [0.0] * (3 - dim))))
    cell_sorted, order = torch.sort(_cell_ids(p32, grid), stable=True)
    total = n_cells[0] * n_cells[1] * n_cells[2]
    grid.update(cell_sorted=cell_sorted.int(), order=order.int(), pos_sorted=p32[order].contiguous(),
                cell_start=torch.searchsorted(cell_sorted, torch.arange(total + 1, device=dev)).int())
    return grid


def _cell_ids(p32: torch.Tensor, grid: dict) -> torch.Tensor:
    coord = ((p32.double() - grid["lo"]) / grid["h"]).floor().long()
    nc = grid["n_cells"]
    stride = [1, nc[0], nc[0] * nc[1]]
    cell = torch.zeros(p32.size(0), dtype=torch.long, device=p32.device)
    for ax in range(grid["dim"]):
        cell += coord[:, ax].clamp_(0, nc[ax] - 1) * stride[ax]
    return cell


def knn_neighbours_device(pos: torch.Tensor, k: int) -> torch.Tensor:
    """[n, k] int64 indices of the k nearest OTHER points of every point of `pos` (a device tensor), nearest first:
    the k-d-tree query of `connect_knn` as an exact cell-grid search on the GPU (`g4c_knn_grid`, csrc/knn_grid.hip)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    g = _bin_cloud(pos, k)
    out = torch.empty((g["n"], k), dtype=torch.long, device=g["dev"])
    _lib.check(lib.g4c_knn_grid(_lib.ptr(g["pos_sorted"]), _lib.ptr(g["cell_sorted"]), _lib.ptr(g["order"]),
                                _lib.ptr(g["cell_start"]), g["n"], g["dim"], g["nc"], g["org"], C.c_float(g["h"]), k,
                                _lib.ptr(out), _lib.stream_handle(g["dev"])))
    return out


def knn_query_device(points: torch.Tensor, queries: torch.Tensor, k: int) -> torch.Tensor:
    """[len(queries), k] int64 indices of the k nearest `points` of every query (device tensors), nearest first
    (`g4c_knn_grid_query`): `knn_neighbours` on the GPU."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    g = _bin_cloud(points, k)
    q32 = queries.detach().float().contiguous()
    _lib.require_hip(g["pos_sorted"], q32)
    q_cell = _cell_ids(q32, g).int()
    m = int(q32.size(0))
    out = torch.empty((m, k), dtype=torch.long, device=g["dev"])
    _lib.check(lib.g4c_knn_grid_query(_lib.ptr(g["pos_sorted"]), _lib.ptr(g["order"]), _lib.ptr(g["cell_start"]), g["n"],
                                      g["dim"], g["nc"], g["org"], C.c_float(g["h"]), _lib.ptr(q32), _lib.ptr(q_cell), m, k,
                                      _lib.ptr(out), _lib.stream_handle(g["dev"])))
    return out


def _connect_knn_periodic_device(pos: torch.Tensor, k: int, per) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """connect_knn for device positions with any set of periodic axes.  The reference ranks neighbours by the Euclidean distance
    between the points' embeddings (a periodic axis -> (cos, sin)(2 pi x / period) on the UNIT circle, transforms/connect.py:38-56: such
    an axis counts in radians).  In the coordinates y = (2 pi x / period for a periodic axis, x otherwise) the chord of a periodic axis
    grows monotonically with the wrapped difference dy and equals it up to a factor 1 - dy^2 / 24, so the k + 1 embedded-nearest
    points of a centre are among its 16 nearest in the wrapped y metric — which a non-periodic search finds when the points within a
    margin of every periodic face are repeated 2 pi away (ghosts, corners included).  The margin is checked per centre after the
    search (its candidates' radius must stay inside the ghosted region) and grown if needed; the candidates are then ordered by their
    float64 embedded distances (stably) and the host path's rule — drop the centre itself, or the farthest hit when coincident points
    pushed it out — leaves k per centre.  None: the cloud is too small for ghosts (a margin would reach half a period) or k too
    large for the grid search; the host path takes it."""
    dim, n, dev = int(pos.size(1)), int(pos.size(0)), pos.device
    kc = min(2 * k + 4, 15)          # (candidates per centre besides itself; the grid search returns at most 16)
    if n <= kc + 1 or kc < k + 2:
        return None
    x = pos.detach().double()
    lo, hi = x.min(0).values, x.max(0).values
    lengths = []
    for ax in range(dim):
        d = per[ax]
        lengths.append(None if d is None else (float(hi[ax] - lo[ax]) if isinstance(d, str) and d == "auto" else float(d)))
    pax = [ax for ax in range(dim) if lengths[ax] is not None]
    if any(lengths[ax] <= 0.0 or float(hi[ax] - lo[ax]) > lengths[ax] * (1 + 1e-12) for ax in pax):
        return None
    two_pi = 2 * np.pi
    y, emb_cols = x.clone(), []
    for ax in range(dim):
        if lengths[ax] is None:
            emb_cols.append(x[:, ax:ax + 1])
        else:
            a = two_pi / lengths[ax] * x[:, ax]
            emb_cols.append(torch.stack((torch.cos(a), torch.sin(a)), 1))
            y[:, ax] = two_pi / lengths[ax] * (x[:, ax] - lo[ax])                  # in [0, 2 pi]
    emb = torch.cat(emb_cols, 1)
    ylo = y.min(0).values
    ext = (y.max(0).values - ylo).clamp(min=1e-300)
    spacing = float(ext.prod()) ** (1.0 / dim) * float(n) ** (-1.0 / dim)
    margin = 2.0 * spacing * float(kc + 1) ** (1.0 / dim)
    ids = torch.arange(n, device=dev)
    for _attempt in range(4):
        if margin >= 0.5 * two_pi:
            return None
        pts, src = y, ids
        for ax in pax:          # (one axis after the other over the growing set: the corners get their combinations)
            low, high = pts[:, ax] < margin, pts[:, ax] > two_pi - margin
            shift = torch.zeros(dim, dtype=torch.float64, device=dev)
            shift[ax] = two_pi
            pts = torch.cat((pts, pts[low] + shift, pts[high] - shift), 0)
            src = torch.cat((src, src[low], src[high]), 0)
        cand_aug = knn_query_device(pts.float(), y.float(), kc + 1)                       # [n, kc + 1] indices into pts, nearest first (fp32)
        r = (pts[cand_aug[:, -1]] - y).norm(dim=1)
        room = torch.full((n,), float("inf"), dtype=torch.float64, device=dev)
        for ax in pax:
            room = torch.minimum(room, torch.minimum(y[:, ax], two_pi - y[:, ax]))
        need = float((r * (1 + 1e-6) - room).max())
        if need <= margin:
            break
        margin = 1.25 * need
    else:
        return None
    cand = src[cand_aug]                                                                   # original point numbers
    if bool((torch.sort(cand, dim=1).values.diff(dim=1) == 0).any()):                     # a point AND its own periodic image among one
        return None                                                                        # centre's candidates (tiny clouds): host path
    d2 = ((emb[cand] - emb[:, None, :]) ** 2).sum(-1)
    d2s, order = torch.sort(d2, dim=1, stable=True)
    # Completeness: the candidates are the kc + 1 nearest in the WRAPPED angle metric y, the reference orders by the
    # EMBEDDED (chord) metric.  A point at chord distance D is at most 2 asin(D / 2) away in y (all of it along one periodic axis), so
    # every point nearer than the (k + 1)-th chord distance is among the candidates iff that bound stays inside the candidates'
    # y-radius; otherwise a true neighbour may be missing: host path.
    dk = d2s[:, k].clamp(min=0).sqrt()
    if bool((dk >= 2.0).any()) or bool((2.0 * torch.asin((dk * 0.5).clamp(max=1.0)) * (1 + 1e-9) > r).any()):
        return None
    order = order[:, :k + 1]
    nbr = torch.gather(cand, 1, order)
    is_self = nbr == ids[:, None]
    drop = torch.where(is_self.any(1), is_self.int().argmax(1), torch.full((n,), k, device=dev))
    keep = torch.ones((n, k + 1), dtype=torch.bool, device=dev)
    keep[ids, drop] = False
    row = nbr[keep]
    col = ids.repeat_interleave(k)
    edge_attr = pos[col] - pos[row]
    for ax in pax:
        d, c = lengths[ax], edge_attr[:, ax]
        edge_attr[:, ax] = torch.where(c < -d / 2, c + d, torch.where(c > d / 2, c - d, c))
    return torch.stack([row, col], 0), edge_attr


def connect_knn(pos: torch.Tensor, k: int, period=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """`connect_knn` (transforms/connect.py:9-72): edges neighbour -> centre, grouped by centre, k per centre, nearest
    first; edge_attr = pos[col] - pos[row].  `period`: one entry per axis — None (not periodic), a length, or "auto" (the
    extent of the point cloud along that axis).  A periodic axis enters the neighbour search as a point on a circle,
    (cos, sin)(2 pi x / period) (:38-56), and its edge components are wrapped into [-period/2, period/2] (:62-71)."""
    dim = int(pos.size(1))
    if dim not in (2, 3):
        raise ValueError(f"Invalid dimension: {dim}, must be 2 or 3.")
    per = [None] * dim if (period is None or all(d is None for d in period)) else list(period)
    if len(per) != dim:
        raise ValueError(f"period needs {dim} entries")
    if pos.is_cuda and all(d is None for d in per):
        # positions resident on the GPU, no periodic axis: the search runs there (the result is the host path's)
        n = int(pos.size(0))
        row = knn_neighbours_device(pos, k).reshape(-1)
        col = torch.arange(n, device=pos.device).repeat_interleave(k)
        return torch.stack([row, col], 0), pos[col] - pos[row]
    if pos.is_cuda and dim == 2 and sum(d is not None for d in per) == 1:
        # positions on the GPU, ONE periodic axis of a 2-D cloud: its embedding (cos, sin, other coordinate) is 3-D, which the cell-grid
        # search takes.  The grid works in fp32, the reference path in the embedded float64 coordinates: 2k + 2 candidates per
        # centre come from the grid, their float64 distances order them (stable: the grid's order for equal distances), the nearest
        # k stay — the host path's neighbours unless more than k + 2 points tie within fp32 resolution at the k-th distance.
        # (two periodic axes embed in 4-D, periodic 3-D clouds in >= 4-D: _connect_knn_periodic_device below)
        n = int(pos.size(0))
        ax = 0 if per[0] is not None else 1
        x = pos[:, ax].detach().double()
        d = float(x.max() - x.min()) if isinstance(per[ax], str) and per[ax] == "auto" else float(per[ax])
        circ = torch.stack((torch.cos(2 * np.pi / d * x), torch.sin(2 * np.pi / d * x)), 1)
        other = pos[:, 1 - ax].detach().double().unsqueeze(1)
        emb = torch.cat((circ, other), 1) if ax == 0 else torch.cat((other, circ), 1)          # (the reference's column order: x part, y part)
        kc = min(n - 1, 2 * k + 2)
        cand = knn_neighbours_device(emb.float(), kc)                                           # [n, kc], nearest first in fp32
        d2 = ((emb[cand] - emb[:, None, :]) ** 2).sum(-1)
        pick = torch.sort(d2, dim=1, stable=True)[1][:, :k]
        row = torch.gather(cand, 1, pick).reshape(-1)
        col = torch.arange(n, device=pos.device).repeat_interleave(k)
        edge_attr = pos[col] - pos[row]
        c = edge_attr[:, ax]
        edge_attr[:, ax] = torch.where(c < -d / 2, c + d, torch.where(c > d / 2, c - d, c))
        return torch.stack([row, col], 0), edge_attr
    if pos.is_cuda:
        # positions on the GPU, two or more periodic axes (or a periodic 3-D cloud): the embedding has >= 4 dimensions, more than the
        # cell grid bins.  The candidates then come from a search in the RAW coordinates over the cloud + ghost copies of the points
        # near the periodic faces (_knn_periodic_candidates), and the embedded float64 distances order them as above.
        hit = _connect_knn_periodic_device(pos, k, per)
        if hit is not None:
            return hit
    lengths, cols = [], []
    for ax in range(dim):
        x = pos[:, ax].detach().cpu().double()
        d = per[ax]
        if d is None:
            lengths.append(None)
            cols.append(x.unsqueeze(1))
        else:
            d = float(x.max() - x.min()) if isinstance(d, str) and d == "auto" else float(d)
            lengths.append(d)
            cols.append(torch.stack((torch.cos(2 * np.pi / d * x), torch.sin(2 * np.pi / d * x)), 1))
    p = torch.cat(cols, 1).numpy()
    n = p.shape[0]
    nbr = knn_neighbours(p, p, k + 1)
    # exactly one of the k + 1 hits is dropped per centre — the centre itself, or, when coincident points pushed it out of the
    # hit list, the farthest — so that the in-degree is k everywhere (REMuS's edgeScalarToNodeVector relies on it)
    is_self = nbr == np.arange(n)[:, None]
    drop = np.where(is_self.any(1), is_self.argmax(1), k)
    keep = np.ones((n, k + 1), dtype=bool)
    keep[np.arange(n), drop] = False
    centre = np.repeat(np.arange(n), k + 1).reshape(n, k + 1)
    row = torch.from_numpy(nbr[keep].astype(np.int64))
    col = torch.from_numpy(centre[keep].astype(np.int64))
    edge_index = torch.stack([row, col], 0)
    edge_attr = pos[col] - pos[row]
    for ax, d in enumerate(lengths):
        if d is not None:
            c = edge_attr[:, ax]
            edge_attr[:, ax] = torch.where(c < -d / 2, c + d, torch.where(c > d / 2, c - d, c))
    return edge_index, edge_attr


def grid_clustering(pos_1: torch.Tensor, cell_size_2: float):
    """`grid_clustering` (transforms/mus.py:9-38): voxel-grid clusters -> (pos_2, cluster_2, mask_2,
    idx1_to_idx2, e_12).  Voxel id = sum_d floor((p_d - min_d)/size) * stride_d with
    stride = exclusive cumprod of floor((max-min)/size)+1 (torch_cluster.grid_cluster).

    With `pos_1` on the GPU everything stays there: the voxel ids and their sorted unique set are device tensors and the
    cluster centres are one `g4c_segment_reduce` launch over the clusters' CSR (rows summed in node order, then divided by
    the count: the same bits as the host path's sequential `index_add_`).  Divisors are device tensors, not Python
    scalars, so that the quotients are true divisions on both sides (a host scalar divisor becomes a multiplication by
    its reciprocal in torch's device kernel)."""
    n, dim = pos_1.shape
    dev = pos_1.device
    size = torch.full((dim,), float(cell_size_2), dtype=pos_1.dtype, device=dev)
    start, end = pos_1.min(0)[0], pos_1.max(0)[0]
    nvox = (end - start).true_divide(size).to(torch.long) + 1
    stride = torch.cat([torch.ones(1, dtype=torch.long, device=dev), nvox.cumprod(0)[:-1]])
    cluster_2 = ((pos_1 - start).true_divide(size).to(torch.long) * stride).sum(1)
    mask_2, idx1_to_idx2 = torch.unique(cluster_2, sorted=True, return_inverse=True)
    n2 = mask_2.numel()
    if pos_1.is_cuda:
        from . import ops, plan
        pos_2 = ops.segment_reduce(pos_1.detach().float().contiguous(), plan.build_csr(idx1_to_idx2, n2, dev), mean=True)
        pos_2 = pos_2.to(pos_1.dtype)
    else:
        cnt = torch.bincount(idx1_to_idx2, minlength=n2).clamp(min=1).to(pos_1.dtype)
        pos_2 = torch.zeros(n2, dim, dtype=pos_1.dtype).index_add_(0, idx1_to_idx2, pos_1) / cnt[:, None]
    e_12 = (pos_2[idx1_to_idx2] - pos_1) / size[0]
    return pos_2, cluster_2, mask_2, idx1_to_idx2, e_12


def add_grid_levels(graph: Graph, cells_size: Sequence[float]) -> Graph:
    """`GridClustering.__call__` (transforms/mus.py:56-65)."""
    pos = graph.pos
    for lvl, cell in enumerate(cells_size, start=2):
        p, c, m, idx, e = grid_clustering(pos, cell)
        setattr(graph, f"pos_{lvl}", p)
        setattr(graph, f"cluster_{lvl}", c)
        setattr(graph, f"mask_{lvl}", m)
        setattr(graph, f"idx{lvl - 1}_to_idx{lvl}", idx)
        setattr(graph, f"e_{lvl - 1}{lvl}", e)
        pos = p
    return graph


def default_cells(n: int, dim: int, levels: int) -> List[float]:
    """Cell sizes 2h * 2^(l-2), h = n^(-1/dim): ~2^dim nodes per level-2 cell, doubling per level like the
    examples' 0.15/0.30/0.60."""
    h = float(n) ** (-1.0 / dim)
    return [2.0 * h * (2 ** i) for i in range(levels - 1)]


def true_divide_by(t: torch.Tensor, divisor: float) -> torch.Tensor:
    """t / divisor as a true division on either side (a host-scalar divisor is a multiplication by its reciprocal in
    torch's device kernel: one ulp away from the host result)."""
    return t / torch.full((), float(divisor), dtype=t.dtype, device=t.device) if t.is_cuda else t / divisor


def mus_graph(n: int, levels: int = 1, k: int = 6, dim: int = 2, nf: int = 3, n_in: int = 1, seed: int = 0,
              r: Optional[float] = None, cells: Optional[Sequence[float]] = None, loc: bool = False, device=None) -> Graph:
    """Synthetic MuS-GNN input: uniform random points in [0,1]^dim, kNN edges scaled by 1/(2r),
    grid-clustered coarse levels, `field ~ N(0,1)`, `glob ~ U(0,1)`, `omega = U(0,1) > 0.9`.  With `device` (a GPU) the
    points are uploaded first and the edges / levels are built there (§4.5 of DESIGN.md): the same graph, bit for bit."""
    gen = torch.Generator().manual_seed(seed)
    pos = torch.rand(n, dim, generator=gen)
    if device is not None:
        pos = pos.to(device)
    g = Graph(pos=pos)
    g.edge_index, ea = connect_knn(pos, k)
    if r is None:
        r = 2.0 * float(n) ** (-1.0 / dim)   # keeps |edge_attr| = O(1) at every mesh size
    g.edge_attr = true_divide_by(ea, 2 * r)
    if levels > 1:
        add_grid_levels(g, cells if cells is not None else default_cells(n, dim, levels))
    g.field = torch.randn(n, nf * n_in, generator=gen).to(pos.device)
    if loc:
        g.loc = torch.randn(n, 2, generator=gen).to(pos.device)
    g.glob = torch.rand(n, 1, generator=gen).to(pos.device)
    g.omega = (torch.rand(n, 1, generator=gen) > 0.9).float().to(pos.device)
    return g


# ------------------------------------------------------------------------------ REMuS graphs
def guillard_coarsening(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Node-nested greedy coarsening (transforms/mugs.py:8-29): visit nodes in order; a node still
    marked coarse removes all its k senders.  Sequential by definition."""
    if edge_index.is_cuda:
        # data-parallel rounds; on a numbering with long removal chains (a mesh numbered along a line: up to O(n) rounds) the bounded
        # number of rounds runs out and the sequential visit below takes over on a host copy
        mask = guillard_coarsening_rounds(edge_index, num_nodes, max_rounds=GUILLARD_MAX_ROUNDS, strict=False)
        if mask is not None:
            return mask
        return guillard_coarsening(edge_index.cpu(), num_nodes).to(edge_index.device)
    row = edge_index[0].numpy()
    k = int((edge_index[1] == 0).sum())
    senders = row.reshape(-1, k)
    coarse = np.ones(num_nodes, dtype=bool)
    for i in range(senders.shape[0]):
        if coarse[i]:
            coarse[senders[i]] = False
    return torch.from_numpy(coarse)


GUILLARD_MAX_ROUNDS = 256      # (random / Morton / spatially sorted numberings of 100k-node meshes need 20 - 60 rounds)


def guillard_coarsening_rounds(edge_index: torch.Tensor, num_nodes: int, max_rounds: int = 100000, strict: bool = True):
    """The same mask as the sequential visit of `guillard_coarsening`, by rounds of data-parallel tensor ops (any device).
    Node l is *active* (still coarse when it is visited, so it removes its k senders) iff no active node visited before it
    lists l among its senders: a lexicographically-first recursion over the arcs l -> j (j a sender of l, l < j).  Each round
    decides every node all of whose earlier removers are decided — inactive as soon as one of them is active, active once
    all are inactive; the number of rounds is the longest chain of such arcs, not the number of nodes.  The final mask:
    a node stays coarse iff no active node (earlier or later) removes it.  `strict=False`: None instead of an error when
    `max_rounds` rounds leave nodes undecided."""
    dev = edge_index.device
    k = int((edge_index[1] == 0).sum())
    remover = torch.arange(num_nodes, device=dev).repeat_interleave(k)        # node l, visited in index order ...
    removed = edge_index[0]                                                     # ... removes its sender j
    fwd = remover < removed                                                     # arcs that act before j's own visit
    a_l, a_j = remover[fwd], removed[fwd]
    UNDECIDED, ACTIVE, INACTIVE = 0, 1, 2
    state = torch.zeros(num_nodes, dtype=torch.int8, device=dev)
    ones = torch.ones_like(a_j, dtype=torch.int32)
    n_pred = torch.zeros(num_nodes, dtype=torch.int32, device=dev).index_add_(0, a_j, ones)
    for _ in range(max_rounds):
        s_l = state[a_l]
        n_act = torch.zeros(num_nodes, dtype=torch.int32, device=dev).index_add_(0, a_j, (s_l == ACTIVE).int())
        n_ina = torch.zeros(num_nodes, dtype=torch.int32, device=dev).index_add_(0, a_j, (s_l == INACTIVE).int())
        open_ = state == UNDECIDED
        new_state = torch.where(open_ & (n_act > 0), torch.full_like(state, INACTIVE),
                                torch.where(open_ & (n_ina == n_pred), torch.full_like(state, ACTIVE), state))
        if torch.equal(new_state, state):
            break
        state = new_state
    if bool((state == UNDECIDED).any()):
        if not strict:
            return None         # out of rounds (every round decides at least one node, so this is not a cycle): the caller falls back
        raise RuntimeError(f"guillard_coarsening_rounds: {max_rounds} rounds were not enough (removal chains longer than that); "
                           "raise max_rounds or use the sequential visit guillard_coarsening() on a host copy")
    hit = torch.zeros(num_nodes, dtype=torch.int32, device=dev).index_add_(0, removed, (state[remover] == ACTIVE).int())
    return hit == 0


def extend_graph(edge_index: torch.Tensor, edge_attr: torch.Tensor, k: int):
    """`extend_graph` (transforms/remus.py:9-45), vectorised: unit vectors, angle_index [2, k|E|]
    (row = the k edges entering the sender of each edge, col = the edge), angle_attr [k|E|, 4]."""
    row, col = edge_index[0], edge_index[1]
    n_edges = edge_index.size(1)
    size = edge_attr.norm(2, dim=1, keepdim=True)
    unit = edge_attr / size
    # edges are grouped by receiver, k per receiver, receivers in increasing order: the edges entering
    # node r are the block starting at k * (rank of r among the receivers)
    receivers = col[::k].contiguous()
    rank = torch.searchsorted(receivers, row)
    dev = edge_index.device      # host or GPU: index arithmetic and torch elementwise ops only
    a_row = (rank[:, None] * k + torch.arange(k, device=dev)[None, :]).reshape(-1)
    a_col = torch.arange(n_edges, device=dev).repeat_interleave(k)
    cos = (unit[a_row] * unit[a_col]).sum(1)
    sin = unit[a_row, 0] * unit[a_col, 1] - unit[a_row, 1] * unit[a_col, 0]
    attr = torch.cat([size[a_row], size[a_col], cos[:, None], sin[:, None]], dim=1)
    return unit, torch.stack([a_row, a_col], 0), attr


def angle_index_down(edge_index1, edge_attr1, edge_index2, edge_attr2, coarse_index2, k):
    """`BuildRemusGraph.angleIndexDownMP` (transforms/remus.py:151-176), vectorised."""
    recv1 = edge_index1[1][::k].contiguous()
    rank1 = torch.searchsorted(recv1, coarse_index2)
    in_edges = rank1[:, None] * k + torch.arange(k, device=rank1.device)[None, :]   # [n2, k] level-1 edges entering each coarse node
    # level-2 edges leaving each coarse node, in edge order
    snd2 = edge_index2[0]
    order = torch.argsort(snd2, stable=True)
    rank_snd = torch.searchsorted(coarse_index2, snd2[order])
    num_out = torch.bincount(rank_snd, minlength=coarse_index2.numel())
    out_edges = order                                                  # grouped by coarse sender, ascending edge id
    row = torch.repeat_interleave(in_edges, num_out, dim=0).reshape(-1)
    col = torch.repeat_interleave(out_edges, k)
    s1 = edge_attr1.norm(2, dim=1, keepdim=True)
    s2 = edge_attr2.norm(2, dim=1, keepdim=True)
    u1, u2 = edge_attr1 / s1, edge_attr2 / s2
    cos = (u1[row] * u2[col]).sum(1)
    sin = u1[row, 0] * u2[col, 1] - u1[row, 1] * u2[col, 0]
    return torch.stack([row, col], 0), torch.cat([s1[row], s2[col], cos[:, None], sin[:, None]], dim=1)


def knn_interp_weights(pos_x: torch.Tensor, pos_y: torch.Tensor, k: int):
    """`get_knn_interpolate_weights` (transforms/interpolate.py:110-131): for every node of pos_y its k
    nearest nodes of pos_x; weights = 1 / max(squared distance, 1e-16)."""
    if pos_x.is_cuda:     # both clouds resident on the GPU: the search runs there (same neighbours, same order)
        x_idx = knn_query_device(pos_x, pos_y, k).reshape(-1)
    else:
        x_idx = torch.from_numpy(knn_neighbours(pos_x.double().numpy(), pos_y.double().numpy(), k).reshape(-1).astype(np.int64))
    y_idx = torch.arange(pos_y.size(0), device=pos_x.device).repeat_interleave(k)
    diff = pos_x[x_idx] - pos_y[y_idx]
    w = 1.0 / torch.clamp((diff * diff).sum(-1, keepdim=True), min=1e-16)
    return y_idx, x_idx, w


# ------------------------------------------------------------------------------ gMuS-GNN graphs
MUGS_LAYERS = {
    "NsTwoGuillardScaleGNN": ("mp111 mp112 mp113 mp114 mp21 mp22 mp23 mp24 mp121 mp122 mp123 mp124", ("mp121",)),
    "NsThreeGuillardScaleGNN": ("mp111 mp112 mp113 mp114 mp211 mp212 mp31 mp32 mp33 mp34 mp221 mp222 mp121 mp122 mp123 mp124",
                                ("mp221", "mp121")),
    "NsFourGuillardScaleGNN": ("mp111 mp112 mp113 mp114 mp211 mp212 mp311 mp312 mp41 mp42 mp43 mp44 mp321 mp322 mp221 mp222 "
                               "mp121 mp122 mp123 mp124", ("mp321", "mp221", "mp121")),
}


def mugs_arch(model: str, hidden: int = 128, nf: int = 3, node_in: int = 5, dim: int = 2) -> dict:
    """arch dict of a gMuS-GNN class with the published layout (nn/mugs_gnn.py:15-42): the first MP after every
    up-sampling takes node latents [interpolated | stashed] = 2H wide."""
    H = hidden
    layers, wide = MUGS_LAYERS[model]
    levels = {"NsTwoGuillardScaleGNN": 2, "NsThreeGuillardScaleGNN": 3, "NsFourGuillardScaleGNN": 4}[model]
    arch = {"edge_encoder": (dim, (H, H, H), False), "node_encoder": (node_in, (H, H, H), False)}
    for l in range(2, levels + 1):
        arch[f"edge_encoder{l}"] = (dim, (H, H, H), False)
    for name in layers.split():
        vw = 2 * H if name in wide else H
        arch[name] = ((H + 2 * vw, (H, H, H), True), (H + vw, (H, H, H), True))
    arch["decoder"] = (H, (H, H, nf), False)
    return arch


def mugs_graph(n: int, levels: int = 2, k: int = 6, seed: int = 0, nf: int = 3, device=None) -> Graph:
    """Synthetic gMuS-GNN input with the attribute layout of `GuillardCoarseningAndConnectKNN` + `BuildKnnInterpWeights`
    (transforms/mugs.py:32-89, interpolate.py:133-155): kNN level 1, node-nested Guillard coarsening, kNN per coarse level
    (edge_index{l} in level-1 ids), interpolation indices / weights between consecutive levels."""
    gen = torch.Generator().manual_seed(seed)
    pos = torch.rand(n, 2, generator=gen)
    if device is not None:
        pos = pos.to(device)       # every level is then built on the GPU
    dev = pos.device
    g = Graph(pos=pos)
    r = 2.0 * float(n) ** -0.5
    g.edge_index, ea = connect_knn(pos, k)
    g.edge_attr = true_divide_by(ea, 2 * r)
    masks = [torch.ones(n, dtype=torch.bool, device=dev)]
    ei_local = g.edge_index
    for l in range(2, levels + 1):
        prev = masks[-1]
        cm = torch.zeros(n, dtype=torch.bool, device=dev)
        cm[prev] = guillard_coarsening(ei_local, int(prev.sum()))
        idx = cm.nonzero().reshape(-1)
        if idx.numel() <= k:
            raise ValueError(f"level {l} has only {idx.numel()} nodes (need > k = {k}): use a larger mesh")
        ei_local, ea = connect_knn(pos[idx], k)
        setattr(g, f"coarse_mask{l}", cm)
        setattr(g, f"edge_index{l}", idx[ei_local])
        setattr(g, f"edge_attr{l}", true_divide_by(ea, 2 * r * 2 ** (l - 1)))
        y, x, w = knn_interp_weights(pos[cm], pos[prev], k)
        setattr(g, f"y_idx_{l}{l - 1}", y); setattr(g, f"x_idx_{l}{l - 1}", x); setattr(g, f"weights_{l}{l - 1}", w)
        masks.append(cm)
    g.field = torch.randn(n, nf, generator=gen).to(dev)
    g.glob = torch.rand(n, 1, generator=gen).to(dev)
    g.omega = (torch.rand(n, 1, generator=gen) > 0.9).float().to(dev)
    return g


def pinv_k2(u: torch.Tensor) -> torch.Tensor:
    """Moore-Penrose pseudo-inverse of a batch of k x 2 blocks, [n, k, 2] -> [n, 2, k], in closed form on any device — what the
    reference obtains from an SVD (`edgeUnitVector.view(n, -1, 2).pinverse()`, transforms/remus.py:59,126-137).  Full column rank:
    (U^T U)^-1 U^T with the 2 x 2 normal matrix formed and inverted in float64 (the squared condition number of nearly collinear
    unit vectors stays far inside float64; one rounding to the input dtype at the end).  Rank one (all k vectors collinear — the
    SVD drops the second singular value below rcond * sigma_1, torch's default rcond = 1e-15 * max(k, 2)): U^T / ||U||_F^2; zero block: 0.
    Elementwise tensor ops only: no host round trip, no per-block LAPACK call (the host SVD was what was left of the device
    build of a REMuS graph: 237 ms at 100k nodes).

    How close to the reference's stored inverses (tests/test_synthetic.py): rtol 1e-5 / atol 1e-6 on the golden REMuS graph's
    blocks (well conditioned); 2e-3 relative on nearly collinear blocks (angles within 1e-3 rad, condition ~1e3) — that is the
    float32 SVD's own error there, this routine works in float64; and NOT equal on exactly rank-deficient blocks: the reference's
    rcond = 1e-15 lies below float32 resolution, so its float32 SVD keeps a noise-level second singular value (~1e-7 sigma_1) and
    inverts it, while this returns the minimum-norm pseudo-inverse (the rank is decided on the exact float64 singular values).  A
    degenerate stencil (all k neighbours on one line through the node) therefore gives a different — finite, meaningful — matrix
    than the reference's arbitrary one."""
    if u.dim() != 3 or u.size(2) != 2:
        raise ValueError(f"pinv_k2 expects [n, k, 2], got {tuple(u.shape)}")
    x, y = u[..., 0].double(), u[..., 1].double()
    a, b, c = (x * x).sum(1), (x * y).sum(1), (y * y).sum(1)             # U^T U = [[a, b], [b, c]]
    tr, det = a + c, a * c - b * b
    # singular values squared: (tr +- sqrt(tr^2 - 4 det)) / 2; the block is rank one when sigma_2 <= rcond * sigma_1
    disc = torch.sqrt(torch.clamp(tr * tr - 4.0 * det, min=0.0))
    s1, s2 = 0.5 * (tr + disc), torch.clamp(0.5 * (tr - disc), min=0.0)
    rcond = 1e-15 * max(int(u.size(1)), 2)
    full = s2 > (rcond * rcond) * s1
    safe_det = torch.where(full, det, torch.ones_like(det))
    px = (c[:, None] * x - b[:, None] * y) / safe_det[:, None]            # first row of (U^T U)^-1 U^T
    py = (a[:, None] * y - b[:, None] * x) / safe_det[:, None]
    safe_tr = torch.where(tr > 0, tr, torch.ones_like(tr))
    rx, ry = x / safe_tr[:, None], y / safe_tr[:, None]                    # rank one: U^T / ||U||_F^2 (zero block: 0 / 1)
    out = torch.stack((torch.where(full[:, None], px, rx), torch.where(full[:, None], py, ry)), 1)
    return out.to(u.dtype)


def remus_graph(n: int, k: int = 5, seed: int = 0, scale: Optional[Sequence[float]] = None,
                pos: Optional[torch.Tensor] = None, period=None, device=None) -> Graph:
    """Synthetic 3-level REMuS-GNN input: `BuildRemusGraph(num_levels=3, k, scale_edge_length)` +
    `BuildKnnInterpWeights(k)` (transforms/remus.py:84-148, interpolate.py:134-155).  With the positions on a GPU (`device`,
    or a device `pos`) the three kNN searches, the interpolation searches and the angle tables are built there (DESIGN.md
    §4.5), the two Guillard coarsenings as data-parallel rounds, the pseudo-inverses in closed form (`pinv_k2`); a host graph uses the
    reference's SVD call."""
    gen = torch.Generator().manual_seed(seed)
    if pos is None:
        pos = torch.rand(n, 2, generator=gen)
    if device is not None:
        pos = pos.to(device)
    dev = pos.device
    n = pos.size(0)
    if scale is None:
        h = 2.0 * float(n) ** -0.5
        scale = (h, 2 * h, 4 * h)
    g = Graph(pos=pos)
    g.edge_index, g.edge_attr = connect_knn(pos, k, period=period)
    g.edge_attr = true_divide_by(g.edge_attr, 2 * scale[0])
    g.coarse_mask2 = guillard_coarsening(g.edge_index, n)
    ci2 = g.coarse_mask2.nonzero().reshape(-1)
    ei2, ea2 = connect_knn(pos[ci2], k, period=period)
    ea2 = true_divide_by(ea2, 2 * scale[1])
    m3 = torch.zeros(n, dtype=torch.bool, device=dev)
    m3[g.coarse_mask2] = guillard_coarsening(ei2, ci2.numel())
    g.coarse_mask3 = m3
    ci3 = m3.nonzero().reshape(-1)
    ei3, ea3 = connect_knn(pos[ci3], k, period=period)
    ea3 = true_divide_by(ea3, 2 * scale[2])
    g.edge_index2, g.edge_attr2 = ci2[ei2], ea2
    g.edge_index3, g.edge_attr3 = ci3[ei3], ea3
    for s, cnt in (("", n), ("2", ci2.numel()), ("3", ci3.numel())):
        u, ai, aa = extend_graph(getattr(g, f"edge_index{s}"), getattr(g, f"edge_attr{s}"), k)
        setattr(g, f"edgeUnitVector{s}", u)
        setattr(g, f"angle_index{s}", ai)
        setattr(g, f"angle_attr{s}", aa)
        # (closed form on the device; on the host the reference's own SVD call)
        setattr(g, f"edgeUnitVectorInverse{s}", pinv_k2(u.reshape(cnt, -1, 2)) if u.is_cuda else torch.linalg.pinv(u.reshape(cnt, -1, 2)))
    g.angle_index12, g.angle_attr12 = angle_index_down(g.edge_index, g.edge_attr, g.edge_index2, g.edge_attr2, ci2, k)
    g.angle_index23, g.angle_attr23 = angle_index_down(g.edge_index2, g.edge_attr2, g.edge_index3, g.edge_attr3, ci3, k)
    g.y_idx_21, g.x_idx_21, g.weights_21 = knn_interp_weights(pos[ci2], pos, k)
    g.y_idx_32, g.x_idx_32, g.weights_32 = knn_interp_weights(pos[ci3], pos[ci2], k)
    g.field = torch.randn(n, 2, generator=gen).to(dev)
    g.glob = torch.rand(n, 1, generator=gen).to(dev)
    g.omega = (torch.rand(n, 1, generator=gen) > 0.9).float().to(dev)
    return g


# ------------------------------------------------------------------------------ arch dicts
MUS_LAYERS: Dict[str, str] = {
    "NsOneScaleGNN": "mp11 mp12 mp13 mp14 mp15 mp16 mp17 mp18",
    "NsTwoScaleGNN": "mp111 mp112 mp113 mp114 down_mp12 mp21 mp22 mp23 mp24 up_mp21 mp121 mp122 mp123 mp124",
    "NsThreeScaleGNN": "mp111 mp112 mp113 mp114 down_mp12 mp211 mp212 down_mp23 mp31 mp32 mp33 mp34 up_mp32 mp221 mp222 "
                       "up_mp21 mp121 mp122 mp123 mp124",
    "NsFourScaleGNN": "mp111 mp112 mp113 mp114 down_mp12 mp211 mp212 down_mp23 mp311 mp312 down_mp34 mp41 mp42 mp43 mp44 "
                      "up_mp43 mp321 mp322 up_mp32 mp221 mp222 up_mp21 mp121 mp122 mp123 mp124",
    "AdvOneScaleGNN": "mp111 mp112 mp121 mp122",
    "AdvTwoScaleGNN": "mp111 mp112 down_mp12 mp21 mp22 mp23 mp24 up_mp21 mp121 mp122",
    "AdvThreeScaleGNN": "mp111 mp112 down_mp12 mp211 mp212 down_mp23 mp31 mp32 mp33 mp34 up_mp32 mp221 mp222 up_mp21 "
                        "mp121 mp122",
    "AdvFourScaleGNN": "mp111 mp112 down_mp12 mp211 mp212 down_mp23 mp311 mp312 down_mp34 mp41 mp42 mp43 mp44 up_mp43 "
                       "mp321 mp322 up_mp32 mp221 mp222 up_mp21 mp121 mp122",
}


def mus_arch(model: str, hidden: int = 128, nf: int = 3, node_in: int = 5, dim: int = 2) -> dict:
    """The arch dict published in the docstring of each reference class (e.g. nn/mus_gnn.py:105-129)
    with latent width `hidden` (128 in every published model)."""
    H = hidden
    mp = ((H + 2 * H, (H, H, H), True), (H + H, (H, H, H), True))
    arch = {"edge_encoder": (dim, (H, H, H), False), "node_encoder": (node_in, (H, H, H), False)}
    for name in MUS_LAYERS[model].split():
        arch[name] = (dim + H, (H, H, H), True) if name.startswith("down") else \
            (dim + H + H, (H, H, H), True) if name.startswith("up") else mp
    arch["decoder"] = (H, (H, H, nf), False)
    return arch


def remus_arch(hidden: int = 128) -> dict:
    """Arch of NsRotEquiTreeScaleGNN (nn/remus_gnn.py:16-58)."""
    H = hidden
    mp = ((H + 2 * H, (H, H), True), (H + H, (H, H), True))
    arch = {}
    for n in ("angle_encoder", "angle_encoder12", "angle_encoder2", "angle_encoder23", "angle_encoder3"):
        arch[n] = (4, (H, H), True)
    for n in ("edge_encoder", "edge_encoder2", "edge_encoder3"):
        arch[n] = (3, (H, H), True)
    for n in ("mp111 mp112 mp113 mp114 down_mp12 mp211 mp212 down_mp23 mp31 mp32 mp33 mp34 mp221 mp222 "
              "mp121 mp122 mp123 mp124").split():
        arch[n] = mp
    arch["up_mp32"] = (H + H, (H, H, H), True)
    arch["up_mp21"] = (H + H, (H, H, H), True)
    arch["decoder"] = (H, (H, 1), False)
    return arch

"""Node-partitioned multi-GPU execution of REMuS-GNN (NsRotEquiTreeScaleGNN): the companion of partition.py for the model
whose latents live on EDGES and ANGLES (new functionality: the reference is single-device).

The mesh nodes are cut into `world` compact parts (recursive coordinate bisection over the level-1 nodes; the coarse levels
of a REMuS graph are subsets of the same nodes and inherit their owner).  From that single table:

  * an EDGE j->i of any level is owned by the rank of its target node i, an ANGLE (k->j, j->i) by the rank of its target edge:
    the angle aggregation onto edges, the per-node least-squares of `edgeScalarToNodeVector` (all k incoming edges of a node),
    the projection of node vectors onto edges (`col` only) and every encoder / decoder are rank-local;
  * what moves is the EDGE-LATENT HALO: before an EdgeMP (or DownEdgeMP) of level l each rank receives the latents of the
    level-l edges k->j that its angles read but that end in a node j owned elsewhere (k edges per halo node).  Halo rows live
    behind the owned rows of the same tensor ([E_own + E_halo, H]), grouped by owner rank: one `all_to_all_single` lands in place
    (partition.HaloExchanger, channels 1..3);
  * UpEdgeMP interpolates node vectors of the coarse level's nodes onto the fine level's nodes: the coarse node vectors of
    the k nearest coarse nodes that are owned elsewhere form a second, small halo ([n_own + n_halo, 2H], channels 4..5).

`RemusPartitionedForward` interprets NsRotEquiTreeScaleGNN._PROGRAM on the local sub-mesh; the arithmetic is delegated to an
`impl` object: `RemusHipImpl` (the HIP kernels, product path) or, in tests only, an oracle-backed implementation that exercises
the partition / halo logic on CPU + gloo.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, ops, plan
from .graph import Graph
from .ops import Source
from .partition import _rcb

SELU, NONE = _lib.ACT_SELU, _lib.ACT_NONE
SFX = {1: "", 2: "2", 3: "3"}
LEVELS = 3
CH_NODE = {2: 4, 3: 5}          # exchanger channel of the node-vector halo of coarse level `lo`


# ------------------------------------------------------------------------------------- host-side partitioner
@dataclass
class RemusLevel:
    edge_ids: np.ndarray             # global ids of owned edges, ascending (grouped by target node, k per node)
    halo_edges: np.ndarray           # global ids of halo edges, grouped by owner rank, ascending inside a group
    halo_owner: np.ndarray
    angle_ids: np.ndarray            # global ids of owned angles of this level, ascending
    angle_index: np.ndarray          # [2, A_own] in LOCAL edge ids (row: own or halo, col: own)
    nodes: np.ndarray                # global (level-1 numbering) ids of the owned nodes of this level, ascending
    down_angle_ids: Optional[np.ndarray] = None     # owned angles level -> level + 1 (target edge of level + 1 owned)
    down_angle_index: Optional[np.ndarray] = None   # [2, A] row: LOCAL edge id of this level, col: LOCAL edge id of level + 1

    @property
    def n_own(self) -> int:
        return int(self.edge_ids.shape[0])

    @property
    def n_halo(self) -> int:
        return int(self.halo_edges.shape[0])


@dataclass
class RemusInterp:
    """knn interpolation coarse level `lo` -> level lo - 1, restricted to the owned fine nodes."""
    halo_nodes: np.ndarray           # global ids of coarse nodes read but owned elsewhere, grouped by owner
    halo_owner: np.ndarray
    y_idx: np.ndarray                # position of the fine node among the owned nodes of level lo - 1 (sorted)
    x_idx: np.ndarray                # LOCAL coarse node id (owned coarse nodes first, then halo)
    entry_ids: np.ndarray            # positions in the Graph's (y_idx, x_idx, weights) lists


@dataclass
class RemusPart:
    rank: int
    nodes: np.ndarray                # owned level-1 nodes, ascending
    levels: List[RemusLevel] = field(default_factory=list)
    interp: Dict[int, RemusInterp] = field(default_factory=dict)     # keyed by the coarse level lo (2, 3)
    # exchanger channels 1..5: per peer local row ids to send, rows received
    send_idx: Dict[int, List[np.ndarray]] = field(default_factory=dict)
    recv_counts: Dict[int, List[int]] = field(default_factory=dict)


def remus_owners(graph: Graph, world: int) -> np.ndarray:
    pos = graph.pos.cpu().numpy().astype(np.float64)
    owner = np.empty(pos.shape[0], dtype=np.int64)
    _rcb(pos, np.ones(pos.shape[0], dtype=np.int64), np.arange(pos.shape[0]), 0, world, owner)
    return owner


def _np(t) -> np.ndarray:
    return np.ascontiguousarray(t.cpu().numpy())


def _group_by_owner(ids: np.ndarray, owner_of: np.ndarray):
    ids = np.unique(ids)
    own = owner_of[ids]
    order = np.lexsort((ids, own))
    return ids[order], own[order]


def build_remus_partition(graph: Graph, world: int, owner: Optional[np.ndarray] = None) -> List[RemusPart]:
    """parts[rank] for every rank (the whole table is cheap, lets every rank derive its send lists without communication and
    lets tests check consistency)."""
    owner = remus_owners(graph, world) if owner is None else owner
    n = int(owner.shape[0])
    ei = {l: _np(getattr(graph, f"edge_index{SFX[l]}")).astype(np.int64) for l in range(1, LEVELS + 1)}
    ai = {l: _np(getattr(graph, f"angle_index{SFX[l]}")).astype(np.int64) for l in range(1, LEVELS + 1)}
    ad = {1: _np(graph.angle_index12).astype(np.int64), 2: _np(graph.angle_index23).astype(np.int64)}
    mask = {1: np.ones(n, dtype=bool), 2: _np(graph.coarse_mask2).astype(bool), 3: _np(graph.coarse_mask3).astype(bool)}
    e_owner = {l: owner[ei[l][1]] for l in ei}
    parts = [RemusPart(rank=r, nodes=np.nonzero(owner == r)[0]) for r in range(world)]
    g2l_e: Dict[tuple, np.ndarray] = {}
    for l in range(1, LEVELS + 1):
        n_e = int(ei[l].shape[1])
        a_owner = e_owner[l][ai[l][1]]
        d_owner = e_owner[l + 1][ad[l][1]] if l < LEVELS else None
        for r in range(world):
            own_e = np.nonzero(e_owner[l] == r)[0]
            own_a = np.nonzero(a_owner == r)[0]
            src = ai[l][0][own_a]
            cand = [src[e_owner[l][src] != r]]
            own_d = None
            if l < LEVELS:
                own_d = np.nonzero(d_owner == r)[0]
                s2 = ad[l][0][own_d]
                cand.append(s2[e_owner[l][s2] != r])
            halo, h_owner = _group_by_owner(np.concatenate(cand), e_owner[l])
            lut = np.full(n_e, -1, dtype=np.int64)
            lut[own_e] = np.arange(own_e.shape[0])
            lut[halo] = own_e.shape[0] + np.arange(halo.shape[0])
            g2l_e[(r, l)] = lut
            a_loc = np.stack([lut[ai[l][0][own_a]], lut[ai[l][1][own_a]]], 0)
            assert (a_loc >= 0).all() and (a_loc[1] < own_e.shape[0]).all()
            parts[r].levels.append(RemusLevel(edge_ids=own_e, halo_edges=halo, halo_owner=h_owner, angle_ids=own_a, angle_index=a_loc,
                                              nodes=np.nonzero(mask[l] & (owner == r))[0], down_angle_ids=own_d))
    for l in range(1, LEVELS):                       # down angles: local ids of both levels are known now
        for r in range(world):
            lv = parts[r].levels[l - 1]
            d = np.stack([g2l_e[(r, l)][ad[l][0][lv.down_angle_ids]], g2l_e[(r, l + 1)][ad[l][1][lv.down_angle_ids]]], 0)
            assert (d >= 0).all() and (d[1] < parts[r].levels[l].n_own).all()
            lv.down_angle_index = d
    # edge-halo send lists (channels 1..3): what q receives from r, in q's halo order
    for l in range(1, LEVELS + 1):
        for r in range(world):
            lut = g2l_e[(r, l)]
            send = []
            for q in range(world):
                lq = parts[q].levels[l - 1]
                need = lq.halo_edges[lq.halo_owner == r]
                send.append(lut[need])
                assert (send[-1] >= 0).all() and (send[-1] < parts[r].levels[l - 1].n_own).all()
            parts[r].send_idx[l] = send
            lr = parts[r].levels[l - 1]
            parts[r].recv_counts[l] = [int((lr.halo_owner == q).sum()) for q in range(world)]
    # interpolation lo -> lo - 1 and its node-vector halo (channels 4..5)
    for lo in (2, 3):
        hi = lo - 1
        y, x = _np(getattr(graph, f"y_idx_{lo}{hi}")).astype(np.int64), _np(getattr(graph, f"x_idx_{lo}{hi}")).astype(np.int64)
        nodes_hi, nodes_lo = np.nonzero(mask[hi])[0], np.nonzero(mask[lo])[0]
        y_owner, x_owner = owner[nodes_hi[y]], owner[nodes_lo[x]]
        luts = {}
        for r in range(world):
            ent = np.nonzero(y_owner == r)[0]
            xs = x[ent]
            halo_c, h_owner = _group_by_owner(xs[x_owner[ent] != r], owner[nodes_lo])      # compact coarse ids
            own_c = np.nonzero(owner[nodes_lo] == r)[0]
            lut = np.full(nodes_lo.shape[0], -1, dtype=np.int64)
            lut[own_c] = np.arange(own_c.shape[0])
            lut[halo_c] = own_c.shape[0] + np.arange(halo_c.shape[0])
            luts[r] = lut
            own_h = np.nonzero(owner[nodes_hi] == r)[0]
            lut_h = np.full(nodes_hi.shape[0], -1, dtype=np.int64)
            lut_h[own_h] = np.arange(own_h.shape[0])
            parts[r].interp[lo] = RemusInterp(halo_nodes=nodes_lo[halo_c], halo_owner=h_owner, y_idx=lut_h[y[ent]], x_idx=lut[xs], entry_ids=ent)
            assert (parts[r].interp[lo].x_idx >= 0).all() and (parts[r].interp[lo].y_idx >= 0).all()
        lut_n = np.full(n, -1, dtype=np.int64)
        lut_n[nodes_lo] = np.arange(nodes_lo.shape[0])
        for r in range(world):
            send = []
            for q in range(world):
                iq = parts[q].interp[lo]
                need = iq.halo_nodes[iq.halo_owner == r]
                send.append(luts[r][lut_n[need]])
                assert (send[-1] >= 0).all() and (send[-1] < parts[r].levels[lo - 1].nodes.shape[0]).all()
            parts[r].send_idx[CH_NODE[lo]] = send
            ir = parts[r].interp[lo]
            parts[r].recv_counts[CH_NODE[lo]] = [int((ir.halo_owner == q).sum()) for q in range(world)]
    return parts


# ------------------------------------------------------------------------------------- local sub-mesh on a device
class RemusLocalMesh:
    """Rank-local tensors of the partitioned REMuS Graph, in local numbering, on `device`.  Exposes the channel tables
    partition.HaloExchanger reads (`n_own`, `send_idx32`, `send_counts`, `recv_counts`, indexed by channel - 1)."""

    def __init__(self, graph: Graph, part: RemusPart, device: torch.device, rank: int, world: int):
        self.device, self.rank, self.world, self.part = device, rank, world, part
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)   # noqa: E731
        sel = lambda x, ids: x[torch.from_numpy(ids)].contiguous().to(device)   # noqa: E731
        n = int(graph.pos.size(0))
        self.n_nodes = int(part.nodes.shape[0])
        self.owned_global = [t(part.nodes)]
        self.inputs = {k: sel(getattr(graph, k), part.nodes) for k in ("field", "glob", "omega")}
        g2l_n = np.full(n, -1, dtype=np.int64)
        g2l_n[part.nodes] = np.arange(self.n_nodes)
        self.k = int(graph.edgeUnitVectorInverse.size(2))
        self.n_edges, self.n_halo_edges, self.col32, self.unit, self.unit_inv = {}, {}, {}, {}, {}
        self.angle_index, self.angle_attr, self.n_level_nodes, self.level_node32 = {}, {}, {}, {}
        self.down_index, self.down_attr = {}, {}
        mask = {1: np.ones(n, dtype=bool), 2: _np(graph.coarse_mask2).astype(bool), 3: _np(graph.coarse_mask3).astype(bool)}
        for l in range(1, LEVELS + 1):
            lv, s = part.levels[l - 1], SFX[l]
            ei = _np(getattr(graph, f"edge_index{s}")).astype(np.int64)
            col = ei[1][lv.edge_ids]
            # owned edges ascending = grouped by owned node of this level in ascending order, k per node (edgeScalarToNodeVector)
            assert lv.n_own == self.k * lv.nodes.shape[0] and np.array_equal(col.reshape(-1, self.k), np.repeat(lv.nodes[:, None], self.k, 1)), \
                "REMuS edges must be grouped by target node with constant in-degree"
            self.n_edges[l], self.n_halo_edges[l] = lv.n_own, lv.n_halo
            self.col32[l] = t(g2l_n[col].astype(np.int32))
            self.unit[l] = sel(getattr(graph, f"edgeUnitVector{s}"), lv.edge_ids)
            compact = np.cumsum(mask[l]) - 1
            self.unit_inv[l] = sel(getattr(graph, f"edgeUnitVectorInverse{s}"), compact[lv.nodes])
            self.angle_index[l] = t(lv.angle_index)
            self.angle_attr[l] = sel(getattr(graph, f"angle_attr{s}"), lv.angle_ids)
            self.n_level_nodes[l] = int(lv.nodes.shape[0])
            self.level_node32[l] = t(g2l_n[lv.nodes].astype(np.int32))      # local level-1 id of the owned nodes of level l
            if l < LEVELS:
                self.down_index[l] = t(lv.down_angle_index)
                self.down_attr[l] = sel(getattr(graph, f"angle_attr{l}{l + 1}"), lv.down_angle_ids)
        self.interp_y, self.interp_x32, self.interp_w, self.n_halo_nodes = {}, {}, {}, {}
        for lo in (2, 3):
            it = part.interp[lo]
            self.interp_y[lo] = t(it.y_idx)
            self.interp_x32[lo] = t(it.x_idx.astype(np.int32))
            self.interp_w[lo] = sel(getattr(graph, f"weights_{lo}{lo - 1}").reshape(-1), it.entry_ids)
            self.n_halo_nodes[lo] = int(it.halo_nodes.shape[0])
        # channel tables for partition.HaloExchanger (channel c at index c - 1)
        chans = [1, 2, 3, 4, 5]
        own_rows = {1: self.n_edges[1], 2: self.n_edges[2], 3: self.n_edges[3], 4: self.n_level_nodes[2], 5: self.n_level_nodes[3]}
        self.n_own = [own_rows[c] for c in chans]
        self.n_halo = [self.n_halo_edges[1], self.n_halo_edges[2], self.n_halo_edges[3], self.n_halo_nodes[2], self.n_halo_nodes[3]]
        self.send_idx32 = [[t(ix.astype(np.int32)) for ix in part.send_idx[c]] for c in chans]
        self.send_counts = [[int(ix.shape[0]) for ix in part.send_idx[c]] for c in chans]
        self.recv_counts = [list(part.recv_counts[c]) for c in chans]


# ------------------------------------------------------------------------------------- compute back-end (HIP)
class RemusHipImpl:
    """Arithmetic of the partitioned REMuS forward on the HIP kernels (the product path)."""

    def __init__(self, model, mesh: RemusLocalMesh):
        self.m, self.mesh = model, mesh
        self.width = int(model.edge_encoder.output_size)
        self._bufs: Dict[tuple, List[torch.Tensor]] = {}
        self._turn: Dict[tuple, int] = {}

    def buf(self, kind: str, lvl: int) -> torch.Tensor:
        """Rotating pair of [own + halo, width] buffers per (kind, level): a launch never writes the tensor it reads."""
        mesh, key = self.mesh, (kind, lvl)
        if key not in self._bufs:
            rows = mesh.n_edges[lvl] + mesh.n_halo_edges[lvl] if kind == "e" else mesh.n_level_nodes[lvl] + mesh.n_halo_nodes[lvl]
            width = self.width if kind == "e" else 2 * self.width
            self._bufs[key] = [torch.zeros((rows, width), dtype=torch.float32, device=mesh.device) for _ in range(2 if kind == "e" else 1)]
            self._turn[key] = 0
        pair = self._bufs[key]
        self._turn[key] = (self._turn[key] + 1) % len(pair)
        return pair[self._turn[key]]

    def encode(self):
        m, mesh = self.m, self.mesh
        nfeat = int(mesh.inputs["field"].size(1)) // 2
        e, a = {}, {}
        for l in range(1, LEVELS + 1):
            s = SFX[l]
            proj = ops.project_to_edges(mesh.inputs["field"], mesh.col32[l], mesh.unit[l], mesh.n_edges[l], nfeat)
            e[l] = self.buf("e", l)
            getattr(m, f"edge_encoder{s}").run_coded([Source(proj), Source(mesh.inputs["glob"], mesh.col32[l]), Source(mesh.inputs["omega"], mesh.col32[l])],
                                                     mesh.n_edges[l], SELU, out=e[l][: mesh.n_edges[l]])
            a[l] = self._angle_latents(f"angle_encoder{s}", mesh.angle_attr[l])
        ax = {1: self._angle_latents("angle_encoder12", mesh.down_attr[1]), 2: self._angle_latents("angle_encoder23", mesh.down_attr[2])}
        return e, a, ax

    def _angle_latents(self, name: str, att: torch.Tensor) -> torch.Tensor:
        """(static inside a rollout: ops.StaticCache, as NsRotEquiTreeScaleGNN._angle_latents)"""
        enc = getattr(self.m, name)
        return ops.static_launch(name, [att], lambda: enc.run_coded([Source(att)], int(att.size(0)), SELU))

    def mp(self, name: str, e: torch.Tensor, a: torch.Tensor, a_pending: int, lvl: int):
        from .nn.blocks import _mp_step
        blk, n_own = getattr(self.m, name), self.mesh.n_edges[lvl]
        out = self.buf("e", lvl)
        _, a_new = _mp_step(blk.angle_mlp, blk.edge_mlp, e, a, self.mesh.angle_index[lvl], blk.aggr, SELU, a_pending,
                            n_targets=n_own, v_out=out[:n_own], compact_messages=True)
        return out, a_new

    def down(self, name: str, e_lo: torch.Tensor, e_hi: torch.Tensor, a_x: torch.Tensor, lvl: int):
        from .nn.blocks import _mp_step
        blk, n_own = getattr(self.m, name), self.mesh.n_edges[lvl + 1]
        out = self.buf("e", lvl + 1)
        _mp_step(blk.angle_mlp, blk.edge_mlp, e_hi, a_x, self.mesh.down_index[lvl], "mean", SELU, v_src=e_lo,
                 n_targets=n_own, v_out=out[:n_own])
        return out

    def node_vectors(self, e_lo: torch.Tensor, lo: int) -> torch.Tensor:
        mesh = self.mesh
        nb = self.buf("n", lo)
        n_own = mesh.n_level_nodes[lo]
        if n_own:
            ops.edge_scalar_to_node_vector(e_lo[: mesh.n_edges[lo]], mesh.unit_inv[lo], n_own, mesh.k, out=nb[:n_own])
        return nb

    def up(self, name: str, nodebuf: torch.Tensor, e_hi: torch.Tensor, lo: int):
        mesh, hi = self.mesh, lo - 1
        nfeat = int(nodebuf.size(1)) // 2
        v1 = torch.zeros((mesh.n_nodes, 2 * nfeat), dtype=torch.float32, device=mesh.device) if hi > 1 else \
            torch.empty((mesh.n_nodes, 2 * nfeat), dtype=torch.float32, device=mesh.device)
        csr = plan.segments_of_sorted(mesh.interp_y[lo], mesh.n_level_nodes[hi])
        ops.weighted_segment_mean(nodebuf, mesh.interp_x32[lo], mesh.interp_w[lo], csr, v1, mesh.level_node32[hi] if hi > 1 else None)
        e1 = ops.project_to_edges(v1, mesh.col32[hi], mesh.unit[hi], mesh.n_edges[hi], nfeat)
        out = self.buf("e", hi)
        getattr(self.m, name).up_mlp.run_coded([Source(e1), Source(e_hi)], mesh.n_edges[hi], SELU, out=out[: mesh.n_edges[hi]])
        return out

    def decode(self, e1: torch.Tensor) -> torch.Tensor:
        mesh = self.mesh
        s = self.m.edge_decoder.run_coded([Source(e1)], mesh.n_edges[1], NONE)
        out = ops.edge_scalar_to_node_vector(s, mesh.unit_inv[1], mesh.n_nodes, mesh.k)
        res = torch.empty_like(out)
        f = mesh.inputs["field"]
        ops.add_cols(f, int(f.size(1)) - 2, out, res)
        return res


class RemusPartitionedForward:
    """NsRotEquiTreeScaleGNN.forward (nn/remus_gnn.py:119-199) on one rank's sub-mesh; returns the prediction of the owned nodes."""

    def __init__(self, program, mesh: RemusLocalMesh, impl, exchanger):
        self.program, self.mesh, self.impl, self.xch = program, mesh, impl, exchanger

    def forward(self) -> torch.Tensor:
        impl, x = self.impl, self.xch
        e, a, ax = impl.encode()
        fresh = {l: False for l in e}            # whether the halo rows of e[l] hold the peers' current latents
        a_pending = {l: NONE for l in e}

        def need_halo(l):
            if not fresh[l]:
                x.exchange(e[l], l)
                fresh[l] = True

        for op, name, lvl in self.program:
            if op == "mp":
                need_halo(lvl)
                e[lvl], a[lvl] = impl.mp(name, e[lvl], a[lvl], a_pending[lvl], lvl)
                from .nn.blocks import pending_act
                a_pending[lvl], fresh[lvl] = pending_act(a[lvl]), False
            elif op == "down":
                need_halo(lvl)
                e[lvl + 1] = impl.down(name, e[lvl], e[lvl + 1], ax[lvl], lvl)
                fresh[lvl + 1] = False
            else:
                nb = impl.node_vectors(e[lvl], lvl)
                x.exchange(nb, CH_NODE[lvl])
                e[lvl - 1] = impl.up(name, nb, e[lvl - 1], lvl)
                fresh[lvl - 1] = False
        return impl.decode(e[1])

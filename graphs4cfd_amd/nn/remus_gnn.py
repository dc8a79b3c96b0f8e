"""REMuS-GNN (rotation-equivariant multi-scale GNN), three-scale model.

Same class name, constructor, arch keys, submodule names and forward contract as the reference's
graphs4cfd/nn/remus_gnn.py (NsRotEquiTreeScaleGNN, :11-199).  The V-cycle over (edges, angles) is
interpreted from a table and runs on the fused HIP blocks (EdgeMP = the GNBlock kernels with edges
in the role of nodes and angles in the role of edges).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _lib, ops, plan
from ..graph import Graph
from ..ops import Source
from . import blocks as _blocks
from .blocks import MLP, EdgeMP, DownEdgeMP, UpEdgeMP, edgeScalarToNodeVector
from .model import GNN

SELU, NONE = _lib.ACT_SELU, _lib.ACT_NONE


# Round 6: the hoisted first-layer products of the FIRST EdgeMP of a run on a level (after the edge encoder, a DownEdgeMP, an UpEdgeMP)
# come out of that producer's own launch as two heads, like the products between consecutive EdgeMPs — instead of two product launches
# per entry (config 3: 10 launches, 0.49 ms of an 8.3 ms step).
ENTRY_PRODUCTS = __import__("os").environ.get("G4C_ENTRY_PRODUCTS", "1") != "0"
# Round 6, rounded-bf16 mode: the (static) angle latents of a DownEdgeMP regrouped by receiver once, so that its angle launch runs on the
# row-split kernel with the aggregation fused and no rows stored, instead of mlp_ws_kernel + a segment_reduce over the stored rows.
GROUP_DOWN_ANGLES = __import__("os").environ.get("G4C_GROUP_DOWN_ANGLES", "1") != "0"
# ... and the edge latents a run of EdgeMPs leaves behind are stored compact too (blocks.COMPACT_LATENTS) where only MLP launches read them
# until the level's next run (a DownEdgeMP from the level, an UpEdgeMP into it, the decoder) — not where an UpEdgeMP projects them in fp32.
LAST_COMPACT = __import__("os").environ.get("G4C_LAST_COMPACT", "1") != "0"


class NsRotEquiTreeScaleGNN(GNN):
    """The three-scale REMuS-GNN for incompressible flow inference from Lino et al. (2022)
    (https://doi.org/10.1063/5.0097679); reference: nn/remus_gnn.py:11-199.

    `arch` keys: angle_encoder{,12,2,23,3}, edge_encoder{,2,3}, mp111..mp114, down_mp12, mp211, mp212,
    down_mp23, mp31..mp34, up_mp32, mp221, mp222, up_mp21, mp121..mp124, decoder.

    Args:
        model (str, optional): Name of the pretrained model to load ("RE3S-GNN-NsEllipse-v1").
    """

    _PRETRAINED = {"RE3S-GNN-NsEllipse-v1": "weights/NsREMuSGNN/NsRotEquiThreeScaleGNN.chk"}
    _ENCODERS = ("angle_encoder", "angle_encoder12", "angle_encoder2", "angle_encoder23", "angle_encoder3",
                 "edge_encoder", "edge_encoder2", "edge_encoder3")
    # (op, module, level)
    _PROGRAM = (("mp", "mp111", 1), ("mp", "mp112", 1), ("mp", "mp113", 1), ("mp", "mp114", 1),
                ("down", "down_mp12", 1),
                ("mp", "mp211", 2), ("mp", "mp212", 2),
                ("down", "down_mp23", 2),
                ("mp", "mp31", 3), ("mp", "mp32", 3), ("mp", "mp33", 3), ("mp", "mp34", 3),
                ("up", "up_mp32", 3),
                ("mp", "mp221", 2), ("mp", "mp222", 2),
                ("up", "up_mp21", 2),
                ("mp", "mp121", 1), ("mp", "mp122", 1), ("mp", "mp123", 1), ("mp", "mp124", 1))

    def __init__(self, model: str = None, *args, **kwargs) -> None:
        if model is not None:
            super().__init__(arch=None, weights=None, checkpoint=self._pretrained(self._PRETRAINED, model), *args, **kwargs)
        else:
            super().__init__(*args, **kwargs)
        self.num_fields = 2

    def load_arch(self, arch: dict):
        self.arch = arch
        for name in self._ENCODERS:
            setattr(self, name, MLP(*arch[name]))
        for op, name, _ in self._PROGRAM:
            if op == "mp":
                setattr(self, name, EdgeMP(*arch[name]))
            elif op == "down":
                setattr(self, name, DownEdgeMP(*arch[name]))
            else:
                setattr(self, name, UpEdgeMP(arch[name]))
        self.edge_decoder = MLP(*arch["decoder"])
        self.to(self.device)

    def _angle_latents(self, name: str, att: torch.Tensor) -> torch.Tensor:
        enc = getattr(self, name)
        return ops.static_launch(name, [att], lambda: enc.run_coded([Source(att)], int(att.size(0)), SELU))

    def _compact_static_angles(self, lvl: int, name: str, att: torch.Tensor, a: torch.Tensor, angle_index: torch.Tensor, n_edges: int):
        """Inside a rollout, rounded-bf16 mode: when the level's first EdgeMP runs its angle launch on the row-split kernel
        (MLP.rs1_ready), the cached (static, already activated) angle latents are kept as the bf16 rows in that kernel's column order it
        would form from them on load — bit for bit the same operand, half the bytes of the step's one launch that read them as fp32."""
        if ops.StaticCache.active is None or torch.is_grad_enabled():
            return a
        first = next((getattr(self, nm) for o, nm, l in self._PROGRAM if o == "mp" and l == lvl), None)
        if first is None or not first.angle_mlp.rs1_ready(int(a.size(0)), plan.edge_csr(angle_index, n_edges)[1]):
            return a
        return ops.static_launch(name + "/rs16", [att], lambda: ops.RsOrderedRows.tag(
            a.to(torch.bfloat16)[:, ops._rs_k_order(a.device)].contiguous()))

    def _grouped_static_angles(self, name: str, att: torch.Tensor, a: torch.Tensor, index: torch.Tensor, n_targets: int, block):
        """(angle latents, angle index) of a DownEdgeMP with the angles grouped by receiver, when that makes its angle launch one the
        row-split kernel takes (uniform in-degree, rounded-bf16 mode, MLP.rs1_ready); unchanged otherwise.  The latents are static: inside a
        rollout the regrouped copy is cached with them."""
        if not GROUP_DOWN_ANGLES or ops.grad_mode() or ops.mlp_precision() != "bf16":
            return a, index
        grouped, perm = plan.grouped_by_target(index, n_targets)
        if perm is None or not block.angle_mlp.rs1_ready(int(a.size(0)), plan.edge_csr(grouped, n_targets)[1]):
            return a, index
        return ops.static_launch(name + "/grouped", [att], lambda: a.index_select(0, perm)), grouped

    @staticmethod
    def _mlp_readers_only(prog, k: int, lvl: int) -> bool:
        """Whether the edge latents of level `lvl` produced at program step k are read, until the level's next EdgeMP replaces them, by
        nothing but MLP launches (as an input block or through hoisted products): a DownEdgeMP from this level, an UpEdgeMP INTO this
        level (its skip input), the decoder.  An UpEdgeMP FROM this level projects them with fp32 arithmetic (edgeScalarToNodeVector):
        not compact then.  (blocks._mp_step compact_v: such latents may be stored as the bf16 rows their readers round them to.)"""
        if not LAST_COMPACT:
            return False
        for o, _, l in prog[k + 1:]:
            if o == "mp" and l == lvl:
                return True
            if o == "up" and l == lvl:          # (op level = the coarse side: this level is being projected)
                return False
        return True

    def _entry_of(self, prog, k: int, lvl: int, aidx, e):
        """(angle MLP, (angle rows, receiver CSR)) of the EdgeMP that follows program step k on level `lvl` — whose hoisted first-layer
        products the inter-level block at step k can emit with its own edge launch — or (None, None)."""
        nxt = prog[k + 1] if k + 1 < len(prog) else None
        if not ENTRY_PRODUCTS or nxt is None or nxt[0] != "mp" or nxt[2] != lvl or ops.grad_mode():
            return None, None
        return getattr(self, nxt[1]).angle_mlp, (int(aidx[lvl].size(1)), plan.edge_csr(aidx[lvl], int(e[lvl].size(0)))[1])

    def forward(self, graph: Graph, t: Optional[int] = None) -> torch.Tensor:
        g = graph
        sfx = {1: "", 2: "2", 3: "3"}
        nfeat = int(g.field.size(1)) // 2
        e, a, aidx = {}, {}, {}
        entry_products = None
        for lvl, s in sfx.items():
            ep = plan.edge_plan(getattr(g, f"edge_index{s}"))
            # project the node vectors along the edges, then [proj | glob[col] | omega[col]] -> encoder
            proj = ops.project_to_edges(g.field, ep.col, getattr(g, f"edgeUnitVector{s}"), ep.n_edges, nfeat)
            enc_src = [Source(proj), Source(g.glob, ep.col), Source(g.omega, ep.col)]
            e[lvl] = None
            if ENTRY_PRODUCTS and lvl == 1 and self._PROGRAM[0][0] == "mp" and not ops.grad_mode():
                # the level's first EdgeMP reads e next: its hoisted first-layer products come out of the encoder's launch
                first = getattr(self, self._PROGRAM[0][1]).angle_mlp
                n_ang = int(getattr(g, f"angle_index{s}").size(1))
                if n_ang >= _blocks.HOIST_MIN_ROWS:
                    enc = getattr(self, f"edge_encoder{s}")
                    out16 = (torch.empty((ep.n_edges, 128), dtype=torch.bfloat16, device=proj.device)
                             if _blocks.compact_latents_now(enc.output_size) else None)       # (e is read by that EdgeMP's update MLP only)
                    got = enc.run_with_heads(enc_src, ep.n_edges, SELU, first, first.input_size - 2 * enc.output_size,
                                             [enc.output_size] * 2, out=out16, rs_rows=first.rs1_ready(n_ang, plan.edge_csr(getattr(g, f"angle_index{s}"), ep.n_edges)[1]))
                    if got is not None:
                        e[lvl], entry_products = got
            if e[lvl] is None:
                e[lvl] = getattr(self, f"edge_encoder{s}").run_coded(enc_src, ep.n_edges, SELU)
            # (the angle attributes are static inside a rollout — nn/model.py:316-320 replaces graph.field only — so a Rollout
            # runs the five angle encoders once per mesh and weights, ops.StaticCache; a bare forward() launches them, like the
            # reference nn/remus_gnn.py:136-140)
            a[lvl] = self._angle_latents(f"angle_encoder{s}", getattr(g, f"angle_attr{s}"))
            aidx[lvl] = getattr(g, f"angle_index{s}")
            a[lvl] = self._compact_static_angles(lvl, f"angle_encoder{s}", getattr(g, f"angle_attr{s}"), a[lvl], aidx[lvl], ep.n_edges)
        a12 = self._angle_latents("angle_encoder12", g.angle_attr12)
        a23 = self._angle_latents("angle_encoder23", g.angle_attr23)
        # (the inter-level angles come in no receiver order; where the DownEdgeMP's angle launch can run on the row-split kernel their
        # — static — latents and the index are regrouped by receiver once: the launch then reduces its own rows and stores none)
        a12, idx12 = self._grouped_static_angles("angle_encoder12", g.angle_attr12, a12, g.angle_index12, int(e[2].size(0)), self.down_mp12)
        a23, idx23 = self._grouped_static_angles("angle_encoder23", g.angle_attr23, a23, g.angle_index23, int(e[3].size(0)), self.down_mp23)
        a_pending = {1: NONE, 2: NONE, 3: NONE}
        products = {1: entry_products, 2: None, 3: None}   # first-layer edge-side terms of the next EdgeMP of a level, if already made
        prog = self._PROGRAM
        for k, (op, name, lvl) in enumerate(prog):
            block = getattr(self, name)
            if op == "mp":
                nxt = prog[k + 1] if k + 1 < len(prog) else None
                if nxt is not None and nxt[0] == "mp" and nxt[2] == lvl:   # same level, consecutive: products ride along
                    e[lvl], a[lvl], products[lvl] = block.step(e[lvl], a[lvl], aidx[lvl], SELU, a_pre_act=a_pending[lvl],
                                                               products=products[lvl], next_msg=getattr(self, nxt[1]).angle_mlp,
                                                               compact_e=True)     # (e' of this EdgeMP is read by the next one only)
                else:
                    # the angle latents of a level are read again by its next EdgeMP — on the way up as well (a1 after mp114 feeds
                    # mp121, nn/remus_gnn.py:150-190); after the level's LAST EdgeMP of the step nothing reads them: not stored
                    # then, when the launch can reduce its own rows (_mp_step keep_e)
                    last_use = not any(o == "mp" and l == lvl for o, _, l in prog[k + 1:])
                    e[lvl], a[lvl] = block.step(e[lvl], a[lvl], aidx[lvl], SELU, a_pre_act=a_pending[lvl], products=products[lvl],
                                                keep_e=not last_use, compact_e=self._mlp_readers_only(prog, k, lvl))
                    products[lvl] = None
                a_pending[lvl] = _blocks.pending_act(a[lvl])       # (SELU; none when the rows came back compact and activated)
            elif op == "down":
                a_x, idx_x = (a12, idx12) if lvl == 1 else (a23, idx23)
                e[lvl + 1], products[lvl + 1] = block.step(e[lvl], e[lvl + 1], a_x, idx_x, "selu", *self._entry_of(prog, k, lvl + 1, aidx, e))
            else:
                lo, hi = lvl, lvl - 1
                nm, ng = self._entry_of(prog, k, hi, aidx, e)
                e[hi], products[hi] = block.step(g.pos, getattr(g, f"y_idx_{lo}{hi}"), getattr(g, f"x_idx_{lo}{hi}"),
                                                 getattr(g, f"weights_{lo}{hi}"), e[lo], getattr(g, f"edge_index{sfx[lo]}"),
                                                 getattr(g, f"edgeUnitVectorInverse{sfx[lo]}"), getattr(g, f"coarse_mask{sfx[lo]}"),
                                                 e[hi], getattr(g, f"edge_index{sfx[hi]}"), getattr(g, f"edgeUnitVector{sfx[hi]}"),
                                                 getattr(g, f"coarse_mask{sfx[hi]}") if hi > 1 else None, activation="selu",
                                                 next_msg=nm, next_graph=ng)
        s = self.edge_decoder.run_coded([Source(e[1])], int(e[1].size(0)), NONE)
        out = edgeScalarToNodeVector(s, g.edge_index, edgeUnitVectorInverse=g.edgeUnitVectorInverse)
        # time step: field[:, -2:] + output (nn/remus_gnn.py:199)
        return _add_last_fields(g.field, out, self.num_fields)


def _add_last_fields(field: torch.Tensor, out: torch.Tensor, nf: int) -> torch.Tensor:
    if ops.grad_mode() and out.requires_grad:
        return field[:, int(field.size(1)) - nf:] + out
    res = torch.empty_like(out)
    ops.add_cols(field, int(field.size(1)) - nf, out, res)
    return res

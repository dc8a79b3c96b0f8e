"""MuS-GNN model classes (1- to 4-scale, Navier-Stokes and advection variants).

Same class names, constructor, arch-dict keys, submodule attribute names (= checkpoint keys) and
`forward(graph, t=None)` contract as the reference's graphs4cfd/nn/mus_gnn.py.  The reference
hand-unrolls each V-cycle; here every class is a table (`_PROGRAM`) interpreted by one forward that
drives the fused HIP blocks:

  * encoders / decoder: one fused launch each (the node inputs `field|loc|glob|omega` are gathered as
    separate narrow column blocks, nn/mus_gnn.py:71 never materialises; the residual time step
    `field[:, -nf:] + output`, :97, is the decoder's epilogue),
  * every MP layer: edge MLP (gather + 3 layers + LayerNorm), CSR mean, node MLP with the model's
    `F.selu` fused; the SELU of the edge latents is deferred to their next reader; between consecutive MP layers
    of one level the node launch also emits the next edge MLP's node-side first-layer terms (blocks.MLP.run_with_heads),
  * DownMP/UpMP on the static plan with `tanh` fused.

The Graph is never mutated (the reference mutates and restores it, nn/mus_gnn.py:174,216).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _lib, ops
from ..graph import Graph
from ..ops import Source
from . import blocks as _blocks
from .blocks import MLP, MP, DownMP, UpMP
from .model import GNN

SELU, TANH, NONE = _lib.ACT_SELU, _lib.ACT_TANH, _lib.ACT_NONE


class _MuSGNN(GNN):
    _PROGRAM: tuple = ()
    _PRETRAINED: dict = {}

    def __init__(self, model: str = None, *args, **kwargs) -> None:
        if model is not None:
            super().__init__(arch=None, weights=None, checkpoint=self._pretrained(self._PRETRAINED, model), *args, **kwargs)
        else:
            super().__init__(*args, **kwargs)

    def load_arch(self, arch: dict):
        self.arch = arch
        self.edge_encoder = MLP(*arch["edge_encoder"])
        self.node_encoder = MLP(*arch["node_encoder"])
        for name in self._PROGRAM:
            if name.startswith("down_mp"):
                setattr(self, name, DownMP(arch[name], int(name[-2])))
            elif name.startswith("up_mp"):
                setattr(self, name, UpMP(arch[name], int(name[-2])))
            else:
                setattr(self, name, MP(*arch[name]))
        self.node_decoder = MLP(*arch["decoder"])
        self.to(self.device)

    def _launch_for(self, mlp: MLP, sources, n_rows: int, act_code: int, k: int, edge_index: torch.Tensor):
        """One launch of `mlp` whose output is the node input of program entry k.  When that entry is an MP layer
        that will hoist its first layer, the launch also emits its node-side products (MLP.run_with_heads).
        Returns (output, products or None)."""
        nxt = self._PROGRAM[k] if k < len(self._PROGRAM) else ""
        # (the consumer hoists from HOIST_MIN_ROWS edges on — and always when it runs as one fused launch per MP layer, whose message
        # part takes the products as additive rows: without them it would make them itself, two more launches at the level's entry)
        n_edges = int(edge_index.size(1))
        if nxt.startswith("mp") and (n_edges >= _blocks.HOIST_MIN_ROWS
                                     or _blocks.will_fuse_layer(getattr(self, nxt).edge_mlp, getattr(self, nxt).node_mlp, edge_index, n_rows)):
            cons = getattr(self, nxt).edge_mlp
            w = mlp.output_size
            res = mlp.run_with_heads(sources, n_rows, act_code, cons, cons.input_size - 2 * w, [w, w])
            if res is not None:
                return res
        return mlp.run_coded(sources, n_rows, act_code), None

    def forward(self, graph: Graph, t: Optional[int] = None) -> torch.Tensor:
        field0 = graph.field
        n = int(field0.size(0))
        inputs = [Source(getattr(graph, k)) for k in ('field', 'loc', 'glob', 'omega') if hasattr(graph, k)]
        edge_index = graph.edge_index
        # (edge_attr never changes inside a rollout — nn/model.py:316-320 replaces graph.field only — so a Rollout computes this
        # launch once per mesh and weights: ops.StaticCache; a bare forward() launches it every time, like the reference :73,178)
        e = ops.static_launch("edge_encoder", [graph.edge_attr],
                              lambda: self.edge_encoder.run_coded([Source(graph.edge_attr)], int(graph.edge_attr.size(0)), SELU))
        # `products`: first-layer node-side terms of the next MP layer, when the launch producing its `v` made them
        v, products = self._launch_for(self.node_encoder, inputs, n, SELU, 0, edge_index)
        e_pending = NONE          # activation not yet applied to `e` (deferred to its readers)
        stash = []
        prog = self._PROGRAM
        for k, name in enumerate(prog):
            block = getattr(self, name)
            if name.startswith("down_mp"):
                stash.append((v, edge_index, e, e_pending))
                v, edge_index, e = block.pool(graph, v, edge_index, e, torch.tanh, e_pre_act=e_pending, target_major=True)
                e_pending, products = NONE, None
            elif name.startswith("up_mp"):
                v_old, edge_index, e, e_pending = stash.pop()
                v, products = self._launch_for(block.up_mlp, block.sources(graph, v, v_old), int(v_old.size(0)), TANH, k + 1, edge_index)
            else:
                nxt = prog[k + 1] if k + 1 < len(prog) else ""
                if nxt.startswith("mp"):      # next MP layer runs on the same graph: its node-side products ride along
                    v, e, products = block.step(v, e, edge_index, SELU, e_pre_act=e_pending, products=products,
                                                next_msg=getattr(self, nxt).edge_mlp)
                else:
                    # the level's edge latents are dropped after this layer when an UpMP or the decoder follows (the up leg
                    # restores the latents stashed before the DownMP): they then need not be stored
                    drop_e = nxt.startswith("up_mp") or nxt == ""
                    v, e = block.step(v, e, edge_index, SELU, e_pre_act=e_pending, products=products, keep_e=not drop_e)
                    products = None
                e_pending = SELU
        nf = self.num_fields
        return self.node_decoder.run_coded([Source(v)], n, NONE, resid=field0, resid_col0=int(field0.size(1)) - nf)


def _model(name: str, program: str, pretrained: dict, doc: str):
    cls = type(name, (_MuSGNN,), {"_PROGRAM": tuple(program.split()), "_PRETRAINED": pretrained, "__doc__": doc})
    cls.__module__ = __name__
    return cls


_DOC = """The {n}-GNN for {what} inference from Lino et al. (2022) (https://doi.org/10.1063/5.0097679)
(reference: nn/mus_gnn.py:{lines}).  `arch` keys: edge_encoder, node_encoder, {keys}, decoder — each MP entry
`((in, widths, layer_norm), (in, widths, layer_norm))`, each down/up entry `(in, widths, layer_norm)`."""

NsOneScaleGNN = _model(
    "NsOneScaleGNN", "mp11 mp12 mp13 mp14 mp15 mp16 mp17 mp18",
    {"1S-GNN-NsCircle-v1": "weights/NsMuSGNN/NsOneScaleGNN.chk"},
    _DOC.format(n="1S", what="incompressible flow", lines="11-97", keys="mp11..mp18"))

NsTwoScaleGNN = _model(
    "NsTwoScaleGNN", "mp111 mp112 mp113 mp114 down_mp12 mp21 mp22 mp23 mp24 up_mp21 mp121 mp122 mp123 mp124",
    {"2S-GNN-NsCircle-v1": "weights/NsMuSGNN/NsTwoScaleGNN.chk"},
    _DOC.format(n="2S", what="incompressible flow", lines="100-218", keys="mp111..mp114, down_mp12, mp21..mp24, up_mp21, mp121..mp124"))

NsThreeScaleGNN = _model(
    "NsThreeScaleGNN",
    "mp111 mp112 mp113 mp114 down_mp12 mp211 mp212 down_mp23 mp31 mp32 mp33 mp34 up_mp32 mp221 mp222 up_mp21 "
    "mp121 mp122 mp123 mp124",
    {"3S-GNN-NsCircle-v1": "weights/NsMuSGNN/NsThreeScaleGNN.chk"},
    _DOC.format(n="3S", what="incompressible flow", lines="221-373", keys="mp111.., down_mp12, mp211, mp212, down_mp23, mp31..mp34, up_mp32, mp221, mp222, up_mp21, mp121..mp124"))

NsFourScaleGNN = _model(
    "NsFourScaleGNN",
    "mp111 mp112 mp113 mp114 down_mp12 mp211 mp212 down_mp23 mp311 mp312 down_mp34 mp41 mp42 mp43 mp44 up_mp43 "
    "mp321 mp322 up_mp32 mp221 mp222 up_mp21 mp121 mp122 mp123 mp124",
    {"4S-GNN-NsCircle-v1": "weights/NsMuSGNN/NsFourScaleGNN.chk"},
    _DOC.format(n="4S", what="incompressible flow", lines="376-562", keys="mp111.., down_mp12, mp211.., down_mp23, mp311.., down_mp34, mp41..mp44, up_mp43, mp321.., up_mp32, mp221.., up_mp21, mp121.."))

AdvOneScaleGNN = _model(
    "AdvOneScaleGNN", "mp111 mp112 mp121 mp122",
    {"1S-GNN-UniformAdv-v1": "weights/AdvMuSGNN/AdvOneScaleGNN.chk"},
    _DOC.format(n="1S", what="advection", lines="566-636", keys="mp111, mp112, mp121, mp122"))

AdvTwoScaleGNN = _model(
    "AdvTwoScaleGNN", "mp111 mp112 down_mp12 mp21 mp22 mp23 mp24 up_mp21 mp121 mp122",
    {"2S-GNN-UniformAdv-v1": "weights/AdvMuSGNN/AdvTwoScaleGNN.chk"},
    _DOC.format(n="2S", what="advection", lines="639-741", keys="mp111, mp112, down_mp12, mp21..mp24, up_mp21, mp121, mp122"))

AdvThreeScaleGNN = _model(
    "AdvThreeScaleGNN",
    "mp111 mp112 down_mp12 mp211 mp212 down_mp23 mp31 mp32 mp33 mp34 up_mp32 mp221 mp222 up_mp21 mp121 mp122",
    {"3S-GNN-UniformAdv-v1": "weights/AdvMuSGNN/AdvThreeScaleGNN.chk"},
    _DOC.format(n="3S", what="advection", lines="744-880", keys="mp111, mp112, down_mp12, mp211, mp212, down_mp23, mp31..mp34, up_mp32, mp221, mp222, up_mp21, mp121, mp122"))

AdvFourScaleGNN = _model(
    "AdvFourScaleGNN",
    "mp111 mp112 down_mp12 mp211 mp212 down_mp23 mp311 mp312 down_mp34 mp41 mp42 mp43 mp44 up_mp43 mp321 mp322 "
    "up_mp32 mp221 mp222 up_mp21 mp121 mp122",
    {"4S-GNN-UniformAdv-v1": "weights/AdvMuSGNN/AdvFourScaleGNN.chk"},
    _DOC.format(n="4S", what="advection", lines="883-1053", keys="mp111, mp112, down_mp12, ..., up_mp21, mp121, mp122"))

__all__ = ["NsOneScaleGNN", "NsTwoScaleGNN", "NsThreeScaleGNN", "NsFourScaleGNN",
           "AdvOneScaleGNN", "AdvTwoScaleGNN", "AdvThreeScaleGNN", "AdvFourScaleGNN"]

"""gMuS-GNN model classes (two to four scales, Guillard-coarsened low-resolution graphs).

Same class names, constructor, arch-dict keys, submodule attribute names (= checkpoint keys) and `forward(graph, t)`
contract as the reference's graphs4cfd/nn/mugs_gnn.py.  Like `mus_gnn.py`, every class is a table
interpreted by one forward over the fused HIP blocks:

  * down-sampling = row selection (`g4c_copy_cols` through the static `plan.restricted_level`) + the statically
    renumbered coarse `edge_index` (`restriction`, nn/blocks.py:9-32); the coarse edge latents are the encoded
    `edge_attr{l}`;
  * up-sampling = `knn_interpolate` (`g4c_weighted_segment_mean`) written straight into the left half of the
    [n, 2H] node-latent buffer of the next MP layer, the stashed fine latents copied into the right half
    (the reference's `torch.cat`, nn/mugs_gnn.py:115-116);
  * MP layers as in MuS-GNN (deferred SELU of the edge latents, heads between consecutive layers of a level); the 2H-wide
    latents after an up-sampling enter the MLPs as two 128-wide column chunks (blocks._split_wide; their first-layer products are
    hoisted at any size), so every launch stays on the split-operand kernels.

The Graph is never mutated.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _lib, ops, plan
from ..graph import Graph
from ..ops import Source
from . import blocks as _blocks
from .blocks import MLP, MP
from .model import GNN

SELU, NONE = _lib.ACT_SELU, _lib.ACT_NONE


class _MuGSGNN(GNN):
    _PROGRAM: tuple = ()
    _LEVELS: int = 2
    _PRETRAINED: dict = {}

    def __init__(self, model: str = None, *args, **kwargs) -> None:
        if model is not None:
            super().__init__(arch=None, weights=None, checkpoint=self._pretrained(self._PRETRAINED, model), *args, **kwargs)
        else:
            super().__init__(*args, **kwargs)

    def load_arch(self, arch: dict):
        self.arch = arch
        self.edge_encoder = MLP(*arch["edge_encoder"])
        for l in range(2, self._LEVELS + 1):
            setattr(self, f"edge_encoder{l}", MLP(*arch[f"edge_encoder{l}"]))
        self.node_encoder = MLP(*arch["node_encoder"])
        for name in self._PROGRAM:
            if name.startswith("mp"):
                setattr(self, name, MP(*arch[name]))
        self.node_decoder = MLP(*arch["decoder"])
        self.to(self.device)

    def forward(self, graph: Graph, t: Optional[int] = None) -> torch.Tensor:
        g = graph
        field0 = g.field
        n = int(field0.size(0))
        inputs = [Source(getattr(g, k)) for k in ('field', 'loc', 'glob', 'omega') if hasattr(g, k)]
        # encoders (nn/mugs_gnn.py:225-228): every level's edge latents up front
        e_enc = {1: self.edge_encoder.run_coded([Source(g.edge_attr)], int(g.edge_attr.size(0)), SELU)}
        for l in range(2, self._LEVELS + 1):
            ea = getattr(g, f"edge_attr{l}")
            e_enc[l] = getattr(self, f"edge_encoder{l}").run_coded([Source(ea)], int(ea.size(0)), SELU)
        v = self.node_encoder.run_coded(inputs, n, SELU)
        e, e_pending, edge_index, level = e_enc[1], NONE, g.edge_index, 1
        stash, products = {}, None
        prog = self._PROGRAM
        for k, name in enumerate(prog):
            if name.startswith("mp"):
                block = getattr(self, name)
                nxt = prog[k + 1] if k + 1 < len(prog) else ""
                if nxt.startswith("mp"):
                    v, e, products = block.step(v, e, edge_index, SELU, e_pre_act=e_pending, products=products,
                                                next_msg=getattr(self, nxt).edge_mlp)
                else:
                    v, e = block.step(v, e, edge_index, SELU, e_pre_act=e_pending, products=products)
                    products = None
                e_pending = SELU
            elif name.startswith("down"):
                l = int(name[4:])
                stash[level] = (v, edge_index, e, e_pending)
                keep32, edge_index = plan.restricted_level(getattr(g, f"coarse_mask{l}"), getattr(g, f"edge_index{l}"),
                                                           getattr(g, f"coarse_mask{level}") if level > 1 else None)
                if ops.grad_mode() and v.requires_grad:
                    from .. import autograd as _ag
                    v_c = _ag.gather_rows(v, keep32)
                else:
                    v_c = torch.empty((int(keep32.numel()), int(v.size(1))), dtype=torch.float32, device=v.device)
                    ops.copy_cols(v, v_c, 0, idx32=keep32)
                v, e, e_pending, level, products = v_c, e_enc[l], NONE, l, None
            else:                                     # "up{hi}{lo}"
                hi, lo = int(name[2]), int(name[3])
                v_old, edge_index, e, e_pending = stash[lo]
                H = int(v.size(1))
                if ops.grad_mode() and (v.requires_grad or v_old.requires_grad):
                    up = _blocks.knn_interpolate(v, getattr(g, f"y_idx_{hi}{lo}"), getattr(g, f"x_idx_{hi}{lo}"),
                                                 getattr(g, f"weights_{hi}{lo}"))
                    buf = torch.cat((up, v_old), 1)        # (recorded for autograd; inference writes both halves in place)
                else:
                    buf = torch.empty((int(v_old.size(0)), H + int(v_old.size(1))), dtype=torch.float32, device=v.device)
                    _blocks.knn_interpolate(v, getattr(g, f"y_idx_{hi}{lo}"), getattr(g, f"x_idx_{hi}{lo}"),
                                            getattr(g, f"weights_{hi}{lo}"), out=buf[:, :H])
                    ops.copy_cols(v_old, buf, H)
                v, level, products = buf, lo, None
        nf = self.num_fields
        return self.node_decoder.run_coded([Source(v)], n, NONE, resid=field0, resid_col0=int(field0.size(1)) - nf)


def _model(name: str, levels: int, program: str, pretrained: dict, lines: str):
    doc = (f"The {levels}-scale gMuS-GNN for incompressible flow from Lino et al. (2022) (https://doi.org/10.1063/5.0097679), "
           f"low-resolution graphs by Guillard's node-nested coarsening (reference: nn/mugs_gnn.py:{lines}).  `arch` keys: "
           f"edge_encoder, edge_encoder2.., node_encoder, the MP layers of the program `{program}` "
           f"(`((in, widths, layer_norm), (in, widths, layer_norm))` each), decoder.")
    cls = type(name, (_MuGSGNN,), {"_PROGRAM": tuple(program.split()), "_LEVELS": levels, "_PRETRAINED": pretrained, "__doc__": doc})
    cls.__module__ = __name__
    return cls


NsTwoGuillardScaleGNN = _model(
    "NsTwoGuillardScaleGNN", 2, "mp111 mp112 mp113 mp114 down2 mp21 mp22 mp23 mp24 up21 mp121 mp122 mp123 mp124",
    {"2GS-GNN-NsCircle-v1": "weights/NsMuGSGNN/NsTwoGuillardScaleGNN.chk"}, "11-132")

NsThreeGuillardScaleGNN = _model(
    "NsThreeGuillardScaleGNN", 3,
    "mp111 mp112 mp113 mp114 down2 mp211 mp212 down3 mp31 mp32 mp33 mp34 up32 mp221 mp222 up21 mp121 mp122 mp123 mp124",
    {"3GS-GNN-NsCircle-v1": "weights/NsMuGSGNN/NsThreeGuillardScaleGNN.chk"}, "135-294")

NsFourGuillardScaleGNN = _model(
    "NsFourGuillardScaleGNN", 4,
    "mp111 mp112 mp113 mp114 down2 mp211 mp212 down3 mp311 mp312 down4 mp41 mp42 mp43 mp44 up43 mp321 mp322 up32 mp221 mp222 up21 "
    "mp121 mp122 mp123 mp124",
    {"4GS-GNN-NsCircle-v1": "weights/NsMuGSGNN/NsFourGuillardScaleGNN.chk"}, "297-489")

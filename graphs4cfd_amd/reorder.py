"""Locality renumbering of a MuS-GNN mesh for a rollout (host-side plan, once per mesh; new functionality — the reference runs
the mesh as numbered).

The edge launch of an MP layer gathers, per edge, the sender's row of the hoisted first-layer products (`nn/blocks.py:181`:
`v[row]`).  Edges are grouped by target, so a tile's senders are the neighbours of a few consecutive targets: when consecutive
node numbers are neighbours in space those rows were touched by the tiles just before and the gather hits L2; on a randomly
numbered mesh (the synthetic benchmark mesh, any mesh after a shuffle) every one of them is a 512-byte miss — 307 MB of HBM reads per
level-1 layer at 100k nodes for 51 MB of distinct rows.  `reorder_nodes` numbers the level-1 nodes along a Morton (Z-order)
curve over `pos`, rewrites `edge_index` accordingly and re-sorts the edges by their new target (stably: the edges of a target keep
their relative order, so its aggregation adds the same values in the same order); `Rollout` runs on the renumbered Graph and maps
its output rows back.  The arithmetic per node and per edge is unchanged; only sums over a cluster / a coarse edge see their
terms in a different order (fp32 round-off).
"""
from __future__ import annotations

import re
from typing import Optional, Tuple

import torch

from .graph import Graph

_NODE = ("pos", "field", "loc", "glob", "omega", "target", "batch", "cluster_2", "idx1_to_idx2", "e_12")
_COARSE = re.compile(r"^(pos_\d+|mask_\d+|cluster_([3-9]|\d\d+)|idx([2-9]|\d\d+)_to_idx\d+|e_([2-9])\d)$")


def morton_order(pos: torch.Tensor, bits: int = 16) -> torch.Tensor:
    """Permutation (new -> old) that sorts points along a Z-order curve over their bounding box."""
    p = pos.detach().to(torch.float64)
    lo, hi = p.min(0).values, p.max(0).values
    q = ((p - lo) / (hi - lo).clamp(min=1e-300) * float(1 << bits)).to(torch.int64).clamp_(0, (1 << bits) - 1)
    dim = int(p.size(1))
    code = torch.zeros(p.size(0), dtype=torch.int64, device=p.device)
    for b in range(bits):
        for d in range(dim):
            code |= ((q[:, d] >> b) & 1) << (b * dim + d)
    return torch.argsort(code, stable=True)


def reorder_nodes(graph: Graph) -> Optional[Tuple[Graph, torch.Tensor]]:
    """(renumbered copy of `graph`, perm) with `new_row[k] = old_row[perm[k]]`, or None when the Graph carries an attribute whose
    indexing this function does not know (REMuS / Guillard graphs, user attributes): the rollout then runs on it as it is."""
    d = graph.to_dict()
    if "pos" not in d or "edge_index" not in d or not torch.is_tensor(d["pos"]) or d["pos"].dim() != 2 or d["pos"].size(1) not in (2, 3):
        return None
    n, n_e = int(d["pos"].size(0)), int(d["edge_index"].size(1))
    for k, v in d.items():
        if not torch.is_tensor(v):
            continue
        if k in _NODE:
            if int(v.size(0)) != n:
                return None
        elif k == "edge_attr":
            if int(v.size(0)) != n_e:
                return None
        elif k != "edge_index" and not _COARSE.match(k):
            return None
    perm = morton_order(d["pos"])
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device=perm.device)
    row, col = inv[d["edge_index"][0]], inv[d["edge_index"][1]]
    order = torch.argsort(col, stable=True)
    out = {}
    for k, v in d.items():
        if not torch.is_tensor(v):
            out[k] = v
        elif k in _NODE:
            out[k] = v[perm].contiguous()
        elif k == "edge_attr":
            out[k] = v[order].contiguous()
        elif k == "edge_index":
            out[k] = torch.stack([row[order], col[order]], 0).contiguous()
        else:
            out[k] = v
    return Graph(**out), perm

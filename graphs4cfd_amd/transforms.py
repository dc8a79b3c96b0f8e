"""`gfd.transforms`: the pre-processing that produces the Graph attribute layouts the models read,
with the reference's class names and constructor arguments (graphs4cfd/transforms/{connect,scale,mus,mugs,remus,
interpolate}.py) on top of the vectorised builders of `synthetic.py` — no torch_cluster / torch_geometric, and none of
the reference's O(E^2) Python loops (transforms/remus.py:36,159-161).  Outputs are checked against the reference's own
transforms in tests/test_synthetic.py (tests/golden/transforms.pt, models_mugs.pt).

Host-side, once per mesh.  The data-scaling / augmentation transforms of the
training pipelines (ScaleNs, AddUniformNoise, NodeSubset, RandomNodeSubset, GraphRotation, RandomGraphRotation, RandomGraphFlip)
live in `augment.py` and are re-exported here.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import synthetic as S
from .augment import (AddUniformNoise, GraphRotation, InterpolateNodes, InterpolateNodesToXml, NodeSubset, interpolate_nodes, RandomGraphFlip, RandomGraphRotation, RandomNodeSubset,      # noqa: F401
                      ScaleNs, flip_graph_dim, rotate_graph)
from .graph import Graph


class Compose:
    """`torchvision.transforms.Compose` as the reference's examples use it: apply the transforms in order."""

    def __init__(self, transforms: Sequence):
        self.transforms = list(transforms)

    def __call__(self, graph: Graph) -> Graph:
        for t in self.transforms:
            graph = t(graph)
        return graph


class ConnectKNN:
    """kNN edges grouped by target, `edge_attr = pos[col] - pos[row]`, optionally on a periodic domain
    (reference: transforms/connect.py:9-92)."""

    def __init__(self, k: int, period: Optional[Union[Tuple, None]] = (None, None)):
        self.k, self.period = k, period

    def __call__(self, graph: Graph) -> Graph:
        graph.edge_index, graph.edge_attr = S.connect_knn(graph.pos, self.k, period=self.period)
        return graph


class ScaleEdgeAttr:
    """`edge_attr /= 2r` (reference: transforms/scale.py:15-31)."""

    def __init__(self, r: float):
        self.r = r

    def __call__(self, graph: Graph) -> Graph:
        graph.edge_attr = S.true_divide_by(graph.edge_attr, 2 * self.r)
        return graph


class GridClustering:
    """Low-resolution node sets of MuS-GNN by voxel clustering (reference: transforms/mus.py:40-65):
    `pos_l, cluster_l, mask_l, idx{l-1}_to_idx{l}, e_{l-1}{l}` for l = 2 .. len(cells_size) + 1."""

    def __init__(self, cells_size: List[float]):
        if not 1 <= len(cells_size) <= 3:
            raise ValueError("MuS-GNN has 2 to 4 levels: cells_size needs 1 to 3 entries")
        self.num_levels = len(cells_size) + 1
        self.cells_size = list(cells_size)

    def __call__(self, graph: Graph) -> Graph:
        return S.add_grid_levels(graph, self.cells_size)


class GuillardCoarseningAndConnectKNN:
    """Low-resolution graphs of gMuS-GNN: node-nested Guillard coarsening + kNN per level
    (reference: transforms/mugs.py:32-89).  Sets edge_index/edge_attr, coarse_mask{l}, edge_index{l} (level-1 ids),
    edge_attr{l}, each scaled by 1 / (2 scale_edge_attr[l-1])."""

    def __init__(self, k: Sequence[int], period=None, scale_edge_attr: Optional[Sequence] = None):
        assert 1 < len(k) < 5, "The number of levels in gMuS-GNN must be between 2 and 4."
        self.k, self.period = list(k), period
        self.scale_edge_attr = list(scale_edge_attr) if scale_edge_attr is not None else [None] * len(k)

    def __call__(self, graph: Graph) -> Graph:
        pos, n = graph.pos, int(graph.pos.size(0))
        graph.edge_index, ea = S.connect_knn(pos, self.k[0], period=self.period)
        graph.edge_attr = ea if self.scale_edge_attr[0] is None else S.true_divide_by(ea, 2 * self.scale_edge_attr[0])
        prev, ei_local = torch.ones(n, dtype=torch.bool, device=pos.device), graph.edge_index
        for l in range(2, len(self.k) + 1):
            cm = torch.zeros(n, dtype=torch.bool, device=pos.device)
            cm[prev] = S.guillard_coarsening(ei_local, int(prev.sum()))
            idx = cm.nonzero().reshape(-1)
            ei_local, ea = S.connect_knn(pos[idx], self.k[l - 1], period=self.period)      # ("auto": the extent of this level's nodes, as the reference)
            sc = self.scale_edge_attr[l - 1]
            setattr(graph, f"coarse_mask{l}", cm)
            setattr(graph, f"edge_index{l}", idx[ei_local])
            setattr(graph, f"edge_attr{l}", ea if sc is None else S.true_divide_by(ea, 2 * sc))
            prev = cm
        return graph


class BuildRemusGraph:
    """The three-level REMuS-GNN graph: kNN + Guillard coarsening per level, edge unit vectors and their pseudo-inverses,
    angle graphs within and between levels (reference: transforms/remus.py:63-148)."""

    def __init__(self, num_levels: int, k: int, period=None, scale_edge_length: Optional[Sequence] = None):
        if num_levels != 3:
            raise NotImplementedError("REMuS-GNN (nn/remus_gnn.py) has exactly 3 levels")
        if scale_edge_length is None or any(s is None or s == "auto" for s in scale_edge_length):
            raise NotImplementedError("scale_edge_length must give one number per level")
        self.num_levels, self.k, self.period, self.scale_edge_length = num_levels, k, period, tuple(scale_edge_length)

    def __call__(self, graph: Graph) -> Graph:
        built = S.remus_graph(int(graph.pos.size(0)), k=self.k, scale=self.scale_edge_length, pos=graph.pos, period=self.period)
        skip = ("field", "glob", "omega", "pos", "y_idx_21", "x_idx_21", "weights_21", "y_idx_32", "x_idx_32", "weights_32")
        for key, val in built.to_dict().items():
            if key not in skip:
                setattr(graph, key, val)
        return graph


class BuildKnnInterpWeights:
    """Indices and inverse-squared-distance weights of the interpolation up-sampling in gMuS-GNN and REMuS-GNN
    (reference: transforms/interpolate.py:110-155): y_idx_{l}{l-1}, x_idx_{l}{l-1}, weights_{l}{l-1} for every coarse level."""

    def __init__(self, k: int):
        self.k = k

    def __call__(self, graph: Graph) -> Graph:
        """On a collated batch (the reference applies this transform to the batch, examples/training/NsREMuSGNN/*.py:39-41, with
        `knn(..., batch_x, batch_y)`): neighbours are searched inside each graph; the indices address the batch's compact
        level-l / level-(l-1) node lists."""
        batch = getattr(graph, "batch", None)
        n_graphs = int(batch.max()) + 1 if batch is not None and batch.numel() else 1
        prev = None
        for l in (2, 3, 4):
            if not hasattr(graph, f"coarse_mask{l}"):
                break
            cm = getattr(graph, f"coarse_mask{l}")
            pos_c, pos_f = graph.pos[cm], (graph.pos if prev is None else graph.pos[prev])
            if n_graphs == 1:
                y, x, w = S.knn_interp_weights(pos_c, pos_f, self.k)
            else:
                b_c, b_f = batch[cm], (batch if prev is None else batch[prev])
                ys, xs, ws = [], [], []
                for b in range(n_graphs):
                    ic, jf = (b_c == b).nonzero().reshape(-1), (b_f == b).nonzero().reshape(-1)
                    if not (int(ic[-1]) - int(ic[0]) + 1 == ic.numel() and int(jf[-1]) - int(jf[0]) + 1 == jf.numel()):
                        raise ValueError("the nodes of every graph must be contiguous in the batch (Collater output)")
                    yb, xb, wb = S.knn_interp_weights(pos_c[ic], pos_f[jf], self.k)
                    ys.append(yb + int(jf[0])); xs.append(xb + int(ic[0])); ws.append(wb)
                y, x, w = torch.cat(ys), torch.cat(xs), torch.cat(ws)
            setattr(graph, f"y_idx_{l}{l - 1}", y); setattr(graph, f"x_idx_{l}{l - 1}", x); setattr(graph, f"weights_{l}{l - 1}", w)
            prev = cm
        return graph

"""Data scaling and augmentation transforms of the training pipelines (reference: graphs4cfd/transforms/scale.py:33-80,
noise.py:6-24, subset.py:7-60, geometric.py:33-252) — the per-sample side of `GNN.fit`'s data path, so the reference's
training scripts (examples/training/*/*.py) port with their transform lists unchanged.  Host-side, on the CPU tensors a
dataset yields; exported through `gfd.transforms`.

Conventions kept from the reference: fields are stored time-major with `num_fields` interleaved components
(`field[:, c::num_fields]` = component c at every input time); a rotation acts on row vectors as `x @ R` with
`R = [[cos, sin], [-sin, cos]]` (geometric.py:61, :67: `(R * x.unsqueeze(-1)).sum(1)`); graphs in the edge-angle formulation
(`angle_index`: REMuS) keep their rotation-invariant `edge_attr` and rotate the edge unit vectors and their
pseudo-inverses instead (:69-86); flipping such graphs is refused (:190-192).
"""
from __future__ import annotations

import math
import random
from typing import Dict, Iterable, Optional, Tuple, Union

import numpy as np
import torch

from .graph import Graph


def _check_eq(eq: Optional[str], format: Optional[str]) -> Optional[str]:
    if eq is None:
        return None
    eq = eq.lower()
    if eq == "ns":
        if format is None:
            raise AssertionError("format must be specified for NS equations")
        if format not in ("uvp", "uv"):
            raise ValueError(f"Unknown format {format}, must be 'uvp' or 'uv'")
    elif eq != "adv":
        raise ValueError(f"Unknown equation type {eq}, must be 'ns' or 'adv'")
    return eq


# ------------------------------------------------------------------------------------- scaling / noise / subsets
class ScaleNs:
    """`x <- (x - c) / d` with c = (a+b)/2, d = |b-a|/2 for the components named in `scaling` ('u', 'v', 'p': columns
    0, 1, 2 of every time slice of `field` and `target`; 'Re': `glob`) (scale.py:33-80)."""

    def __init__(self, scaling: Dict[str, Tuple[float, float]], format: str):
        assert format in ["uvp", "uv"], f"Unknown format {format}, must be 'uvp' or 'uv'"
        self.num_fields = 3 if format == "uvp" else 2
        mid = lambda k: (0.5 * (scaling[k][0] + scaling[k][1]), 0.5 * abs(scaling[k][1] - scaling[k][0])) if k in scaling else None
        self.components = [mid("u"), mid("v")] + ([mid("p")] if format == "uvp" else [])
        self.Re = mid("Re")

    def __call__(self, graph: Graph) -> Graph:
        nf = self.num_fields
        for c, cd in enumerate(self.components):
            if cd is None:
                continue
            for name in ("field", "target"):
                if hasattr(graph, name):
                    t = getattr(graph, name)
                    t[:, c::nf] = (t[:, c::nf] - cd[0]) / cd[1]
        if self.Re is not None and hasattr(graph, "glob"):
            graph.glob = (graph.glob - self.Re[0]) / self.Re[1]
        return graph


class AddUniformNoise:
    """`field += U[-eps, eps]` (noise.py:6-24)."""

    def __init__(self, eps: float):
        self.eps = eps

    def __call__(self, graph: Graph) -> Graph:
        graph.field += self.eps * (2 * torch.rand_like(graph.field) - 1)
        return graph


_NODE_ATTRS = ("pos", "field", "omega", "target", "bound", "loc", "glob")


def _take_nodes(graph: Graph, idx) -> Graph:
    for name in _NODE_ATTRS:
        if hasattr(graph, name):
            setattr(graph, name, getattr(graph, name)[idx])
    return graph


class NodeSubset:
    """Keep the nodes `idx` (before any connectivity is built) (subset.py:7-30)."""

    def __init__(self, idx: Iterable[int]):
        self.idx = idx

    def __call__(self, graph: Graph) -> Graph:
        return _take_nodes(graph, self.idx)


class RandomNodeSubset:
    """Keep a random subset: a float is a fraction of the nodes, an int a count (subset.py:32-60)."""

    def __init__(self, num_nodes: Union[float, int]):
        self.num_nodes = num_nodes

    def __call__(self, graph: Graph) -> Graph:
        n = graph.num_nodes
        k = int(self.num_nodes * n) if isinstance(self.num_nodes, float) else self.num_nodes
        return _take_nodes(graph, random.sample(range(n), k=k))


# ------------------------------------------------------------------------------------- rotation / flip
def _rotation(theta_deg, dim: int) -> torch.Tensor:
    if dim == 2:
        assert isinstance(theta_deg, float), "theta must be a float"
        t = math.radians(theta_deg)
        return torch.tensor([[math.cos(t), math.sin(t)], [-math.sin(t), math.cos(t)]], dtype=torch.float32)
    if dim == 3:
        assert isinstance(theta_deg, Iterable) and len(theta_deg) == 3, "theta must be an iterable of length 3"
        a, b, c = (math.radians(float(x)) for x in theta_deg)
        ca, sa, cb, sb, cc, sc = math.cos(a), math.sin(a), math.cos(b), math.sin(b), math.cos(c), math.sin(c)
        return torch.tensor([[ca * cb, ca * sb * sc - sa * cc, ca * sb * cc + sa * sc],
                             [sa * cb, sa * sb * sc + ca * cc, sa * sb * cc - ca * sc],
                             [-sb, cb * sc, cb * cc]], dtype=torch.float32)
    raise ValueError("dim must be 2 or 3")


def rotate_graph(graph: Graph, theta, eq: Optional[str] = None, format: Optional[str] = None) -> Graph:
    """Rotate positions, edge vectors and vector fields by `theta` degrees (2-D: about z; 3-D: Tait-Bryan angles)
    (geometric.py:33-114)."""
    eq = _check_eq(eq, format)
    R = _rotation(theta, int(graph.pos.size(1)))
    rot = lambda x: x @ R
    graph.pos = rot(graph.pos)
    if hasattr(graph, "angle_index"):
        for s in ("", "2", "3", "4"):
            name = f"edgeUnitVector{s}"
            if hasattr(graph, name):
                u = rot(getattr(graph, name))
                setattr(graph, name, u)
                n = graph.num_nodes if s == "" else int(getattr(graph, f"coarse_mask{s}").sum())
                setattr(graph, f"edgeUnitVectorInverse{s}", torch.linalg.pinv(u.view(n, -1, 2)))
    else:
        for s in ("", "2", "3", "4"):
            name = f"edge_attr{s}"
            if getattr(graph, name, None) is not None:
                setattr(graph, name, rot(getattr(graph, name)))
    if eq == "adv":
        graph.loc = rot(graph.loc)
    elif eq == "ns":
        nf = 3 if format == "uvp" else 2
        for name in ("field", "target"):
            t = getattr(graph, name)
            for c in range(0, int(t.size(1)), nf):
                t[:, c:c + 2] = rot(t[:, c:c + 2])
    return graph


class GraphRotation:
    def __init__(self, theta, eq: Optional[str] = None, format: Optional[str] = None):
        self.theta, self.eq, self.format = theta, eq, format

    def __call__(self, graph: Graph) -> Graph:
        return rotate_graph(graph, self.theta, eq=self.eq, format=self.format)


class RandomGraphRotation:
    def __init__(self, eq: Optional[str] = None, format: Optional[str] = None):
        self.eq, self.format = eq, format

    def __call__(self, graph: Graph) -> Graph:
        dim = int(graph.pos.size(1))
        theta = float(np.random.uniform(0, 360)) if dim == 2 else np.random.uniform(0, 360, size=(3,))
        return rotate_graph(graph, theta, eq=self.eq, format=self.format)


def flip_graph_dim(graph: Graph, dim: int, eq: Optional[str] = None, format: Optional[str] = None) -> Graph:
    """Mirror the graph along axis `dim`: positions, `loc`, edge vectors and that velocity component (geometric.py:170-216)."""
    eq = _check_eq(eq, format)
    if dim >= graph.pos.size(1):
        raise ValueError(f"Dimension {dim} is greater than the maximum dimension of the graph ({graph.pos.size(1)})")
    if hasattr(graph, "angle_index"):
        raise ValueError("Flipping graphs with angle_index is not supported")
    graph.pos[:, dim] = -graph.pos[:, dim]
    if hasattr(graph, "loc"):
        graph.loc[:, dim] = -graph.loc[:, dim]
    for s in ("", "2", "3", "4"):
        t = getattr(graph, f"edge_attr{s}", None)
        if t is not None:
            t[:, dim] = -t[:, dim]
    if eq == "ns":
        nf = 3 if format == "uvp" else 2
        graph.field[:, dim::nf] = -graph.field[:, dim::nf]
        graph.target[:, dim::nf] = -graph.target[:, dim::nf]
    return graph


class RandomGraphFlip:
    """Each enabled axis is mirrored with probability 1/2 (geometric.py:219-252)."""

    def __init__(self, x_flip: bool = True, y_flip: bool = True, z_flip: bool = True, eq: Optional[str] = None,
                 format: Optional[str] = None):
        self.flip, self.eq, self.format = (x_flip, y_flip, z_flip), eq, format

    def __call__(self, graph: Graph) -> Graph:
        for axis, flag in enumerate(self.flip[: int(graph.pos.size(1))]):
            if flag and np.random.randint(2):
                graph = flip_graph_dim(graph, axis, eq=self.eq, format=self.format)
        return graph


# ------------------------------------------------------------------------------------- interpolation to another node set
def interpolate_nodes(graph: Graph, pos: torch.Tensor, method: Optional[str] = None) -> Graph:
    """Interpolate the fields of a point cloud (no edges yet) to the nodes `pos` with `scipy.interpolate.griddata`
    (transforms/interpolate.py:13-49): `loc`, `glob`, `field`, `target` with `method` (default 'cubic' in 2-D, 'linear' in 3-D),
    `omega` and `bound` linearly — omega thresholded at 0.9, bound rounded."""
    from scipy.interpolate import griddata
    if getattr(graph, "edge_index", None) is not None:
        raise ValueError("Graphs cannot be interpolated, only sets of nodes.")
    if method is None:
        method = "cubic" if pos.size(1) == 2 else "linear"
    src = graph.pos.numpy()
    resample = lambda t, m: griddata(src, t.numpy(), pos.numpy(), method=m)
    for name in ("loc", "glob"):
        if hasattr(graph, name):
            setattr(graph, name, torch.tensor(resample(getattr(graph, name), method).astype(np.float32)))
    graph.field = torch.tensor(resample(graph.field, method).astype(np.float32))
    graph.target = torch.tensor(resample(graph.target, method).astype(np.float32))
    omega = torch.tensor(resample(graph.omega, "linear").astype(np.float32))
    graph.bound = torch.tensor(np.round(resample(graph.bound, "linear")), dtype=torch.uint8)
    graph.omega = (omega >= 0.9).float()
    graph.pos = pos
    return graph


class InterpolateNodes:
    """`interpolate_nodes` to a fixed node set (transforms/interpolate.py:52-68)."""

    def __init__(self, pos: torch.Tensor) -> None:
        self.pos = pos

    def __call__(self, graph: Graph) -> Graph:
        return interpolate_nodes(graph, self.pos)


class InterpolateNodesToXml:
    """`interpolate_nodes` to the vertices of a NekMesh-generated xml file (`GEOMETRY/VERTEX/V`), or of one drawn at random
    from a directory whose name ends in `_xml` (`num_meshes` of its files, chosen with replacement, or 'all')
    (transforms/interpolate.py:71-107)."""

    def __init__(self, xml_file: str, num_meshes: Union[int, str] = "all"):
        import os
        if isinstance(num_meshes, str):
            assert num_meshes == "all", "num_meshes must be an integer or 'all'"
        if xml_file[-4:] == ".xml":
            self.xml_files = [xml_file]
        elif xml_file[-4:] == "_xml":
            files = [os.path.join(xml_file, f) for f in sorted(os.listdir(xml_file))]
            self.xml_files = random.choices(files, k=len(files) if num_meshes == "all" else num_meshes)
        else:
            raise ValueError(f"{xml_file}: expected an .xml file or a directory named *_xml")

    def __call__(self, graph: Graph) -> Graph:
        from xml.etree import ElementTree
        dim = int(graph.pos.size(1))
        vertices = ElementTree.parse(random.choice(self.xml_files)).findall("GEOMETRY/VERTEX/V")
        pos = torch.tensor([list(map(float, v.text.split()[:dim])) for v in vertices], dtype=torch.float32)
        return interpolate_nodes(graph, pos)

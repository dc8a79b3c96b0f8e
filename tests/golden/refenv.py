"""Test-only import environment for the *reference* (used in the authoring container only).

`/root/reference/graphs4cfd` imports third-party modules that are not installed in this
image (torch_geometric, torch_cluster, h5py, torchvision, tensorboard).  This module
registers minimal stand-ins for exactly the names the reference imports, restating the
*documented* third-party semantics, so that the reference's own `graphs4cfd/nn` and
`graphs4cfd/transforms` sources run **unmodified** and can generate golden vectors
(`make_golden.py`).  Nothing here ships to the product path and nothing under
`graphs4cfd_amd/` imports it.  On the GPU box `/root/reference` does not exist and this
module is never imported.

Restated semantics (PyG >= 2.3, torch_cluster HEAD):
  * `torch_geometric.utils.scatter(src, index, dim, dim_size, reduce)`:
      sum  = zeros(dim_size).scatter_add_(index, src)
      mean = sum / count.clamp(min=1)           (empty targets give 0)
      dim_size=None -> int(index.max()) + 1
  * `coalesce(edge_index, edge_attr, num_nodes, reduce)`: sort by row*num_nodes+col,
      duplicates reduced with `scatter(reduce)`.
  * `remove_self_loops`: mask row != col on both index and attr.
  * `torch_geometric.nn.voxel_grid(pos, size, batch)` -> torch_cluster.grid_cluster on
      cat(pos, batch) with start=min, end=max, voxel id = sum_d floor((p_d-start_d)/size_d)*stride_d,
      stride = exclusive cumprod of floor((end-start)/size)+1.
  * `knn_graph(x, k)` (loop=False, flow='source_to_target'): row = neighbour, col = centre,
      grouped by centre, neighbours by ascending distance.
  * `knn(x, y, k)` -> [2, |y|*k]: row0 = query (y) index, row1 = neighbour (x) index.
"""
import os
import sys
import types

import numpy as np
import torch
from scipy.spatial import cKDTree

REFERENCE_ROOT = "/root/reference"


# ----------------------------------------------------------------------------- data
class Data:
    """Attribute bag with the slice of the PyG `Data` behaviour the reference relies on."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        for key in ("pos", "x", "field", "batch"):
            if key in self.__dict__ and self.__dict__[key] is not None:
                return self.__dict__[key].size(0)
        return int(self.edge_index.max()) + 1

    @property
    def num_edges(self):
        return self.edge_index.size(1)

    def keys(self):
        return list(self.__dict__.keys())

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
        return self

    def clone(self):
        out = self.__class__()
        for k, v in self.__dict__.items():
            out.__dict__[k] = v.clone() if torch.is_tensor(v) else v
        return out


class Batch(Data):
    @classmethod
    def from_data_list(cls, data_list):
        """`torch_geometric.data.Batch.from_data_list` as PyG documents it (Data.__cat_dim__ / Data.__inc__): a tensor attribute
        whose name contains 'index' (or is 'face') is concatenated along its LAST dimension with the running node count added to
        each graph's copy, every other tensor along dim 0 unchanged; `batch` = graph id of every node, `ptr` = node offsets;
        numbers are stacked into a tensor, other objects collected in a list.  (The reference's own Collater corrects the REMuS
        angle indices BEFORE this call, loader.py:17-51 — `solve([g1, g2])`, nn/model.py:308-309, calls it bare.)"""
        out = cls()
        keys = data_list[0].keys()
        offset, batch, ptr = 0, [], [0]
        cols = {k: [] for k in keys}
        for gi, g in enumerate(data_list):
            n = g.num_nodes
            for k in keys:
                v = getattr(g, k)
                if torch.is_tensor(v):
                    cols[k].append(v + offset if ("index" in k or k == "face") else v)
                else:
                    cols[k].append(v)
            batch.append(torch.full((n,), gi, dtype=torch.long))
            offset += n
            ptr.append(offset)
        for k, vs in cols.items():
            if torch.is_tensor(vs[0]):
                setattr(out, k, torch.cat(vs, dim=-1 if ("index" in k or k == "face") else 0))
            elif isinstance(vs[0], (int, float)):
                setattr(out, k, torch.tensor(vs))
            else:
                setattr(out, k, vs)
        out.batch = torch.cat(batch)
        out.ptr = torch.tensor(ptr)
        return out


class Dataset(torch.utils.data.Dataset):
    pass


# ----------------------------------------------------------------------------- utils
def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    size = (dim_size,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    if reduce in ("sum", "add"):
        return src.new_zeros(size).scatter_add_(0, idx, src)
    if reduce == "mean":
        count = src.new_zeros(dim_size).scatter_add_(0, index, src.new_ones(src.size(0)))
        count = count.clamp(min=1)
        out = src.new_zeros(size).scatter_add_(0, idx, src)
        return out / count.view(-1, *([1] * (src.dim() - 1)))
    raise ValueError(reduce)


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    if edge_attr is None:
        return edge_index, None
    return edge_index, edge_attr[mask]


def coalesce(edge_index, edge_attr=None, num_nodes=None, reduce="sum"):
    nnz = edge_index.size(1)
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1
    idx = edge_index.new_empty(nnz + 1)
    idx[0] = -1
    idx[1:] = edge_index[0] * num_nodes + edge_index[1]
    sorted_idx, perm = torch.sort(idx[1:], stable=True)
    idx[1:] = sorted_idx
    edge_index = edge_index[:, perm]
    if edge_attr is not None:
        edge_attr = edge_attr[perm]
    mask = idx[1:] > idx[:-1]
    if bool(mask.all()):
        return edge_index, edge_attr
    edge_index = edge_index[:, mask]
    dim_size = edge_index.size(1)
    seg = torch.arange(0, nnz, device=edge_index.device)
    seg = seg - (~mask).cumsum(0)
    if edge_attr is not None:
        edge_attr = scatter(edge_attr, seg, 0, dim_size, reduce)
    return edge_index, edge_attr


# ----------------------------------------------------------------------------- nn (cluster)
def voxel_grid(pos, size, batch=None, start=None, end=None):
    pos = pos.unsqueeze(-1) if pos.dim() == 1 else pos
    dim = pos.size(1)
    if batch is None:
        batch = torch.zeros(pos.size(0), dtype=torch.long)
    pos = torch.cat([pos, batch.view(-1, 1).to(pos.dtype)], dim=-1)
    if not isinstance(size, (list, tuple)):
        size = [float(size)] * dim
    size = torch.tensor(list(size) + [1.0], dtype=pos.dtype)
    start = pos.min(0)[0]
    end = pos.max(0)[0]
    p = pos - start.unsqueeze(0)
    num_voxels = (end - start).true_divide(size).to(torch.long) + 1
    num_voxels = num_voxels.cumprod(0)
    num_voxels = torch.cat([torch.ones(1, dtype=torch.long), num_voxels], 0)
    num_voxels = num_voxels.narrow(0, 0, size.size(0))
    out = p.true_divide(size.view(1, -1)).to(torch.long)
    out = out * num_voxels.view(1, -1)
    return out.sum(1)


def knn(x, y, k, batch_x=None, batch_y=None, cosine=False, num_workers=1):
    assert batch_x is None and batch_y is None
    tree = cKDTree(x.detach().cpu().double().numpy())
    _, nbr = tree.query(y.detach().cpu().double().numpy(), k=k)
    nbr = np.asarray(nbr).reshape(y.size(0), k)
    row = torch.arange(y.size(0)).repeat_interleave(k)
    col = torch.from_numpy(nbr.reshape(-1).astype(np.int64))
    return torch.stack([row, col], 0)


def knn_graph(x, k, batch=None, loop=False, flow="source_to_target", cosine=False, num_workers=1):
    assert batch is None
    edge_index = knn(x, x, k if loop else k + 1)
    if flow == "source_to_target":
        row, col = edge_index[1], edge_index[0]
    else:
        row, col = edge_index[0], edge_index[1]
    if not loop:
        mask = row != col
        row, col = row[mask], col[mask]
    return torch.stack([row, col], 0)


# ----------------------------------------------------------------------------- install
class _Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class _SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def install():
    """Register the stand-ins and put the reference on sys.path. Idempotent."""
    if "torch_geometric" in sys.modules and getattr(sys.modules["torch_geometric"], "_g4c_standin", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tg = mod("torch_geometric", _g4c_standin=True)
    tg.data = mod("torch_geometric.data", Data=Data, Batch=Batch, Dataset=Dataset)
    tg.utils = mod("torch_geometric.utils", scatter=scatter, coalesce=coalesce,
                   remove_self_loops=remove_self_loops)
    tg.nn = mod("torch_geometric.nn", voxel_grid=voxel_grid, knn_graph=knn_graph, knn=knn)
    mod("h5py")
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=_Compose)
    if "torch.utils.tensorboard" not in sys.modules:
        tb = mod("torch.utils.tensorboard", SummaryWriter=_SummaryWriter)
        torch.utils.tensorboard = tb
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    install()
    import graphs4cfd  # noqa: the reference, from /root/reference
    assert os.path.abspath(graphs4cfd.__file__).startswith(REFERENCE_ROOT), \
        f"'graphs4cfd' resolved to {graphs4cfd.__file__} (this repository's alias package?), not to the reference"
    return graphs4cfd

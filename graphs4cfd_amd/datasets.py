"""`gfd.datasets`: the reference's simulation datasets (graphs4cfd/datasets.py:10-337) — `Adv`, `NsCircle`, `NsEllipse` on top
of `Dataset` — the data format on the input side of `GNN.fit`.

A dataset is ONE array `data[simulation, node, column]`, NaN-padded along the node axis to the largest mesh; the column layout
per class is the reference's (`data2graph`).  A sample is a `Graph` holding only the point cloud and its fields: `pos`, `field`
(`n_in` time steps, `step` apart, components interleaved), `target` (`n_out` steps), `glob` / `loc`, `bound`, `omega`; the
connectivity is built by the transforms.  `__getitem__` draws the start of the window at random in `[0, T - length]`
(datasets.py:68-72); `get_sequence` is the deterministic form.

The reference reads the array from an HDF5 file (`h5py.File(path)["data"]`, per access or preloaded).  h5py is not part of
this image: `path` may name an `.h5` / `.hdf5` file (needs h5py: fails loudly without it), a `.npy` file (memory-mapped when
not preloaded) or a `.pt` tensor file; or the array can be handed over directly (`data=`).
"""
from __future__ import annotations

import random
from typing import Callable, Dict, Optional

import numpy as np
import torch

from .graph import Graph


class Dataset(torch.utils.data.Dataset):
    """Base class: storage, window arithmetic, transform application (datasets.py:10-137).

    Args: path, transform (applied to every sample), training_info {'n_in', 'n_out', 'step', 'T'}, idx (load only that
    simulation; needs preload), preload (keep the array in memory), data (the array itself instead of a file)."""

    def __init__(self, path: Optional[str] = None, transform: Optional[Callable] = None, training_info: Optional[Dict] = None,
                 idx: Optional[int] = None, preload: bool = False, data=None):
        self.path, self.transform, self.training_info, self.preload = path, transform, training_info, preload
        if training_info:
            self.training_sequences_length = (training_info["n_in"] + training_info["n_out"]) * training_info["step"] - (training_info["step"] - 1)
            self.training_sequences_T = training_info["T"]
        self.h5_data = None
        if data is not None:
            self.h5_data = torch.as_tensor(np.asarray(data) if not torch.is_tensor(data) else data, dtype=torch.float32)
            self.preload = True
        if idx is not None:
            if not self.preload:
                raise ValueError('If input argument to Dataset.__init__() idx is not None, then argument preload must be True.')
            one = self.h5_data[idx] if self.h5_data is not None else self._read(idx)
            self.h5_data = one.unsqueeze(0) if one.ndim == 2 else one
        elif self.preload and self.h5_data is None:
            self.load()

    # -- storage ------------------------------------------------------------------------------------------------
    def _open(self):
        """(array-like [S, N, C], closer)"""
        if self.path is None:
            raise ValueError("Dataset needs `path` or `data`")
        ext = self.path.rsplit(".", 1)[-1].lower()
        if ext in ("h5", "hdf5"):
            try:
                import h5py
            except ImportError as exc:
                raise ImportError(f"{self.path}: reading HDF5 needs h5py, which is not installed here; convert the 'data' array "
                                  "to .npy / .pt or pass it as data=") from exc
            f = h5py.File(self.path, "r")
            return f["data"], f.close
        if ext == "npy":
            return np.load(self.path, mmap_mode="r"), (lambda: None)
        if ext in ("pt", "pth"):
            return torch.load(self.path, map_location="cpu"), (lambda: None)
        raise ValueError(f"{self.path}: expected .h5 / .hdf5, .npy or .pt")

    def _read(self, idx=None) -> torch.Tensor:
        arr, close = self._open()
        try:
            sel = arr if idx is None else arr[idx]
            return sel.to(torch.float32) if torch.is_tensor(sel) else torch.tensor(np.array(sel), dtype=torch.float32)
        finally:
            close()

    def load(self):
        """Load the dataset in memory."""
        print("Loading dataset:", self.path)
        self.h5_data = self._read()
        self.preload = True

    def __len__(self) -> int:
        if self.h5_data is not None:
            return int(self.h5_data.shape[0])
        arr, close = self._open()
        try:
            return int(arr.shape[0])
        finally:
            close()

    # -- samples ------------------------------------------------------------------------------------------------
    def __getitem__(self, idx: int) -> Graph:
        start = random.randint(0, self.training_sequences_T - self.training_sequences_length)
        return self.get_sequence(idx, start, n_in=self.training_info["n_in"], n_out=self.training_info["n_out"], step=self.training_info["step"])

    def get_sequence(self, idx: int, sequence_start: int = 0, n_in: int = 1, n_out: int = 1, step: int = 1) -> Graph:
        """The idx-th simulation from time index `sequence_start`: n_in input steps, then n_out target steps, `step` apart."""
        data = self.h5_data[idx] if self.h5_data is not None else self._read(idx)
        length = (n_in + n_out) * step - (step - 1)
        idx0, idx1, idx2 = sequence_start, sequence_start + n_in * step, sequence_start + length
        graph = self.data2graph(data, idx0, idx1, idx2, step)
        if self.transform:
            out = self.transform(graph)
            graph = graph if out is None else out
        return graph

    def data2graph(self, data: torch.Tensor, idx0: int, idx1: int, idx2: int, step: int) -> Graph:
        raise NotImplementedError

    @staticmethod
    def _real_nodes(data: torch.Tensor) -> torch.Tensor:
        """Rows before the NaN padding."""
        return data[: int((data[:, 0] == data[:, 0]).sum())]


class Adv(Dataset):
    """Advection (https://doi.org/10.5281/zenodo.7861710).  Columns: x, y | loc (2) | boundary code | field(t0), field(t1), ...
    Boundary codes: 0 inner, 1 periodic, 2 inlet, 3 outlet; omega = inlet (datasets.py:158-197)."""

    def data2graph(self, data, idx0, idx1, idx2, step) -> Graph:
        data = self._real_nodes(data)
        g = Graph(pos=data[:, :2], loc=data[:, 2:4], field=data[:, 5 + idx0:5 + idx1:step], target=data[:, 5 + idx1:5 + idx2:step])
        g.bound = data[:, 4].type(torch.uint8)
        g.omega = (g.bound == 2).float().unsqueeze(1)
        return g


class _Ns(Dataset):
    """Columns: x, y | Re | boundary code | `stride` values per time step, of which the first 3 (uvp) or 2 (uv) are kept.
    Boundary codes: 0 inner, 1 periodic, 2 inlet, 3 outlet, 4 wall; omega = inlet or wall."""
    _STRIDE = 3

    def __init__(self, format: str, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert format in ["uv", "uvp"], f"Format {format} not supported, use 'uv' or 'uvp'"
        self.format = format

    def data2graph(self, data, idx0, idx1, idx2, step) -> Graph:
        data = self._real_nodes(data)
        n, nf = int(data.size(0)), (3 if self.format == "uvp" else 2)
        series = data[:, 4:].reshape(n, -1, self._STRIDE)
        g = Graph(pos=data[:, :2], glob=data[:, 2:3],
                  field=series[:, idx0:idx1:step, :nf].reshape(n, -1), target=series[:, idx1:idx2:step, :nf].reshape(n, -1))
        g.bound = data[:, 3].type(torch.uint8)
        g.omega = ((g.bound == 2) | (g.bound == 4)).float().unsqueeze(1)
        return g


class NsCircle(_Ns):
    """Incompressible flow around a circular cylinder (https://doi.org/10.5281/zenodo.7870707): u, v, p per time step
    (datasets.py:200-266)."""
    _STRIDE = 3


class NsEllipse(_Ns):
    """Incompressible flow around an elliptical cylinder (https://doi.org/10.5281/zenodo.7892171): six values per time step,
    u, v, p first (datasets.py:269-337)."""
    _STRIDE = 6

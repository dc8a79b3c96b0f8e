"""`GNN` base class: model construction / checkpoint loading and the rollout loop.

Mirrors the inference-side surface of the reference's graphs4cfd/nn/model.py: constructor
`(arch, weights, checkpoint, device)`, `load_model` (:112-130), `solve` (:303-321),
`shift_and_replace` (:323-327), `save_checkpoint` (:329-349), `num_params` (:351-354).  The training
loop (`fit`, `TrainConfig`, :14-82,152-301) is out of scope.

`solve` keeps the whole rollout on the device: the history-window shift and the write into
`outputs[:, nf*t:nf*(t+1)]` are one kernel reading the step index from device memory, so a step has no
host synchronisation and can be captured once in a hipGraph and replayed (`capture=True`).
"""
from __future__ import annotations

import os
from typing import List, Optional, Union

import torch
from torch import nn

from .. import _lib, ops
from ..graph import Graph


def collate(graphs: List[Graph]) -> Graph:
    """Minimal stand-in for `torch_geometric.data.Batch.from_data_list` (called at nn/model.py:309):
    tensors are concatenated along dim 0 (dim -1 for attributes whose name contains 'index', which are
    also offset by the running node count), and `batch` holds the graph id of every node."""
    out, offset, batch = {}, 0, []
    keys = graphs[0].keys()
    for gi, g in enumerate(graphs):
        n = g.num_nodes
        for k in keys:
            v = getattr(g, k)
            if not torch.is_tensor(v):
                out[k] = v
                continue
            out.setdefault(k, []).append(v + offset if "index" in k else v)
        batch.append(torch.full((n,), gi, dtype=torch.long, device=g.pos.device if "pos" in g else None))
        offset += n
    merged = {k: (torch.cat(v, dim=-1 if "index" in k else 0) if isinstance(v, list) else v) for k, v in out.items()}
    merged["batch"] = torch.cat(batch)
    return Graph(**merged)


class GNN(nn.Module):
    r"""Base class for all the GNN models.

    Args:
        arch (Optional[dict]): Dictionary with the model architecture. Defaults to `None`.
        weights (Optional[str]): Path of the weights file. Defaults to `None`.
        checkpoint (Optional[str]): Path of the checkpoint file. Defaults to `None`.
        device (Optional[torch.device]): Device where the model is loaded. Defaults to `torch.device('cpu')`
            (the reference's default); forward/solve need a HIP device.
    """

    def __init__(self, arch: Optional[dict] = None, weights: Optional[str] = None, checkpoint: Optional[str] = None,
                 device: Optional[torch.device] = torch.device('cpu')):
        super().__init__()
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        self.load_model(arch, weights, checkpoint)

    def load_model(self, arch, weights, checkpoint):
        """Architecture from an arch dict (+ optional weights file), or both from a `.chk` checkpoint
        written by `save_checkpoint` (same file format as the reference)."""
        if arch is not None and checkpoint is None:
            self.load_arch(arch)
            self.to(self.device)
            if weights is not None:
                self.load_state_dict(torch.load(weights, map_location=self.device))
            self.num_fields = arch["decoder"][1][-1] if 'decoder' in arch.keys() else None
        elif arch is None and weights is None and checkpoint is not None:
            chk = torch.load(checkpoint, map_location=self.device, weights_only=False)
            self.load_arch(chk['arch'])
            self.to(self.device)
            self.load_state_dict(chk['weights'])
            self.num_fields = chk['arch']["decoder"][1][-1] if 'decoder' in chk['arch'].keys() else None
        return

    def to(self, *args, **kwargs):
        out = super().to(*args, **kwargs)
        try:
            self.device = next(self.parameters()).device
        except StopIteration:
            pass
        return out

    # To be overwritten
    def load_arch(self, arch: dict):
        """Defines the hyper-parameters of the model; overloaded by each model class."""
        pass

    def _pretrained(self, table: dict, model: str) -> str:
        if model not in table:
            raise ValueError(f"Model {model} not recognized.")
        path = os.path.join(os.path.dirname(__file__), table[model])
        if not os.path.exists(path):
            raise FileNotFoundError(f"pretrained checkpoint {path} is not shipped (Git-LFS blob of the reference); "
                                    "pass checkpoint=<path to .chk> instead")
        return path

    # ---------------------------------------------------------------------------------- rollout
    def solve(self, graph: Union[Graph, List[Graph]], n_out: int, *, capture: Optional[bool] = None) -> torch.Tensor:
        """Evaluate the model on the graph for n_out time-steps. Returns [N, num_fields*n_out].

        capture: replay steps 2..n_out from a hipGraph captured on step 2 (default: on when
        n_out >= 4 and the environment variable G4C_HIPGRAPH is not '0')."""
        assert n_out > 0, "n_out must be greater than 0."
        self.eval()
        with torch.no_grad():
            if type(graph) is list:
                graph = collate(graph)
            else:
                graph.batch = torch.zeros(graph.num_nodes, dtype=torch.long, device=self.device)
            graph.to(self.device)
            if capture is None:
                capture = n_out >= 4 and os.environ.get("G4C_HIPGRAPH", "1") != "0"
            with Rollout(self, graph, n_out, capture=capture) as ro:
                ro.run(n_out)
                return ro.outputs

    def shift_and_replace(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """Shift the fields in x by num_fields and replace the last num_fields with y (nn/model.py:323-327).
        Returns a new tensor; `solve` uses the fused in-place device kernel instead."""
        _lib.require_hip(x, y)
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
        nf, w = self.num_fields, int(x.size(1))
        if w > nf:
            ops.copy_cols(x, out, 0, scol0=nf, width=w - nf)
        ops.copy_cols(y, out, w - nf, scol0=0, width=nf)
        return out

    def save_checkpoint(self, file_name: str, n_out: int = 1, epoch: int = 0, optimiser=None, scheduler=None, scaler=None):
        """Writes `{'arch', 'weights', ...}` in the reference's checkpoint format (nn/model.py:329-349)."""
        checkpoint = {'arch': self.arch, 'weights': self.state_dict(), 'n_out': n_out, 'epoch': epoch}
        if optimiser is not None:
            checkpoint['optimiser'] = optimiser.state_dict()
            checkpoint['lr'] = optimiser.param_groups[0]['lr']
        if scheduler is not None:
            checkpoint['scheduler'] = scheduler.state_dict()
        if scaler is not None:
            checkpoint['scaler'] = scaler.state_dict()
        torch.save(checkpoint, file_name)
        return

    @property
    def num_params(self):
        """Returns the number of trainable parameters."""
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class Rollout:
    """Device-resident autoregressive rollout state for one (model, Graph) pair (GNN.solve,
    nn/model.py:303-321).

    `step()` = one forward + one `g4c_rollout_advance` (writes outputs[:, nf*t:nf*(t+1)], shifts the
    history window, bumps the device-side step counter).  The first step runs eagerly and builds all
    static plans / packed weights; with `capture=True` the second step is captured into a hipGraph and
    every later step is a replay: no host synchronisation, no per-kernel launch cost.
    The caller's `graph.field` is swapped for a private working copy and restored on close()."""

    def __init__(self, model: "GNN", graph: Graph, max_steps: int, capture: bool = True):
        _lib.require_hip(graph.field)
        self.model, self.graph, self.capture = model, graph, capture
        self.nf = int(model.num_fields)
        self.max_steps = int(max_steps)
        dev = graph.field.device
        self._orig_field = graph.field
        self.field = graph.field.to(torch.float32).clone(memory_format=torch.contiguous_format)
        self.outputs = torch.zeros((graph.num_nodes, self.nf * self.max_steps), dtype=torch.float32, device=dev)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.steps_done = 0
        self._hipgraph = None
        graph.field = self.field

    def _one(self):
        pred = self.model.forward(self.graph, self.steps_done)
        ops.rollout_advance(self.field, pred, self.outputs, self.step_counter, self.nf)

    def step(self) -> None:
        if self.steps_done >= self.max_steps:
            raise RuntimeError(f"rollout buffer holds {self.max_steps} steps")
        with torch.no_grad():
            if self.steps_done == 0 or not self.capture:
                self._one()
            elif self._hipgraph is None:
                torch.cuda.synchronize(self.field.device)
                self._hipgraph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._hipgraph):   # records only; nothing executes during capture
                    self._one()
                self._hipgraph.replay()
            else:
                self._hipgraph.replay()
        self.steps_done += 1

    def run(self, n: int) -> None:
        for _ in range(n):
            self.step()

    def rewind(self) -> None:
        """Restart writing at step 0 (benchmarks: keeps the captured hipGraph and the current field)."""
        self.step_counter.zero_()
        self.steps_done = 1 if self.steps_done > 0 else 0
        if self.steps_done:   # slot 0 is kept so that replays continue from slot 1
            self.step_counter.fill_(1)

    def close(self) -> None:
        self.graph.field = self._orig_field

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

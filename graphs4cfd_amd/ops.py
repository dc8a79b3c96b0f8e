"""Thin Python wrappers over the C-ABI of libg4c.so: argument checking, output allocation
(torch is used for device memory and the current HIP stream only), one launch per call.
Nothing here computes on the host or with torch ops."""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from .plan import CsrPlan

Tensor = torch.Tensor


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).
    `with KernelTimer() as kt: ...` brackets every g4c_mlp_forward / g4c_segment_reduce launch with an
    event pair and records its algorithmic work; `kt.summary()` after a device sync."""
    active = None

    def __init__(self):
        self.records = []   # (kind, flops, bytes, start_event, end_event)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None
        return False

    def launch(self, kind: str, flops: float, nbytes: float, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        if kind.startswith("mlp_"):
            # the library picks the kernel family per launch (arithmetic, shape, row count): label the record by what ran
            kind = _lib.KERNEL_NAMES.get(int(_lib.load().g4c_mlp_last_kernel()), kind)
        self.records.append((kind, flops, nbytes, a, b))
        return out

    def summary(self):
        out = {}
        for kind, flops, nbytes, a, b in self.records:
            d = out.setdefault(kind, {"launches": 0, "seconds": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["seconds"] += a.elapsed_time(b) * 1e-3
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


def _timed(kind: str, flops: float, nbytes: float, fn):
    kt = KernelTimer.active
    return fn() if kt is None else kt.launch(kind, flops, nbytes, fn)


class StaticCache:
    """Results of launches whose inputs cannot change between the steps of one rollout (reference: nn/model.py:316-320 — `solve`
    only ever replaces `graph.field`): `selu(edge_encoder(edge_attr))` of the MuS-GNN models (nn/mus_gnn.py:73,178 of the reference)
    and REMuS-GNN's five angle encoders (nn/remus_gnn.py:136-140).  A `Rollout` / `DistributedRollout` activates its cache around
    every step; the first (eager) step fills it, the captured step finds the entries and therefore contains neither the launches
    nor the tensors in its write set.  A bare `model.forward()` has no active cache and recomputes, like the reference.

    An entry is valid for one (weights epoch, arithmetic, identity + version + shape of every input tensor): an in-place edit of
    `edge_attr`, new weights or `set_mlp_precision` recompute it — the same bits either way, the cached tensor is what the launch
    would have produced again."""
    active: Optional["StaticCache"] = None

    def __init__(self):
        self.store = {}
        self.hits = self.misses = 0
        self._prev = None

    def __enter__(self):
        self._prev, StaticCache.active = StaticCache.active, self
        return self

    def __exit__(self, *exc):
        StaticCache.active = self._prev
        return False

    def get(self, name: str, inputs: Sequence[Tensor], fn):
        key = (weights_epoch(), mlp_precision()) + tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype) for t in inputs)
        hit = self.store.get(name)
        if hit is not None and hit[0] == key:
            self.hits += 1
            return hit[1]
        self.misses += 1
        val = fn()
        self.store[name] = (key, val, tuple(inputs))      # (the inputs are kept alive: the key holds their addresses)
        return val

    def stale(self) -> bool:
        """True when an entry no longer matches what it was computed from (an in-place edit of a static input, new weights, another
        arithmetic).  A captured step contains neither the cached launches nor a look-up, so a rollout asks this before every replay
        and falls back to one eager step (which recomputes the entry) + a new capture — ADVICE r04: a replay must not keep reading
        latents of an `edge_attr` that was edited since."""
        head = (weights_epoch(), mlp_precision())
        for key, _, inputs in self.store.values():
            if key != head + tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype) for t in inputs):
                return True
        return False


def static_launch(name: str, inputs: Sequence[Tensor], fn):
    """`fn()` — or its cached result when a rollout's StaticCache is active (never while a call is recorded for autograd)."""
    c = StaticCache.active
    return fn() if (c is None or torch.is_grad_enabled()) else c.get(name, inputs, fn)


def _f32_2d(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if t.dim() != 2:
        raise ValueError(f"{name}: expected a 2-D tensor, got shape {tuple(t.shape)}")
    if t.stride(1) != 1 and t.size(1) > 1:
        t = t.contiguous()
    if t.size(0) > 1 and t.stride(0) < t.size(1):
        t = t.contiguous()
    return t


def _bf16_2d(t: Tensor) -> Tensor:
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"bf16 source: expected a 2-D tensor with unit column stride, got shape {tuple(t.shape)}")
    return t


def _ld(t: Tensor) -> int:
    return int(t.stride(0)) if t.size(0) > 1 else int(max(t.size(1), t.stride(0)))


class Source:
    """One column block of a virtually concatenated MLP input."""

    __slots__ = ("tensor", "index", "col0", "width", "negate", "pre_act", "additive", "segments", "seg_mean")

    def __init__(self, tensor: Tensor, index: Optional[Tensor] = None, col0: int = 0, width: Optional[int] = None,
                 negate: bool = False, pre_act: int = _lib.ACT_NONE, additive: bool = False,
                 segments: Optional[CsrPlan] = None, seg_mean: bool = True):
        # (bf16 rows: the message tensors a fused-aggregation launch stored in the rounded-bf16 mode, mlp_forward(rows_dtype=);
        # only as a plain 128-wide block of a launch in that mode — the C entry point checks)
        self.tensor = _bf16_2d(tensor) if tensor.dtype == torch.bfloat16 else _f32_2d(tensor, "source")
        self.index = index          # int32 gather index or None
        self.col0 = col0
        self.width = int(self.tensor.size(1) - col0 if width is None else width)
        self.negate = negate        # folded into the packed weights
        self.pre_act = pre_act      # activation applied while loading (producer stored the raw tensor)
        self.additive = additive    # already multiplied by its block of the first layer: gathered and added, not multiplied
        # aggregation on load (bf16x6 kernels): row r of the block = sum / mean of the tensor's rows in segment r of this
        # CSR plan (through its permutation, if any; `pre_act` then applies to the rows before they are added), i.e.
        # scatter(pre_act(tensor), col, reduce) without materialising it
        self.segments = segments
        self.seg_mean = seg_mean
        if segments is not None and (index is not None or additive or self.width != 128):
            raise ValueError("aggregation on load needs a 128-wide block without gather index")


def segment_reduce(src: Tensor, csr: CsrPlan, mean: bool, act: int = _lib.ACT_NONE, out: Optional[Tensor] = None,
                   src_act: int = _lib.ACT_NONE) -> Tensor:
    """out[s] = act(sum|mean of src_act(src[perm[p]]) over the plan's segments) (g4c_segment_reduce)."""
    if torch.is_grad_enabled() and src.requires_grad:
        if out is not None:
            raise NotImplementedError("segment_reduce(out=...) is not differentiable")
        from . import autograd as _ag
        return _ag.segment_reduce(src, csr, mean, act, src_act)
    lib = _lib.load()
    src = _f32_2d(src, "src")
    dev = _lib.require_hip(src, csr.off, csr.perm)
    width = int(src.size(1))
    if out is None:
        out = torch.empty((csr.n_seg, width), dtype=torch.float32, device=dev)
    # algorithmic bytes: messages read + rows written + offsets (+ permutation)
    nbytes = 4.0 * (csr.n * width + csr.n_seg * width + csr.n_seg + 1 + (csr.n if csr.perm is not None else 0))
    _timed("segment_reduce", 0.0, nbytes, lambda: _lib.check(lib.g4c_segment_reduce(
        _lib.ptr(src), _ld(src), _lib.ptr(csr.perm), _lib.ptr(csr.off), csr.n_seg, width,
        1 if mean else 0, src_act, act, _lib.ptr(out), _ld(out), _lib.stream_handle(dev))))
    return out


def activation_(x: Tensor, act: int) -> Tensor:
    """In-place activation of a contiguous tensor (g4c_activation_inplace)."""
    lib = _lib.load()
    dev = _lib.require_hip(x)
    if not x.is_contiguous():
        raise ValueError("activation_: tensor must be contiguous")
    _lib.check(lib.g4c_activation_inplace(_lib.ptr(x), x.numel(), act, _lib.stream_handle(dev)))
    return x


def weighted_segment_mean(x: Tensor, x_idx32: Tensor, w: Tensor, csr: CsrPlan, out: Optional[Tensor] = None,
                          out_idx32: Optional[Tensor] = None) -> Tensor:
    if torch.is_grad_enabled() and x.requires_grad:
        if out is not None:
            raise NotImplementedError("weighted_segment_mean(out=...) is not differentiable (autograd.weighted_segment_mean "
                                      "returns the full tensor)")
        from . import autograd as _ag
        return _ag.weighted_segment_mean(x, x_idx32, w, csr)
    lib = _lib.load()
    x = _f32_2d(x, "x")
    w = w.reshape(-1).contiguous()
    dev = _lib.require_hip(x, x_idx32, w, csr.off)
    if csr.perm is not None:
        raise NotImplementedError("knn_interpolate needs y_idx sorted (the BuildKnnInterpWeights layout)")
    width = int(x.size(1))
    if out is None:
        out = torch.empty((csr.n_seg, width), dtype=torch.float32, device=dev)
    _lib.check(lib.g4c_weighted_segment_mean(_lib.ptr(x), _ld(x), _lib.ptr(x_idx32), _lib.ptr(w), _lib.ptr(csr.off),
                                             csr.n_seg, width, _lib.ptr(out), _ld(out), _lib.ptr(out_idx32),
                                             _lib.stream_handle(dev)))
    return out


def project_to_edges(v: Tensor, node32: Optional[Tensor], unit: Tensor, n_edges: int, n_feat: int) -> Tensor:
    if torch.is_grad_enabled() and v.requires_grad:
        from . import autograd as _ag
        return _ag.project_to_edges(v, node32, unit, n_edges, n_feat)
    lib = _lib.load()
    v = _f32_2d(v, "v")
    unit = _f32_2d(unit, "edgeUnitVector").contiguous()
    dev = _lib.require_hip(v, node32, unit)
    out = torch.empty((n_edges, n_feat), dtype=torch.float32, device=dev)
    _lib.check(lib.g4c_project_to_edges(_lib.ptr(v), _ld(v), _lib.ptr(node32), _lib.ptr(unit), n_edges, n_feat,
                                        _lib.ptr(out), _ld(out), _lib.stream_handle(dev)))
    return out


def edge_scalar_to_node_vector(e: Tensor, unit_inv: Tensor, n_nodes: int, k: int, out: Optional[Tensor] = None) -> Tensor:
    if torch.is_grad_enabled() and e.requires_grad:
        if out is not None:
            raise NotImplementedError("edge_scalar_to_node_vector(out=...) is not differentiable")
        from . import autograd as _ag
        return _ag.edge_scalar_to_node_vector(e, unit_inv, n_nodes, k)
    lib = _lib.load()
    e = _f32_2d(e, "edge_attr")
    unit_inv = unit_inv.contiguous()
    dev = _lib.require_hip(e, unit_inv, out)
    n_feat = int(e.size(1))
    if int(e.size(0)) != n_nodes * k:
        raise ValueError(f"{int(e.size(0))} edges cannot be viewed as {n_nodes} nodes x {k} incoming edges")
    if out is None:
        out = torch.empty((n_nodes, 2 * n_feat), dtype=torch.float32, device=dev)
    elif tuple(out.shape) != (n_nodes, 2 * n_feat) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError(f"out must be a float32 [{n_nodes}, {2 * n_feat}] tensor with unit column stride")
    _lib.check(lib.g4c_edge_scalar_to_node_vector(_lib.ptr(e), _ld(e), _lib.ptr(unit_inv), k, n_nodes, n_feat,
                                                  _lib.ptr(out), _ld(out), _lib.stream_handle(dev)))
    return out


def copy_cols(src: Tensor, dst: Tensor, dcol0: int, scol0: int = 0, width: Optional[int] = None,
              idx32: Optional[Tensor] = None, n_rows: Optional[int] = None) -> None:
    lib = _lib.load()
    dev = _lib.require_hip(src, dst, idx32)
    width = int(src.size(1) - scol0 if width is None else width)
    n_rows = int(dst.size(0) if n_rows is None else n_rows)
    _lib.check(lib.g4c_copy_cols(_lib.ptr(src), _ld(src), scol0, _lib.ptr(idx32), _lib.ptr(dst), _ld(dst), dcol0, width,
                                 n_rows, _lib.stream_handle(dev)))


def add_cols(a: Tensor, a_col0: int, b: Tensor, out: Tensor) -> Tensor:
    """out = a[:, a_col0:a_col0+w] + b (g4c_add_cols)."""
    lib = _lib.load()
    dev = _lib.require_hip(a, b, out)
    _lib.check(lib.g4c_add_cols(_lib.ptr(a), _ld(a), a_col0, _lib.ptr(b), _ld(b), _lib.ptr(out), _ld(out),
                                int(b.size(1)), int(b.size(0)), _lib.stream_handle(dev)))
    return out


def layer_norm(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float, act: int = _lib.ACT_NONE,
               out: Optional[Tensor] = None) -> Tensor:
    """Row-wise LayerNorm (+ activation) over any width (g4c_layer_norm): the fused MLP kernels' own epilogue covers <= 128 columns."""
    lib = _lib.load()
    x = _f32_2d(x, "x")
    dev = _lib.require_hip(x, gamma, beta, out)
    if out is None:
        out = torch.empty((int(x.size(0)), int(x.size(1))), dtype=torch.float32, device=dev)
    _lib.check(lib.g4c_layer_norm(_lib.ptr(x), _ld(x), int(x.size(0)), int(x.size(1)), _lib.ptr(gamma), _lib.ptr(beta), float(eps), int(act),
                                  _lib.ptr(out), _ld(out), _lib.stream_handle(dev)))
    return out


def rollout_advance(field: Tensor, pred: Tensor, outputs: Tensor, step: Tensor, nf: int) -> None:
    lib = _lib.load()
    dev = _lib.require_hip(field, pred, outputs, step)
    assert field.is_contiguous() and pred.is_contiguous() and outputs.is_contiguous()
    if step.dtype != torch.int32 or step.numel() < 2:
        raise ValueError("rollout_advance: `step` is an int32 tensor of two entries — [step index, the launch's ticket counter (zero)]")
    if outputs.dim() == 3:          # step-major [steps, n_nodes, nf]: a step writes one contiguous block
        if outputs.size(1) != field.size(0) or outputs.size(2) != nf:
            raise ValueError(f"rollout_advance: step-major outputs {tuple(outputs.shape)} for {field.size(0)} nodes x {nf} fields")
        out_ld = 0
    else:
        out_ld = int(outputs.size(1))
    _lib.check(lib.g4c_rollout_advance(_lib.ptr(field), int(field.size(1)), _lib.ptr(pred), nf, _lib.ptr(outputs),
                                       out_ld, _lib.ptr(step), int(field.size(0)), _lib.stream_handle(dev)))


def steps_to_columns(out_steps: Tensor) -> Tensor:
    """Step-major rollout outputs [steps, n_nodes, nf] -> the reference's layout [n_nodes, nf * steps] (nn/model.py:322-326)."""
    return out_steps.permute(1, 0, 2).reshape(out_steps.size(1), -1)


# Arithmetic of the fused MLPs — all fp32 in, fp32 out:
#   "f16x3" (default)   fp32-class products on the f16 matrix pipe: both operands split two ways into fp16 terms,
#                       x = h + l * 2^-11 (22 significand bits), three partial products, the 2^-11 terms in their own fp32
#                       accumulator (g4c_mlp_pack_layer_f16x3 + the g4c_mlp_forward_bx6* entry points); measured error against fp64
#                       below that of the fp32-MFMA kernel (scripts/mlp_accuracy.py, test_mlp_precisions_vs_fp64).  Range: fp16's — an MLP
#                       input or hidden activation beyond +-65504 is clipped there (no inf / NaN) AND FLAGGED on the device; the
#                       arithmetic is run optimistically: a rollout reads its flags where it hands out results (Rollout.validate) and
#                       recomputes itself in "bf16x6" if anything was clipped, so solve() never returns a clipped value (normalised CFD
#                       fields and LayerNorm'd latents are far inside the range; raw inputs only pass the fp32 vector path);
#   "bf16x6"            the same kernels with both operands split EXACTLY into three bf16 terms (fp32 exponent range), the six
#                       largest partial products accumulated in fp32 (g4c_mlp_forward_bx6): twice the matrix-pipe work;
#                       MLPs outside the envelope of these two (an input block wider than 128) use the fp32 kernels;
#   "fp32"              v_mfma_f32_32x32x2_f32 (g4c_mlp_forward): the original kernel;
#   "bf16"              operands ROUNDED to bf16 (~1e-2 deviation): BASELINE config 3's "bf16 edge-MLP MFMA", opt-in only.
PRECISIONS = ("fp32", "bf16", "bf16x6", "f16x3")
_PRECISION = os.environ.get("G4C_MLP_PRECISION", "f16x3")


# Fused aggregation in the edge-MLP launch (g4c_mlp_forward_bx6_agg): bit-identical to the separate g4c_segment_reduce, and the
# node launch no longer re-reads the 307 MB of messages (the largest single stream of a level-1 MP layer after the messages'
# own write).  The launch then runs on tiles of WHOLE segments: with in-degree 6 (or 5) a 32-row tile holds 30 rows, i.e.
# 6.7 % more tiles — speed-neutral on the 100k rollout (+0.8 %), 2.5 GB less traffic per step.  On from FUSE_AGG_MIN_ROWS
# rows (below, launches are latency-bound and the row tiles of whole segments only cost).  (Module attributes, not environment
# switches: the A/B runs that fixed them are in DESIGN.md / HISTORY.md; tests flip them directly.)
FUSE_AGG = True
FUSE_AGG_MIN_ROWS = 50000
# Aggregation on load (g4c_src_t.seg_off): the node-MLP launch averages each target's messages while it gathers its input,
# instead of a separate g4c_segment_reduce pass (bit-identical values; no tile-alignment constraint, unlike FUSE_AGG).
AGG_ON_LOAD = True
# below this many rows to aggregate, the launch that gathers them is latency-bound and a separate g4c_segment_reduce is quicker
# (same-box sweep: 12.5k-node 3-scale mesh 789 -> 822 steps/s, 2-scale 10k nodes 1079 -> 1110, 25k nodes 503 -> 516 with the
# threshold at 50k rows; neutral at 100k nodes)
AGG_ON_LOAD_MIN_ROWS = 50000


# ---- fp16 range of the "f16x3" arithmetic made observable (g4c_mlp_t.range_flag): every launch in that arithmetic carries a slot
# of a per-device int32 array; a kernel that converted a value of magnitude >= 65504 to fp16 (it was clipped there) writes 1 into
# its slot.  Slots are named after the MLP that launched ("NsThreeScaleGNN.mp112.edge_mlp"); f16_range_report() reads the array
# (one device synchronisation), check_f16_range() turns a non-empty report into a RuntimeWarning.  Rollout.validate (called by
# Rollout.result, hence Model.solve) and DistributedRollout.validate (gather_outputs) read it and RECOMPUTE a clipped rollout in "bf16x6";
# GNN.fit warns per epoch; a rollout clears its own model's slots on entry and looks at those only, so a clip is attributed to the
# model that launched it (RangeWatch below: every reader hands hits to all live consumers before it clears).  A bare model.forward()
# validates itself the same way (nn/model.py: GNN.__init_subclass__): one flag read per call, the forward again in "bf16x6" if it clipped.
RANGE_SLOTS = 4096
_range_bufs = {}            # device -> int32 [RANGE_SLOTS]
_range_sites: List[set] = [set() for _ in range(RANGE_SLOTS)]
_range_slot_of = {}
_range_wrapped = False


def _range_buffer(dev: torch.device) -> Tensor:
    buf = _range_bufs.get(dev)
    if buf is None:
        buf = _range_bufs[dev] = torch.zeros(RANGE_SLOTS, dtype=torch.int32, device=dev)
    return buf


def _range_slot(site: str) -> int:
    global _range_wrapped
    slot = _range_slot_of.get(site)
    if slot is None:
        n = len(_range_slot_of)
        if n >= RANGE_SLOTS and not _range_wrapped:
            _range_wrapped = True
            import warnings
            warnings.warn(f"more than {RANGE_SLOTS} named MLP sites: fp16 range flags are shared between sites from here on "
                          "(a clip is then reported under every name that shares its slot)", RuntimeWarning)
        slot = _range_slot_of[site] = n % RANGE_SLOTS
        _range_sites[slot].add(site)
    return slot


def _indexed(device) -> Optional[torch.device]:
    """torch.device('cuda') / 'cuda' name the CURRENT device: the flag buffers are keyed by indexed devices."""
    if device is None:
        return None
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def range_slots_of(sites) -> List[int]:
    """Flag slots of the named MLP sites that have launched in the f16x3 arithmetic so far (others have no slot yet)."""
    return sorted({_range_slot_of[s] for s in sites if s in _range_slot_of})


class RangeWatch:
    """One consumer of the fp16 range flags: a `Rollout`, a `DistributedRollout`, a validated bare `forward()`.  The flags are one
    int per MLP site and device, shared by everything that launches that MLP; whoever reads them (`f16_range_poll`: one
    synchronisation) hands every hit to ALL live watches that name the site and only then clears the device array — so a second
    rollout of the same model, a bare forward between two steps or a `check_f16_range()` call can no longer erase the evidence
    another rollout still has to act on.  A hit that cannot be told apart (two live consumers of one site) reaches
    both: each recomputes in the exact arithmetic, which is always right.  `drain=True` (a rollout's entry): the flags are read
    once first, so that what earlier launches of these MLPs left behind goes to the watches that were live then, not to this one."""
    _live = weakref.WeakSet()

    def __init__(self, device, sites=None, drain: bool = True):
        self.device = _indexed(device)
        self.sites = None if sites is None else frozenset(sites)
        self.hits = set()
        if drain:
            f16_range_poll(self.device)
        RangeWatch._live.add(self)

    def wants(self, site: str) -> bool:
        return self.sites is None or site in self.sites

    def take(self) -> List[str]:
        """Names of this watch's MLPs whose launches clipped a value since the last take (synchronises with the device)."""
        f16_range_poll(self.device)
        out = sorted(self.hits)
        self.hits.clear()
        return out

    def close(self) -> None:
        RangeWatch._live.discard(self)


_range_unclaimed = set()            # (device, site) of hits no live watch asked for: what f16_range_report / check_f16_range answer from


def f16_range_poll(device: Optional[torch.device] = None) -> List[str]:
    """Read the flag array(s) (one synchronisation per device), distribute the hits to the live watches, clear the array(s).
    Returns every name that was set."""
    device = _indexed(device)
    names_out: List[str] = []
    for dev, buf in list(_range_bufs.items()):
        if device is not None and device != dev:
            continue
        flags = buf.cpu()
        slots = torch.nonzero(flags).flatten().tolist()
        if not slots:
            continue
        buf.zero_()
        watches = [w for w in list(RangeWatch._live) if w.device is None or w.device == dev]
        for slot in slots:
            for name in sorted(_range_sites[slot]) or [f"slot {slot}"]:
                names_out.append(name)
                claimed = False
                for w in watches:
                    if w.wants(name):
                        w.hits.add(name)
                        claimed = True
                if not claimed:
                    _range_unclaimed.add((dev, name))
    return names_out


def f16_range_clear(device: Optional[torch.device] = None, sites=None) -> None:
    """Forget the recorded clips that no live watch is waiting for: every site, or only `sites` (names as in f16_range_report).
    Synchronises (the flags are read first: what a live `Rollout` still has to see reaches it)."""
    f16_range_report(device, clear=True, sites=sites)


def f16_range_report(device: Optional[torch.device] = None, clear: bool = True, sites=None) -> List[str]:
    """Names of the MLPs whose launches clipped a value at the end of the fp16 range and that no rollout / validated forward has
    dealt with, since the last report / clear (all devices, or one; `sites`: only these names are looked at and cleared).
    Synchronises with the device(s)."""
    f16_range_poll(device)
    device = _indexed(device)
    only = None if sites is None else set(sites)
    mine = sorted(k for k in _range_unclaimed if (device is None or k[0] == device) and (only is None or k[1] in only))
    if clear:
        _range_unclaimed.difference_update(mine)
    return [name for _, name in mine]


def check_f16_range(device: Optional[torch.device] = None, where: str = "", sites=None) -> List[str]:
    """RuntimeWarning when a launch in the "f16x3" arithmetic clipped an MLP input or hidden activation at +-65504 (the reference
    computes these in fp32: nn/model.py:303-321).  Returns the offending MLPs' names.  `sites`: see f16_range_report."""
    hit = f16_range_report(device, sites=sites)
    if hit:
        import warnings
        warnings.warn(f"{where + ': ' if where else ''}the 'f16x3' MLP arithmetic clipped values at the end of the fp16 range (|x| >= 65504) in "
                      f"{', '.join(hit[:8])}{' ...' if len(hit) > 8 else ''}: the result differs from an fp32 evaluation.  "
                      "gfd.set_mlp_precision('bf16x6') keeps fp32's exponent range.", RuntimeWarning, stacklevel=2)
    return hit


_weights_epoch = 0


def weights_epoch() -> int:
    """Counter folded into the validity key of every packed-weight image (nn/blocks.py MLP._signature).  The key otherwise
    relies on the parameters' version counters, which in-place updates that bypass autograd's bookkeeping do not advance —
    torch's own `fused=True` optimizers among them (measured: `Adam(fused=True).step()` leaves `p._version` unchanged).  Every
    backward pass through a fused MLP advances the epoch, so the forward after a training step always repacks."""
    return _weights_epoch


def bump_weights_epoch() -> None:
    global _weights_epoch
    _weights_epoch += 1


def grad_mode() -> bool:
    """True when calls are being recorded for autograd: the block / model code then keeps to the plain forms of the
    launches (no heads, no pre-multiplied products, no in-place epilogues), which are the differentiable ones."""
    return torch.is_grad_enabled()


def can_fuse_aggregation(csr: CsrPlan, width: int) -> bool:
    """The edge launch itself can reduce its rows per target (g4c_mlp_forward_bx6_agg): rows in segment order, segments of
    at most 32 rows, the exact-split kernels, a 128-wide output, enough rows to be throughput-bound."""
    return (FUSE_AGG and _PRECISION in ("bf16x6", "f16x3", "bf16") and width == 128 and csr.perm is None and csr.n >= FUSE_AGG_MIN_ROWS
            and not grad_mode() and csr.tiles() is not None)


def can_aggregate_on_load(csr: CsrPlan, width: int, consumer_widths: Sequence[int]) -> bool:
    """`consumer_widths`: input blocks of the MLP that would aggregate while loading (it must run on the bf16x6 kernels)."""
    # (rows in segment order only: through a permutation the kernel supports it too — Source(segments=csr with perm) — but the
    # extra dependent index round trip in the consumer's prologue gives back what the separate reduction costs; measured neutral)
    return (AGG_ON_LOAD and effective_precision(consumer_widths) != "fp32" and csr.perm is None and width == 128
            and csr.n > 0 and csr.n >= AGG_ON_LOAD_MIN_ROWS)


def mlp_precision() -> str:
    return _PRECISION


def effective_precision(seg_widths: Sequence[int]) -> str:
    """Precision an MLP with these input blocks is packed for: the selected one, or fp32 outside the bf16 kernels' envelope."""
    if _PRECISION != "fp32" and any(int(w) > 128 for w in seg_widths):
        return "fp32"
    return _PRECISION


def set_mlp_precision(precision: str) -> str:
    """Select the arithmetic of every fused MLP launched from now on; returns the previous setting."""
    global _PRECISION
    if precision not in PRECISIONS:
        raise ValueError(f"unknown MLP precision {precision!r} ({' | '.join(PRECISIONS)})")
    old, _PRECISION = _PRECISION, precision
    return old


def _rs_k_order(dev) -> Tensor:
    """Column order of the row-split kernel's stream and of its bf16 rows (G4C_ROWS_RS_ORDER): position 32 j + 8 g + 4 h + e holds
    feature 32 j + 16 h + 4 g + e — the eight values a lane of a 16x16x32 MFMA holds of a 32-feature step, side by side."""
    pos = torch.arange(128, device=dev)
    j, g, h, e = pos // 32, (pos % 32) // 8, (pos % 8) // 4, pos % 4
    return 32 * j + 16 * h + 4 * g + e


class RsOrderedRows(Tensor):
    """bf16 [n, 128] rows whose columns are in the row-split kernel's order (`_rs_k_order`): the compact message rows and the hoisted
    product tables its launches exchange (mlp_rs.hip, rounded-bf16 mode).  The subclass is only a tag that travels with the tensor
    (row slices and views keep it): `mlp_forward` hands such rows to that kernel as they are and restores the natural column order
    for every other reader."""

    @staticmethod
    def tag(t: Tensor) -> Tensor:
        if t.dtype != torch.bfloat16 or t.dim() != 2 or t.size(1) != 128:
            raise ValueError("RsOrderedRows: bf16 [n, 128] rows only")
        return t.as_subclass(RsOrderedRows)


def rs_rows_to_natural(t: Tensor) -> Tensor:
    """The rows of an RsOrderedRows tensor with their columns in feature order (a copy, plain Tensor)."""
    inv = torch.argsort(_rs_k_order(t.device))
    return t.as_subclass(Tensor).index_select(1, inv)


class PackedMLP:
    """Device-side packed weights of one MLP for a given input block structure."""

    def __init__(self, weights: Sequence[Tensor], biases: Sequence[Tensor], ln: Optional[Tuple[Tensor, Tensor, float]],
                 seg_widths: Sequence[int], seg_negate: Sequence[bool], heads: Sequence[Tensor] = (),
                 precision: str = "fp32", narrow: Optional[Sequence[bool]] = None, site: Optional[str] = None, rs_order: bool = False,
                 rs_blocks: Optional[Sequence[bool]] = None, rs2: int = 0):
        """`heads`: bias-free [128, 128] weights applied to the MLP's final output row (g4c_mlp_forward_heads); their
        packed images continue the weight stream after the last layer.  `precision` "bf16": the bf16 stream of
        g4c_mlp_pack_layer_bx6 (every input block padded to 128 k; "bf16" uses the same stream, leading plane only).
        `narrow[s]` (bf16x6 / bf16 only): input block s (<= 8 columns, read without index or activation) is multiplied in
        fp32 on the vector ALUs by its rows of the first layer's weight (g4c_src_t.additive == 2) instead of being padded
        to a 128-k block of the matrix-pipe stream."""
        lib = _lib.load()
        # "f16x3" is the "bf16x6" kernel family (same entry points, stream layout and launch envelope) on a stream written by
        # g4c_mlp_pack_layer_f16x3; desc.w_format tells the library which arithmetic the stream is for
        self.split = "f16x2" if precision == "f16x3" else ("bf16x3" if precision == "bf16x6" else None)
        if precision == "f16x3":
            precision = "bf16x6"
        self.precision = precision
        bf16 = precision in ("bf16", "bf16x6")
        narrow = tuple(bool(x) for x in narrow) if narrow is not None else (False,) * len(seg_widths)
        if any(narrow) and not bf16:
            raise NotImplementedError("narrow input blocks need the bf16x6 kernels")
        if any(nw and w > _lib.NARROW_MAX for nw, w in zip(narrow, seg_widths)):
            raise ValueError(f"a narrow input block has more than {_lib.NARROW_MAX} columns")
        self.narrow = narrow       # three-plane bf16 weight stream, 128-k input blocks ("bf16" reads plane 0 only)
        planes = 3
        dev = _lib.require_hip(*weights, *[b for b in biases if b is not None], *heads)
        n_layers = len(weights)
        if not 1 <= n_layers <= _lib.MAX_LAYERS:
            raise NotImplementedError(f"MLP with {n_layers} Linear layers (supported: 1..{_lib.MAX_LAYERS})")
        if len(seg_widths) > _lib.MAX_SRC:
            raise NotImplementedError(f"MLP input concatenated from {len(seg_widths)} blocks (max {_lib.MAX_SRC})")
        self.desc = _lib.g4c_mlp_t()
        self.desc.n_layers = n_layers
        # `rs_order` (rounded-bf16 mode, every layer 128 x 128 over ONE 128-wide input block): the stream is written for the row-split
        # kernel (mlp_rs.hip, G4C_WFMT_BF16_RS) — the columns of every layer in the order `_rs_k_order`, the order in which a 16x16x32
        # MFMA leaves a layer's output in the lane that needs it as the next layer's operand.  No other kernel can read such a stream:
        # the library fails a launch outside that kernel's envelope.
        self.rs_order = bool(rs_order)
        if self.rs_order and (precision != "bf16" or heads or len(seg_widths) != 1 or seg_widths[0] != 128 or any(narrow or ())
                              or any(tuple(W.shape) != (128, 128) for W in weights)):
            raise NotImplementedError("rs_order: rounded-bf16 mode, one 128-wide input block, 128 x 128 layers, no heads")
        # `rs2` (4 / 5 = G4C_WFMT_BF16_RS2 / _RS2N): the stream of the row-split UPDATE kernel (mlp_rs2_kernel): two 128-wide input
        # blocks, two layers, optional heads — the columns of every 128-wide block of every layer and head in the order `_rs_k_order`
        self.rs2 = int(rs2)
        if self.rs2 and (self.rs2 not in (4, 5) or precision != "bf16" or self.rs_order or tuple(seg_widths) != (128, 128) or any(narrow or ())
                         or len(weights) != 2 or tuple(weights[0].shape) != (128, 256) or tuple(weights[1].shape) != (128, 128)
                         or len(heads) not in (0, 2) or any(tuple(h.shape) != (128, 128) for h in heads) or any(seg_negate)):
            raise NotImplementedError("rs2: rounded-bf16 mode, [128 | 128] -> 128 -> 128, no or two 128 x 128 heads")
        self.desc.w_format = 1 if self.split == "f16x2" else (3 if self.rs_order else (self.rs2 if self.rs2 else 0))
        # `rs_blocks[j]` (rounded-bf16 mode): input block j arrives as RsOrderedRows — bf16 rows in the row-split kernel's column order
        # (its aggregate, G4C_AGG_OUT_BF16) — and is read by a kernel that knows nothing of that order: the COLUMNS of the first layer's
        # block j are packed in the same order instead, the product is the same sum in another order.
        self.rs_blocks = tuple(bool(x) for x in rs_blocks) if rs_blocks is not None else (False,) * len(seg_widths)
        if any(self.rs_blocks) and (self.rs_order or precision != "bf16" or any(narrow) or len(self.rs_blocks) != len(seg_widths)
                                    or any(r and w != 128 for r, w in zip(self.rs_blocks, seg_widths))):
            raise NotImplementedError("rs_blocks: rounded-bf16 mode, 128-wide blocks, no narrow blocks")
        # (`site`: the name a clipped value is reported under — f16_range_report)
        self.site = site or "an MLP created outside a model"
        if self.split == "f16x2":
            self.desc.range_flag, self.desc.range_slot = _range_buffer(dev).data_ptr(), _range_slot(self.site)
        else:
            self.desc.range_flag, self.desc.range_slot = None, 0
        self._keep: List[Tensor] = []
        stream = _lib.stream_handle(dev)
        KC, NP = 32, 128                       # kernel constants: K chunk, computed layer width
        if bf16 and any(s > NP for s in seg_widths):
            raise NotImplementedError("bf16 MLP with an input block wider than 128")
        wide = [j for j in range(len(seg_widths)) if not narrow[j]]            # the blocks that go through the packed stream
        k_pad0 = NP * len(wide) if bf16 else sum((s + KC - 1) // KC * KC for s in seg_widths)
        k_pads = [k_pad0] + [NP] * (n_layers - 1)
        # one contiguous weight stream (layer after layer, 32-k chunk after chunk) + one chunk of slack:
        # the kernel's register ring prefetches one chunk past the end
        if len(heads) > _lib.MAX_HEADS or any(tuple(h.shape) != (NP, NP) for h in heads):
            raise NotImplementedError(f"heads must be at most {_lib.MAX_HEADS} weights of shape [128, 128]")
        if heads and int(weights[-1].size(0)) != NP:
            raise NotImplementedError("heads need a 128-wide MLP output")
        # (bf16: 2-byte elements, one 128-k block of slack)
        stream_buf = torch.zeros((sum(k_pads) + NP * len(heads) + (NP if bf16 else KC)) * NP * planes,
                                 dtype=torch.bfloat16 if bf16 else torch.float32, device=dev)
        esz = 2 * planes if bf16 else 4
        pack = (lib.g4c_mlp_pack_layer_f16x3 if self.split == "f16x2" else lib.g4c_mlp_pack_layer_bx6) if bf16 else lib.g4c_mlp_pack_layer
        bias_buf = torch.zeros(n_layers * NP, dtype=torch.float32, device=dev)
        self._keep += [stream_buf, bias_buf]
        off = 0
        for l, (W, b) in enumerate(zip(weights, biases)):
            n_out, k_in = int(W.size(0)), int(W.size(1))
            if n_out > NP:
                raise NotImplementedError(f"layer width {n_out} > 128 is outside the fused-MLP kernel envelope")
            if l == 0:
                segs, negs = list(seg_widths), [1 if x else 0 for x in seg_negate]
                self.narrow_w = {}
                if any(narrow):          # fp32 rows of W1^T for the narrow blocks; the stream gets the remaining columns only
                    col0s = [sum(seg_widths[:j]) for j in range(len(seg_widths))]
                    W32 = W.detach().to(torch.float32)
                    for j in range(len(seg_widths)):
                        if narrow[j]:
                            rows = torch.zeros((seg_widths[j], NP), dtype=torch.float32, device=dev)
                            rows[:, :n_out] = W32[:, col0s[j]:col0s[j] + seg_widths[j]].T * (-1.0 if seg_negate[j] else 1.0)
                            self.narrow_w[j] = rows.contiguous()
                            self._keep.append(self.narrow_w[j])
                    if wide:
                        W = torch.cat([W32[:, col0s[j]:col0s[j] + seg_widths[j]] for j in wide], dim=1)
                        segs, negs = [seg_widths[j] for j in wide], [1 if seg_negate[j] else 0 for j in wide]
                        k_in = int(W.size(1))
                    else:
                        W = None
            else:
                prev = int(weights[l - 1].size(0))
                if k_in != prev:
                    raise ValueError(f"layer {l + 1} expects {k_in} inputs, previous layer gives {prev}")
                segs, negs = [k_in], [0]
            wptr = stream_buf.data_ptr() + esz * off
            if W is not None:            # (None: every input block of the first layer is narrow, nothing to stream)
                Wc = W.detach().to(torch.float32).contiguous()
                if self.rs_order:
                    Wc = Wc[:, _rs_k_order(dev)].contiguous()
                if self.rs2:
                    o = _rs_k_order(dev)
                    Wc = Wc[:, torch.cat([c0 + o for c0 in range(0, k_in, 128)])].contiguous()
                if l == 0 and any(self.rs_blocks):
                    cols = torch.arange(k_in, device=dev)
                    for j, r in enumerate(self.rs_blocks):
                        if r:
                            c0 = sum(seg_widths[:j])
                            cols[c0:c0 + 128] = c0 + _rs_k_order(dev)
                    Wc = Wc[:, cols].contiguous()
                seg_arr = (C.c_int32 * len(segs))(*segs)
                neg_arr = (C.c_int32 * len(segs))(*negs)
                _lib.check(pack(_lib.ptr(Wc), n_out, k_in, seg_arr, neg_arr, len(segs), wptr, k_pads[l], NP, stream))
            if b is not None:
                bias_buf[l * NP: l * NP + n_out].copy_(b.detach())
            self.desc.k_pad[l], self.desc.n_pad[l] = k_pads[l], NP
            self.desc.w[l], self.desc.b[l] = wptr, bias_buf.data_ptr() + 4 * l * NP
            off += k_pads[l] * NP
        self.n_heads = len(heads)
        self.head_w = stream_buf.data_ptr() + esz * off if heads else None
        one = (C.c_int32 * 1)(NP)
        zero = (C.c_int32 * 1)(0)
        for W in heads:
            Wc = W.detach().to(torch.float32).contiguous()
            if self.rs2:
                Wc = Wc[:, _rs_k_order(dev)].contiguous()
            _lib.check(pack(_lib.ptr(Wc), NP, NP, one, zero, 1, stream_buf.data_ptr() + esz * off, NP, NP, stream))
            off += NP * NP
        self.n_out = int(weights[-1].size(0))
        self.desc.n_out = self.n_out
        # nn.Linear MACs x 2 (+ the heads' products)
        self.flops_per_row = float(sum(2 * int(W.size(0)) * int(W.size(1)) for W in list(weights) + list(heads)))
        if ln is not None:
            g, be, eps = ln
            g, be = g.detach().to(torch.float32).contiguous(), be.detach().to(torch.float32).contiguous()
            self.desc.ln_gamma, self.desc.ln_beta, self.desc.ln_eps = g.data_ptr(), be.data_ptr(), float(eps)
            self._keep += [g, be]
        else:
            self.desc.ln_gamma, self.desc.ln_beta, self.desc.ln_eps = None, None, 0.0
        self.seg_widths = tuple(seg_widths)
        self.device = dev
        # the parameter tensors this image was packed from (the training path differentiates with respect to them: autograd.py)
        self.params = (list(weights), list(biases), ln)
        self.heads_params = bool(heads)


def _src_array(sources: Sequence[Source]):
    arr = (_lib.g4c_src_t * len(sources))()
    for a, s in zip(arr, sources):
        a.ptr, a.idx, a.width, a.ld, a.col0, a.pre_act = (s.tensor.data_ptr(), _lib.ptr(s.index), s.width, _ld(s.tensor),
                                                          s.col0, s.pre_act)
        a.additive = 1 if s.additive else 0
        a.dtype = 1 if s.tensor.dtype == torch.bfloat16 else 0
        if s.segments is not None:
            a.seg_off, a.seg_mean, a.seg_perm = _lib.ptr(s.segments.off), 1 if s.seg_mean else 0, _lib.ptr(s.segments.perm)
    return arr


def _agg_mode(agg_mean: bool, csr: CsrPlan) -> int:
    """`agg_mean` argument of the fused-aggregation launches: bit 0 = mean, G4C_AGG_UNIFORM(k) when every segment has k rows."""
    return (1 if agg_mean else 0) | (csr.uniform_deg << 8)


def mp_layer_forward(msg: PackedMLP, sources: Sequence[Source], n_rows: int, csr: CsrPlan, agg_mean: bool, upd: PackedMLP,
                     v: Tensor, act: int, store_rows: bool = True, head_outs: Optional[Sequence[Tensor]] = None,
                     v_out: Optional[Tensor] = None) -> Tuple[Optional[Tensor], Tensor, Optional[Sequence[Tensor]]]:
    """One launch for a whole MP layer (g4c_mp_layer_forward_bx6; reference nn/blocks.py:175-186): the hoisted message MLP `msg`
    on `sources` (one 128-wide weighted block + the two gathered node-side products as additive sources) with the aggregation over
    `csr`, and — in the same persistent workgroups — the node MLP `upd` on [aggregate | v] with LayerNorm / `act` and, when
    `head_outs` is given, the heads packed behind `upd` (the next layer's products).  f16x3 arithmetic only, inference only.
    Returns (e' rows or None, v', head_outs)."""
    lib = _lib.load()
    dev = _lib.require_hip(*[s.tensor for s in sources], *[s.index for s in sources], v, v_out)
    if msg.split != "f16x2" or upd.split != "f16x2":
        raise NotImplementedError("mp_layer_forward needs both MLPs packed for the f16x3 arithmetic")
    tiles = csr.tiles()
    if tiles is None or n_rows != csr.n:
        raise ValueError("mp_layer_forward: the rows must be in segment order with segments of at most 32 rows")
    v = _f32_2d(v, "v")
    n_t = csr.n_seg
    t_rows, t_seg, nt = tiles
    e_out = torch.empty((n_rows, 128), dtype=torch.float32, device=dev) if store_rows else None
    agg = torch.empty((n_t, 128), dtype=torch.float32, device=dev)          # (scratch: written and re-read by the same workgroup, L2-resident)
    if v_out is None:
        v_out = torch.empty((n_t, 128), dtype=torch.float32, device=dev)
    n_heads = 0 if head_outs is None else len(head_outs)
    if n_heads and (n_heads != upd.n_heads):
        raise ValueError(f"{n_heads} head outputs for a packing with {upd.n_heads} heads")
    ho = (C.c_void_p * max(n_heads, 1))(*([h.data_ptr() for h in head_outs] if n_heads else [None]))
    arr = _src_array(sources)
    call = lambda: _lib.check(lib.g4c_mp_layer_forward_bx6(
        C.byref(msg.desc), arr, len(sources), n_rows, _lib.ptr(e_out), 128 if e_out is None else _ld(e_out),
        _lib.ptr(t_rows), _lib.ptr(t_seg), _lib.ptr(csr.off), nt, _lib.ptr(agg), _ld(agg), _agg_mode(agg_mean, csr),
        C.byref(upd.desc), _lib.ptr(v), _ld(v), act, _lib.ptr(v_out), _ld(v_out),
        upd.head_w if n_heads else None, n_heads, ho, _ld(head_outs[0]) if n_heads else 128, _lib.stream_handle(dev)))
    if KernelTimer.active is None:
        call()
    else:
        in_b = 4.0 * 128 * (n_rows * (2 if store_rows else 1) + n_t * (2 + n_heads))
        _timed("mlp_bx6_kernel", msg.flops_per_row * n_rows + upd.flops_per_row * n_t, in_b, call)
    return e_out, v_out, head_outs


def mlp_forward(packed: PackedMLP, sources: Sequence[Source], n_rows: int, act: int = _lib.ACT_NONE,
                out: Optional[Tensor] = None, out_idx32: Optional[Tensor] = None,
                resid: Optional[Tensor] = None, resid_col0: int = 0, tile_mode: Optional[int] = None,
                head_outs: Optional[Sequence[Tensor]] = None, agg: Optional[Tuple[CsrPlan, Tensor, bool]] = None,
                save: Optional[Sequence[Optional[Tensor]]] = None, mul: Optional[Sequence[Optional[Tensor]]] = None,
                store_rows: bool = True, rows_dtype: Optional[torch.dtype] = None, rows_act: int = _lib.ACT_NONE) -> Optional[Tensor]:
    """One fused MLP launch (g4c_mlp_forward).  `tile_mode` (tests / tuning) runs every row through
    g4c_mlp_forward_rows with that kernel variant instead of the library's own choice.
    `head_outs` ([n_rows, 128] tensors, one per head of `packed`): g4c_mlp_forward_heads.
    `agg` = (csr, out [n_seg, 128], mean): also aggregate the output rows over the segments of `csr` (rows must be in segment
    order) — inside the launch when the kernel can (g4c_mlp_forward_bx6_agg), otherwise with a g4c_segment_reduce afterwards.
    `save` (training forward, bf16x6 only): one [n_rows, 128] fp32 tensor (or None) per layer, receiving that layer's output rows
    (g4c_mlp_forward_bx6_save); `mul` (with `save`): per hidden layer the SELU-output rows whose slope multiplies that layer's
    result instead of bias + SELU (the backward chain of a block, see include/g4c.h).
    `rows_dtype=torch.bfloat16` (rounded-bf16 mode, with an aggregation the launch fuses; ignored otherwise): the output rows are
    stored as bf16 — their consumer rounds them to bf16 on load anyway, and the launch is HBM-bound on them; the aggregate stays fp32.
    `rows_act=ACT_SELU` (only together with bf16 rows): the stored rows are bf16(SELU(row)) — the activation their reader would apply
    on load, applied before the one rounding (G4C_DTYPE_BF16_SELU); the aggregate is taken from the un-activated fp32 rows.  Check
    `out.dtype == torch.bfloat16` on the result to know whether the rows came back compact and activated.
    With gradients enabled and a differentiable input / parameter, the call is recorded for autograd (autograd.py)."""
    if torch.is_grad_enabled():
        from . import autograd as _ag
        if _ag.wants_grad(packed, sources, resid):
            if out is not None or out_idx32 is not None or head_outs is not None or agg is not None or tile_mode is not None or save is not None:
                raise NotImplementedError("out= / heads / fused aggregation are inference-only forms of mlp_forward; "
                                          "call under torch.no_grad() or use the plain form")
            return _ag.mlp(packed, sources, n_rows, act, resid, resid_col0)
    lib = _lib.load()
    if packed.rs2:
        ok = [isinstance(s.tensor, RsOrderedRows) for s in sources]
        if len(sources) != 2 or not ok[0] or ok[1] != (packed.rs2 == 4) or any(s.tensor.dtype != torch.bfloat16 for s in sources):
            raise ValueError("weights packed for the row-split update kernel: [aggregate | e] as bf16 rows, the aggregate (and, format 4, e) RsOrderedRows")
    elif packed.rs_order and packed.precision == "bf16":
        # (the row-split kernel's rounded-bf16 stream: its bf16 rows are in ITS column order, nobody else's)
        if any(s.tensor.dtype == torch.bfloat16 and not isinstance(s.tensor, RsOrderedRows) for s in sources):
            raise ValueError("weights packed for the row-split kernel: bf16 rows must be RsOrderedRows (its column order)")
    elif any(isinstance(s.tensor, RsOrderedRows) for s in sources) or any(packed.rs_blocks):
        # tagged rows reach a kernel that reads feature order: fine where the pack has that block's columns in the same order
        # (`rs_blocks`), restored to feature order (a copy) anywhere else
        fixed, j = [], 0
        for s in sources:
            tagged = isinstance(s.tensor, RsOrderedRows)
            ok = (not s.additive) and j < len(packed.rs_blocks) and packed.rs_blocks[j] and s.col0 == 0 and s.width == 128
            if not s.additive:
                if j < len(packed.rs_blocks) and packed.rs_blocks[j] and not tagged:
                    raise ValueError(f"input block {j}: the weights expect rows in the row-split order (ops.RsOrderedRows)")
                j += 1
            fixed.append(s if (not tagged or ok) else
                         Source(rs_rows_to_natural(s.tensor), s.index, s.col0, s.width, s.negate, s.pre_act, s.additive, s.segments, s.seg_mean))
        sources = fixed
    dev = _lib.require_hip(*[s.tensor for s in sources], *[s.index for s in sources], out, out_idx32, resid)
    if dev != packed.device:
        raise RuntimeError(f"MLP weights on {packed.device}, inputs on {dev}")
    if tuple(s.width for s in sources if not s.additive) != packed.seg_widths:
        raise ValueError(f"input blocks {[s.width for s in sources if not s.additive]} do not match packed layout {packed.seg_widths}")
    arr = _src_array(sources)
    if any(packed.narrow):
        j = 0
        for a, src in zip(arr, sources):
            if src.additive:
                continue
            if packed.narrow[j]:
                if src.index is not None or src.pre_act != _lib.ACT_NONE:
                    raise ValueError("a narrow input block cannot be gathered through an index or activated on load")
                a.additive, a.w = 2, packed.narrow_w[j].data_ptr()
            j += 1
    # store_rows=False (with a fused aggregation only): the output rows are not written, only their aggregate
    fusable = (agg is not None and FUSE_AGG and packed.precision in ("bf16x6", "bf16") and packed.n_out == 128 and head_outs is None
               and tile_mode is None and out_idx32 is None and resid is None and n_rows == agg[0].n and agg[0].tiles() is not None)
    if not store_rows and not fusable:
        raise ValueError("store_rows=False needs an aggregation the launch can fuse (ops.can_fuse_aggregation)")
    rows16 = bool(fusable and store_rows and rows_dtype == torch.bfloat16 and packed.precision == "bf16" and out is None)
    if out is None and store_rows:
        out = torch.empty((n_rows, packed.n_out), dtype=torch.bfloat16 if rows16 else torch.float32, device=dev)
    # algorithmic bytes per row of the weighted input blocks (bf16 rows count 2 bytes per value)
    in_bytes = float(sum(s.width * (2 if s.tensor.dtype == torch.bfloat16 else 4) for s in sources if not s.additive))
    if store_rows:
        args = (_lib.ptr(out), _ld(out), _lib.ptr(out_idx32), act, _lib.ptr(resid), _ld(resid) if resid is not None else 0,
                resid_col0, _lib.stream_handle(dev))
    if agg is not None:
        csr, agg_out, agg_mean = agg
        tiles = csr.tiles() if fusable else None
        if tiles is None:        # not fusable here: the plain launch, then the separate reduction
            y = mlp_forward(packed, sources, n_rows, act, out, out_idx32, resid, resid_col0, tile_mode, head_outs)
            segment_reduce(y, csr, agg_mean, out=agg_out)
            return y
        _lib.require_hip(agg_out)
        t_rows, t_seg, nt = tiles
        mode = _agg_mode(agg_mean, csr)
        if agg_out.dtype == torch.bfloat16:          # (G4C_AGG_OUT_BF16: bf16 aggregate rows in the row-split kernel's column order)
            if not (packed.rs_order and packed.precision == "bf16"):
                raise ValueError("a bf16 aggregate needs weights packed for the row-split kernel (rounded-bf16 mode)")
            mode |= 1 << 16
        elif agg_out.dtype != torch.float32:
            raise TypeError(f"aggregate: expected float32 (or bfloat16 on the row-split kernel), got {agg_out.dtype}")
        tail = (_lib.ptr(t_rows), _lib.ptr(t_seg), _lib.ptr(csr.off), nt, _lib.ptr(agg_out), _ld(agg_out), mode,
                _lib.stream_handle(dev))
        o_ld = _ld(out) if out is not None else 128
        if packed.precision == "bf16x6":
            call = lambda: _lib.check(lib.g4c_mlp_forward_bx6_agg(C.byref(packed.desc), arr, len(sources), n_rows, _lib.ptr(out), o_ld, act, *tail))
        else:
            if out is not None and out.dtype not in (torch.float32, torch.bfloat16):
                raise TypeError(f"out: expected float32 or bfloat16, got {out.dtype}")
            o_dt = 1 if (out is not None and out.dtype == torch.bfloat16) else 0
            if o_dt and rows_act == _lib.ACT_SELU and rows16:
                o_dt = 2
            elif rows_act != _lib.ACT_NONE and rows16:
                raise ValueError("rows_act: only ACT_SELU")
            call = lambda: _lib.check(lib.g4c_mlp_forward_bf16_agg(C.byref(packed.desc), arr, len(sources), n_rows, _lib.ptr(out), o_ld, o_dt,
                                                                   act, *tail))
        if KernelTimer.active is None:
            call()
        else:
            out_b = 0 if not store_rows else packed.n_out * (2 if out.dtype == torch.bfloat16 else 4)
            _timed("mlp_bx6_kernel", packed.flops_per_row * n_rows, (in_bytes + out_b) * n_rows + 4.0 * packed.n_out * csr.n_seg, call)
    elif save is not None:
        if packed.precision != "bf16x6" or head_outs is not None or out_idx32 is not None or tile_mode is not None:
            raise NotImplementedError("save= needs the bf16x6 kernel without heads / output index / forced tile mode")
        if len(save) != packed.desc.n_layers:
            raise ValueError(f"{len(save)} save tensors for {packed.desc.n_layers} layers")
        live = [t for t in save if t is not None]
        _lib.require_hip(*live)
        if any(t.dim() != 2 or t.size(0) < n_rows or t.size(1) < 128 or t.stride(1) != 1 or _ld(t) != _ld(live[0]) for t in live):
            raise ValueError("save tensors must be [n_rows, >= 128] fp32 with one common leading dimension")
        sv = (C.c_void_p * len(save))(*[None if t is None else t.data_ptr() for t in save])
        ml, ml_ld = None, 0
        if mul is not None:
            lm = [t for t in mul if t is not None]
            _lib.require_hip(*lm)
            if len(mul) != len(save) or any(t.size(0) < n_rows or t.size(1) < 128 or t.stride(1) != 1 or _ld(t) != _ld(lm[0]) for t in lm):
                raise ValueError("mul tensors must be [n_rows, >= 128] fp32 with one common leading dimension, one entry per layer")
            ml = (C.c_void_p * len(mul))(*[None if t is None else t.data_ptr() for t in mul])
            ml_ld = _ld(lm[0]) if lm else 128
        call = lambda: _lib.check(lib.g4c_mlp_forward_bx6_save(
            C.byref(packed.desc), arr, len(sources), n_rows, _lib.ptr(out), _ld(out), act, _lib.ptr(resid),
            _ld(resid) if resid is not None else 0, resid_col0, sv, _ld(live[0]) if live else 128, ml, ml_ld, _lib.stream_handle(dev)))
        if KernelTimer.active is None:
            call()
        else:
            _timed("mlp_bx6_kernel", packed.flops_per_row * n_rows, 4.0 * (sum(packed.seg_widths) + packed.n_out + 128 * len(live)) * n_rows, call)
    elif packed.precision != "fp32":
        if tile_mode is not None:
            raise NotImplementedError("bf16 MLP with a forced tile mode")
        if head_outs is not None:
            if len(head_outs) != packed.n_heads or packed.n_heads == 0:
                raise ValueError(f"{len(head_outs)} head outputs for a packing with {packed.n_heads} heads")
            if out_idx32 is not None or resid is not None:
                raise NotImplementedError("heads with an output index / residual")
            _lib.require_hip(*head_outs)
            ho = (C.c_void_p * len(head_outs))(*[h.data_ptr() for h in head_outs])
            if any(h.dtype == torch.bfloat16 for h in head_outs):
                # rounded-bf16 mode: the head rows (the next message MLP's first-layer products) stored as bf16 (g4c_mlp_forward_heads_bf16_out)
                if packed.precision != "bf16" or any(h.dtype != torch.bfloat16 for h in head_outs):
                    raise TypeError("bf16 head outputs need the rounded-bf16 mode, and every head in bf16")
                # (a bf16 `out` as well: the launch's own rows stored as bf16 — g4c_mlp_forward_heads_bf16_rows)
                call = lambda: _lib.check(lib.g4c_mlp_forward_heads_bf16_rows(C.byref(packed.desc), arr, len(sources), n_rows, _lib.ptr(out),
                                                                              _ld(out), 1 if out.dtype == torch.bfloat16 else 0, act, packed.head_w,
                                                                              len(head_outs), ho, _ld(head_outs[0]), 1, _lib.stream_handle(dev)))
            else:
                if out is not None and out.dtype != torch.float32:
                    raise TypeError("heads stored as fp32 need fp32 output rows (bf16 rows: bf16 heads, g4c_mlp_forward_heads_bf16_rows)")
                heads_fn = lib.g4c_mlp_forward_heads_bf16 if packed.precision == "bf16" else lib.g4c_mlp_forward_heads_bx6
                call = lambda: _lib.check(heads_fn(C.byref(packed.desc), arr, len(sources), n_rows, _lib.ptr(out),
                                                   _ld(out), act, packed.head_w, len(head_outs), ho,
                                                   _ld(head_outs[0]), _lib.stream_handle(dev)))
        elif out is not None and out.dtype == torch.bfloat16:
            # rounded-bf16 mode: plain 128-wide output rows stored as bf16 (g4c_mlp_forward_bf16_out: first-layer products)
            if packed.precision != "bf16" or out_idx32 is not None or resid is not None:
                raise TypeError("a bf16 `out` needs the rounded-bf16 mode and no output index / residual")
            call = lambda: _lib.check(lib.g4c_mlp_forward_bf16_out(C.byref(packed.desc), arr, len(sources), n_rows, _lib.ptr(out), _ld(out), 1, act,
                                                                   _lib.stream_handle(dev)))
        else:
            fwd = lib.g4c_mlp_forward_bf16 if packed.precision == "bf16" else lib.g4c_mlp_forward_bx6
            call = lambda: _lib.check(fwd(C.byref(packed.desc), arr, len(sources), n_rows, *args))
        if KernelTimer.active is None:
            call()
        else:
            _timed("mlp_bx6_kernel", packed.flops_per_row * n_rows, (in_bytes + 4.0 * packed.n_out) * n_rows, call)
    elif head_outs is not None:
        if len(head_outs) != packed.n_heads or packed.n_heads == 0:
            raise ValueError(f"{len(head_outs)} head outputs for a packing with {packed.n_heads} heads")
        if out_idx32 is not None or resid is not None or tile_mode is not None:
            raise NotImplementedError("heads with an output index / residual / forced tile mode")
        _lib.require_hip(*head_outs)
        ho = (C.c_void_p * len(head_outs))(*[h.data_ptr() for h in head_outs])
        call = lambda: _lib.check(lib.g4c_mlp_forward_heads(C.byref(packed.desc), arr, len(sources), n_rows, _lib.ptr(out), _ld(out),
                                                            act, packed.head_w, len(head_outs), ho, _ld(head_outs[0]),
                                                            _lib.stream_handle(dev)))
        if KernelTimer.active is None:
            call()
        else:
            _timed("mlp_split_kernel<4>", packed.flops_per_row * n_rows, 4.0 * (sum(packed.seg_widths) + packed.n_out * (1 + packed.n_heads)) * n_rows, call)
    elif tile_mode is not None:
        _lib.check(lib.g4c_mlp_forward_rows(C.byref(packed.desc), arr, len(sources), n_rows, 0, n_rows, tile_mode, *args))
    elif KernelTimer.active is None:
        _lib.check(lib.g4c_mlp_forward(C.byref(packed.desc), arr, len(sources), n_rows, *args))
    else:
        _timed("mlp_split_kernel<4>", packed.flops_per_row * n_rows, 4.0 * (sum(packed.seg_widths) + packed.n_out) * n_rows,
               lambda: _lib.check(lib.g4c_mlp_forward(C.byref(packed.desc), arr, len(sources), n_rows, *args)))
    return out

"""`gfd.nn.blocks` on MI355X: same classes, constructor arguments, `forward` signatures, submodule
names and `state_dict` keys as the reference's graphs4cfd/nn/blocks.py, with every forward routed
to the hand-written HIP kernels of libg4c.so (fused gather+MLP+LayerNorm+activation with fp32-accurate products on the
bf16 matrix pipe, CSR segmented reductions, static-plan pooling).  There is no torch / CPU fallback: tensors must live
on a HIP device and the library must be built.

With gradients enabled every block is recorded for autograd (autograd.py: HIP forward and backward); under
torch.no_grad() the inference-only forms of the launches (heads, pre-multiplied products, fused aggregation) are used.

Extensions over the reference signatures are keyword-only and optional (`activation=` on the MP
blocks fuses the `F.selu` the model applies right after the block, nn/mus_gnn.py:182).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import nn

from .. import _lib, ops, plan
from ..graph import Graph
from ..ops import Source

import os
# first-layer hoisting (MLP.run_hoisted) pays off only when the launch is throughput-bound
HOIST_MIN_ROWS = 24576

Tensor = torch.Tensor


def _narrow_flags(sources: Sequence[Source]) -> List[bool]:
    """Input blocks the bf16x6 kernels multiply on the vector ALUs instead of padding them to a 128-k matrix block:
    at most 8 columns, read row-for-row (no gather index) and without an activation on load."""
    return [s.width <= _lib.NARROW_MAX and s.index is None and s.pre_act == _lib.ACT_NONE and not s.additive for s in sources]


def _split_wide(sources: Sequence[Source]) -> List[Source]:
    """Input blocks wider than 128 columns as consecutive column chunks of at most 128 of the same tensor (same rows, same index):
    the split-operand kernels take 128-wide blocks, and `cat` of the chunks is the block.  Unchanged when nothing is wider, when the
    chunks would exceed the kernels' block count (the caller then hoists or the MLP runs on the fp32-MFMA kernels), and while a
    call is recorded for autograd."""
    if ops.grad_mode() or not any(s.width > 128 and not s.additive and s.segments is None for s in sources):
        return list(sources)
    out: List[Source] = []
    for s in sources:
        if s.width > 128 and not s.additive and s.segments is None:
            for c in range(0, s.width, 128):
                out.append(Source(s.tensor, s.index, s.col0 + c, min(128, s.width - c), s.negate, s.pre_act))
        else:
            out.append(s)
    return out if len(out) <= _lib.MAX_SRC else list(sources)


def _ld_of(t: Tensor) -> int:
    return int(t.stride(0)) if t.dim() == 2 else int(t.numel())


def _finish(x: Tensor, activation, code: Optional[int]) -> Tensor:
    """Apply an activation that could not be fused into the kernel epilogue."""
    return x if (activation is None or code is not None) else activation(x)


# ------------------------------------------------------------------------------------- MLP
class MLP(nn.Module):
    r"""Multi-layer perceptron with SELU activations (reference: nn/blocks.py:117-144).

    Args:
        input_size (int): The size of the input.
        layers_width (Tuple[int]): The width of each layer, excluding the input layer.
        layer_norm (bool, optional): LayerNorm after the last layer. Defaults to False.

    Parameters live in `self.MLP` (`linear_<i>`, `selu_<i>`, `layer_norm`) so checkpoints of the
    reference load unchanged.
    """

    def __init__(self, input_size: int, layers_width: Tuple[int], layer_norm: bool = False):
        super().__init__()
        sizes = [int(input_size)] + [int(w) for w in layers_width]
        n = len(layers_width)
        if n < 2:
            raise ValueError("layers_width needs at least two entries (nn/blocks.py:135-140)")
        mods = OrderedDict()
        for i in range(1, n + 1):
            mods[f"linear_{i}"] = nn.Linear(sizes[i - 1], sizes[i])
            if i < n:
                mods[f"selu_{i}"] = nn.SELU()
        if layer_norm:
            mods["layer_norm"] = nn.LayerNorm(sizes[-1])
        self.MLP = nn.Sequential(mods)
        self.input_size, self.output_size = sizes[0], sizes[-1]
        self._packed = {}
        self._site = None         # name in reports of clipped fp16 values (ops.f16_range_report); models set "<Model>.<attribute path>"

    # -- kernel-side weight cache ---------------------------------------------------------
    def _linears(self) -> List[nn.Linear]:
        return [m for m in self.MLP if isinstance(m, nn.Linear)]

    def _signature(self):
        return (ops.weights_epoch(),) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed(self, seg_widths: Sequence[int], seg_negate: Sequence[bool], narrow: Optional[Sequence[bool]] = None,
               rs_blocks: Optional[Sequence[bool]] = None) -> ops.PackedMLP:
        prec = ops.effective_precision(seg_widths)
        narrow = tuple(bool(x) for x in narrow) if (narrow is not None and prec != "fp32") else (False,) * len(seg_widths)
        rs_blocks = tuple(bool(x) for x in rs_blocks) if (rs_blocks is not None and prec == "bf16") else (False,) * len(seg_widths)
        key = (tuple(seg_widths), tuple(bool(x) for x in seg_negate), prec, narrow, rs_blocks)
        sig = self._signature()
        hit = self._packed.get(key)
        if hit is None or hit[0] != sig:
            if sum(seg_widths) != self.input_size:
                raise ValueError(f"MLP expects {self.input_size} input columns, got blocks {list(seg_widths)}")
            lin = self._linears()
            ln = getattr(self.MLP, "layer_norm", None)
            pk = ops.PackedMLP([l.weight for l in lin], [l.bias for l in lin],
                               None if ln is None else (ln.weight, ln.bias, ln.eps), key[0], key[1], precision=prec, narrow=narrow,
                               site=self._site, rs_blocks=rs_blocks if any(rs_blocks) else None)
            self._packed[key] = (sig, pk)
            hit = self._packed[key]
        return hit[1]

    # -- MLPs outside the one-launch envelope ---------------------------------------------
    def fits_one_launch(self) -> bool:
        """The fused kernels take up to _lib.MAX_LAYERS Linear layers of at most 128 outputs each (every published graphs4cfd
        architecture).  The reference's MLP (nn/blocks.py:129-141) accepts any widths and depth: such an MLP runs as a chain of
        launches (`_run_stages`)."""
        lin = self._linears()
        return len(lin) <= _lib.MAX_LAYERS and all(l.out_features <= 128 for l in lin)

    def _stage(self, key, weights, biases, ln, sources: Sequence[Source]) -> ops.PackedMLP:
        prec = ops.effective_precision([s.width for s in sources])
        narrow = tuple(_narrow_flags(sources)) if prec != "fp32" else (False,) * len(sources)
        key = ("stage",) + key + (tuple(s.width for s in sources), tuple(s.negate for s in sources), prec, narrow)
        sig = self._signature()
        hit = self._packed.get(key)
        if hit is None or hit[0] != sig:
            pk = ops.PackedMLP(weights, biases, ln, key[-4], key[-3], precision=prec, narrow=narrow, site=self._site)
            self._packed[key] = (sig, pk)
            hit = self._packed[key]
        return hit[1]

    def _run_stages(self, sources: Sequence[Source], n_rows: int, act_code: int, **kw) -> Tensor:
        """Any widths / depth as a chain of fused launches: consecutive layers of <= 128 outputs share a launch (at most
        _lib.MAX_LAYERS of them, SELU between them as always); a layer with more than 128 outputs is a launch per 128-column
        chunk of its output (rows [c, c + 128) of its weight), whose results enter the next layer as its input blocks; a layer
        with more than _lib.MAX_SRC input blocks is a chain of launches over groups of blocks, each adding its products to the
        partial sums of the one before (an additive source).  LayerNorm and the caller's activation / residual / output index
        belong to the last launch; a LayerNorm over more than 128 columns is its own launch (ops.layer_norm) behind the chunks.
        Recorded for autograd (gradients enabled): every launch of the chain is a differentiable fused launch of its own; what joins
        them — the concatenation of a wide last layer's chunks, the sum over groups of input blocks, a LayerNorm / activation behind
        either — is then plain torch on the device (these shapes occur in no published architecture)."""
        grad = ops.grad_mode()
        if grad and any(kw.get(k) is not None for k in ("out", "out_idx32", "agg", "head_outs")):
            raise NotImplementedError("out= / output index / aggregation / heads are inference-only forms; call under torch.no_grad()")
        lin = self._linears()
        ln = getattr(self.MLP, "layer_norm", None)
        dev = sources[0].tensor.device
        cur, i, n = list(_split_wide(sources)), 0, len(lin)

        def torch_tail(y: Tensor, ln_args, act: int) -> Tensor:
            if ln_args is not None:
                y = torch.nn.functional.layer_norm(y, (y.size(1),), ln_args[0], ln_args[1], ln_args[2])
            return torch.nn.functional.selu(y) if act == _lib.ACT_SELU else (torch.tanh(y) if act == _lib.ACT_TANH else y)

        def one_layer(li: int, c: int, c1: int, blocks: Sequence[Source], act: int, ln_args, **kw2) -> Tensor:
            """act(LayerNorm(W[c:c1] cat(blocks) + b[c:c1])) — in groups of blocks when there are more than a launch takes"""
            w, b = lin[li].weight[c:c1], (lin[li].bias[c:c1] if lin[li].bias is not None else None)
            if len(blocks) <= _lib.MAX_SRC:
                return ops.mlp_forward(self._stage((li, c), [w], [b], ln_args, blocks), blocks, n_rows, act, **kw2)
            gsz = _lib.MAX_SRC if grad else _lib.MAX_SRC - 1      # (inference: the previous group's partial sums take a source slot)
            groups = [blocks[g:g + gsz] for g in range(0, len(blocks), gsz)]
            partial, k0 = None, 0
            for gi, grp in enumerate(groups):
                k1 = k0 + sum(s_.width for s_ in grp)
                last_g = gi == len(groups) - 1
                fused_tail = last_g and not grad
                pk = self._stage((li, c, k0, grad), [w[:, k0:k1]], [b if gi == 0 else None], ln_args if fused_tail else None, grp)
                srcs = list(grp) + ([Source(partial, additive=True)] if (partial is not None and not grad) else [])
                y = ops.mlp_forward(pk, srcs, n_rows, act if fused_tail else _lib.ACT_NONE, **(kw2 if fused_tail else {}))
                partial = y if (partial is None or not grad) else partial + y
                k0 = k1
            if grad:
                if kw2.get("resid") is not None:
                    raise NotImplementedError("a residual behind a layer of more than 4 input blocks, recorded for autograd")
                partial = torch_tail(partial, ln_args, act)
            return partial

        while i < n:
            if lin[i].out_features > 128:
                w_out = lin[i].out_features
                last = i == n - 1
                if last and any(kw.get(k) is not None for k in ("out_idx32", "resid", "agg", "head_outs")):
                    raise NotImplementedError("output index / residual / aggregation / heads on an output wider than 128 columns")
                wide_ln = last and ln is not None
                act_i = _lib.ACT_NONE if wide_ln else (act_code if last else _lib.ACT_SELU)
                spans = [(c, min(c + 128, w_out)) for c in range(0, w_out, 128)]
                if grad:
                    parts = [one_layer(i, c, c1, cur, act_i, None) for c, c1 in spans]
                    if last:
                        y = torch.cat(parts, 1)
                        return torch_tail(y, (ln.weight, ln.bias, ln.eps), act_code) if wide_ln else y
                    cur = [Source(t) for t in parts]
                    i += 1
                    continue
                wide = kw.get("out") if (last and kw.get("out") is not None) else torch.empty((n_rows, w_out), dtype=torch.float32, device=dev)
                for c, c1 in spans:
                    one_layer(i, c, c1, cur, act_i, None, out=wide[:, c:c1])
                if wide_ln:         # (a LayerNorm over more than 128 columns: its own launch, in place)
                    ops.layer_norm(wide, ln.weight, ln.bias, ln.eps, act_code, out=wide)
                if last:
                    return wide
                cur = [Source(wide, col0=c, width=c1 - c) for c, c1 in spans]
                i += 1
                continue
            if len(cur) > _lib.MAX_SRC:          # (too many input blocks for one launch: this layer alone, group by group)
                last = i == n - 1
                ln_args = (ln.weight, ln.bias, ln.eps) if (last and ln is not None) else None
                y = one_layer(i, 0, lin[i].out_features, cur, act_code if last else _lib.ACT_SELU, ln_args, **(kw if last else {}))
                if last:
                    return y
                cur, i = [Source(y)], i + 1
                continue
            j = i
            while j < n and j - i < _lib.MAX_LAYERS and lin[j].out_features <= 128:
                j += 1
            last = j == n
            pk = self._stage((i, j), [l.weight for l in lin[i:j]], [l.bias for l in lin[i:j]],
                             (ln.weight, ln.bias, ln.eps) if (last and ln is not None) else None, cur)
            if last:
                return ops.mlp_forward(pk, cur, n_rows, act_code, **kw)
            cur = [Source(ops.mlp_forward(pk, cur, n_rows, _lib.ACT_SELU))]
            i = j
        raise AssertionError("unreachable")

    def run(self, sources: Sequence[Source], n_rows: int, activation=None, out: Optional[Tensor] = None,
            out_idx32: Optional[Tensor] = None, resid: Optional[Tensor] = None, resid_col0: int = 0) -> Tensor:
        """cat(sources) -> MLP -> activation (+ resid), one fused launch."""
        code = _lib.act_code(activation)
        if code is None and resid is not None:
            raise NotImplementedError("a residual after a non-fusable activation")
        if not self.fits_one_launch():
            y = self._run_stages(sources, n_rows, _lib.ACT_NONE if code is None else code, out=out, out_idx32=out_idx32, resid=resid,
                                 resid_col0=resid_col0)
            return _finish(y, activation, code)
        sources = _split_wide(sources)
        pk = self.packed([s.width for s in sources], [s.negate for s in sources], _narrow_flags(sources), _rs_blocks(sources))
        y = ops.mlp_forward(pk, sources, n_rows, _lib.ACT_NONE if code is None else code, out, out_idx32, resid, resid_col0)
        return _finish(y, activation, code)

    def run_coded(self, sources: Sequence[Source], n_rows: int, act_code: int = _lib.ACT_NONE, **kw) -> Tensor:
        if not self.fits_one_launch():
            return self._run_stages(sources, n_rows, act_code, **kw)
        fmt = self._rs2_format(sources, n_rows, act_code, kw)
        if fmt:
            y = ops.mlp_forward(self._rs2_packed(fmt, None, 0, ()), sources, n_rows, act_code, **kw)
            return ops.RsOrderedRows.tag(y) if y.dtype == torch.bfloat16 else y
        sources = _split_wide(sources)
        pk = self.packed([s.width for s in sources], [s.negate for s in sources], _narrow_flags(sources), _rs_blocks(sources))
        return ops.mlp_forward(pk, sources, n_rows, act_code, **kw)

    def _heads_packed(self, seg_widths: Sequence[int], consumer: "MLP", k_cols: int, widths: Sequence[int]) -> Optional[ops.PackedMLP]:
        """This MLP (plain 128-wide input blocks `seg_widths`) packed with heads = the column blocks `widths` behind the first `k_cols`
        columns of `consumer`'s first layer (see run_with_heads; same cache entry)."""
        if self.output_size != 128 or any(int(w) != 128 for w in widths) or not 1 <= len(widths) <= _lib.MAX_HEADS or not self.fits_one_launch():
            return None
        prec = ops.effective_precision(seg_widths)
        narrow = (False,) * len(seg_widths)
        key = ("heads", id(consumer), k_cols, tuple(widths), tuple(seg_widths), (False,) * len(seg_widths), prec, narrow, False,
               (False,) * len(seg_widths))
        sig = (self._signature(), consumer._signature())
        hit = self._packed.get(key)
        if hit is None or hit[0] != sig:
            lin = self._linears()
            ln = getattr(self.MLP, "layer_norm", None)
            w1 = consumer._linears()[0].weight.detach()
            if int(w1.size(0)) != 128:
                return None
            heads, off = [], k_cols
            for w in widths:
                heads.append(w1[:, off:off + w].contiguous())
                off += w
            pk = ops.PackedMLP([l.weight for l in lin], [l.bias for l in lin],
                               None if ln is None else (ln.weight, ln.bias, ln.eps), key[4], key[5], heads=heads, precision=prec,
                               narrow=narrow, site=self._site)
            self._packed[key] = (sig, pk)
            hit = self._packed[key]
        return hit[1]

    def run_with_heads(self, sources: Sequence[Source], n_rows: int, act_code: int, consumer: "MLP", k_cols: int,
                       widths: Sequence[int], out: Optional[Tensor] = None,
                       head_outs: Optional[Sequence[Optional[Tensor]]] = None, rs_rows: bool = False) -> Optional[Tuple[Tensor, List[Tensor]]]:
        """This MLP on `sources`, plus — from the same launch — the first-layer products `consumer` will need from
        this MLP's output y: [W1c[:, a:b] y for consecutive column blocks [a, b) of `widths` after the first `k_cols`
        columns of consumer's first layer] (see MLP.run_hoisted; g4c_mlp_forward_heads).
        Returns None when the launch cannot carry heads (shape envelope / kernel variant): the caller then lets the
        consumer compute its products itself.  `out` / `head_outs` (entries may be None): caller-provided [n_rows, 128]
        destinations (row-sliced views of wider buffers are fine).  `rs_rows` (rounded-bf16 mode, bf16 products: the consumer's message
        launch will run on the row-split kernel, MLP.rs1_ready): the products come back as ops.RsOrderedRows, their columns in that
        kernel's order — the heads' weight rows are permuted, nothing else changes."""
        if self.output_size != 128 or any(int(w) != 128 for w in widths) or not 1 <= len(widths) <= _lib.MAX_HEADS:
            return None
        if not self.fits_one_launch():
            return None
        if ops.grad_mode():          # recorded for autograd: the plain launches are the differentiable ones
            return None
        if not any(t is not None for t in (head_outs or ())) and PRODUCTS_BF16 and HOIST_BF16:
            fmt = self._rs2_format(sources, n_rows, act_code, {} if out is None else {"out": out}, rs_rows)
            pk2 = self._rs2_packed(fmt, consumer, k_cols, widths) if fmt else None
            if pk2 is not None:          # the row-split update kernel: e' (+ the consumer's products) from rows in its own order
                dev = sources[0].tensor.device
                y = out if out is not None else torch.empty((n_rows, 128), dtype=torch.float32, device=dev)
                outs = [torch.empty((n_rows, 128), dtype=torch.bfloat16, device=dev) for _ in range(2)]
                ops.mlp_forward(pk2, sources, n_rows, act_code, out=y, head_outs=outs)
                return (ops.RsOrderedRows.tag(y) if y.dtype == torch.bfloat16 else y), [ops.RsOrderedRows.tag(t) for t in outs]
        sources = _split_wide(sources)
        prec = ops.effective_precision([s.width for s in sources])
        if prec == "bf16" and not HOIST_BF16:
            return None
        narrow = tuple(_narrow_flags(sources)) if prec != "fp32" else (False,) * len(sources)
        rs_rows = bool(rs_rows and prec == "bf16" and PRODUCTS_BF16 and not any(t is not None for t in (head_outs or ())))
        rs_blocks = _rs_blocks(sources) if prec == "bf16" else (False,) * len(sources)
        key = ("heads", id(consumer), k_cols, tuple(widths), tuple(s.width for s in sources), tuple(s.negate for s in sources), prec, narrow, rs_rows,
               rs_blocks)
        sig = (self._signature(), consumer._signature())
        hit = self._packed.get(key)
        if hit is None or hit[0] != sig:
            lin = self._linears()
            ln = getattr(self.MLP, "layer_norm", None)
            w1 = consumer._linears()[0].weight.detach()
            if rs_rows and int(w1.size(0)) == 128:
                w1 = w1[ops._rs_k_order(w1.device)]
            heads, off = [], k_cols
            for w in widths:
                heads.append(w1[:, off:off + w].contiguous())
                off += w
            if int(w1.size(0)) != 128:
                return None
            pk = ops.PackedMLP([l.weight for l in lin], [l.bias for l in lin],
                               None if ln is None else (ln.weight, ln.bias, ln.eps), key[4], key[5], heads=heads, precision=prec,
                               narrow=narrow, site=self._site, rs_blocks=rs_blocks if any(rs_blocks) else None)
            self._packed[key] = (sig, pk)
            hit = self._packed[key]
        pk = hit[1]
        dev = sources[0].tensor.device
        y = out if out is not None else torch.empty((n_rows, 128), dtype=torch.float32, device=dev)
        given = [t for t in (head_outs or ()) if t is not None]
        hdt = given[0].dtype if given else (torch.bfloat16 if (prec == "bf16" and PRODUCTS_BF16) else torch.float32)
        outs = [(head_outs[j] if head_outs is not None and head_outs[j] is not None else
                 torch.empty((n_rows, 128), dtype=hdt, device=dev)) for j in range(len(widths))]
        if any(_ld_of(t) != _ld_of(outs[0]) or t.dtype != hdt for t in outs):        # one leading dimension / type for all heads (g4c_mlp_forward_heads)
            return None
        ops.mlp_forward(pk, sources, n_rows, act_code, out=y, head_outs=outs)
        return y, ([ops.RsOrderedRows.tag(t) for t in outs] if rs_rows else outs)

    # -- first-layer hoisting ------------------------------------------------------------------
    def _packed_cols(self, tag: str, a: int, b: int, seg_widths, seg_negate, first_only: bool, rs_order: bool = False,
                     rs_rows: bool = False, rs_in: bool = False) -> ops.PackedMLP:
        """Packed variant using only columns [a, b) of the first Linear layer (`first_only`: that layer alone, no bias).
        `rs_order`: the stream of the row-split kernel (ops.PackedMLP); `rs_rows` (with `first_only`): the product's output columns in that
        kernel's order (ops.RsOrderedRows) — the weight's rows permuted."""
        prec = ops.effective_precision(seg_widths)
        key = (tag, a, b, tuple(seg_widths), tuple(bool(x) for x in seg_negate), prec)
        sig = self._signature()
        hit = self._packed.get(key)
        if hit is None or hit[0] != sig:
            lin = self._linears()
            w1 = lin[0].weight.detach()[:, a:b].contiguous()
            if rs_rows:
                w1 = w1[ops._rs_k_order(w1.device)].contiguous()
            if first_only:
                # (`rs_in`: the one input block arrives as ops.RsOrderedRows — its weight columns are packed in that order)
                pk = ops.PackedMLP([w1], [None], None, key[3], key[4], precision=prec, site=self._site,
                                   rs_blocks=[True] if (rs_in and prec == "bf16") else None)
            else:
                ln = getattr(self.MLP, "layer_norm", None)
                pk = ops.PackedMLP([w1] + [l.weight for l in lin[1:]], [l.bias for l in lin],
                                   None if ln is None else (ln.weight, ln.bias, ln.eps), key[3], key[4], precision=prec, site=self._site,
                                   rs_order=rs_order)
            self._packed[key] = (sig, pk)
            hit = self._packed[key]
        return hit[1]

    def run_hoisted(self, k_sources: Sequence[Source], gathered: Sequence[Tuple[Tensor, Tensor]], n_rows: int,
                    act_code: int = _lib.ACT_NONE, products: Optional[Sequence[Tensor]] = None, **kw) -> Tensor:
        """MLP(cat(k_sources..., t0[idx0], t1[idx1], ...)) with the first layer's products of the gathered node-side
        inputs hoisted: W1 [x | t[idx]] = W1x x + (W1t t)[idx] (exact up to fp32 re-association), so `W1t t` costs
        rows(t) instead of n_rows.  `gathered` = [(tensor [n_t, w_t], int32 index [n_rows])] in concat order.
        Below HOIST_MIN_ROWS the launch is latency-bound and the extra product launches cost more than the MFMA
        work they save (measured crossover ~25k rows, scripts/sweep_tile_modes.py): one plain fused launch then.
        `products` (from the producer's launch, MLP.run_with_heads): the per-node terms, already multiplied."""
        if self._rs1_takes(k_sources, gathered, n_rows, act_code, products, kw):
            return self._run_rs1(k_sources[0], gathered, n_rows, products, kw)
        plain = list(k_sources) + [Source(t, index=idx) for t, idx in gathered]
        if not self.fits_one_launch():          # (a chain of launches: nothing to hoist into)
            return self.run_coded(plain, n_rows, act_code, **kw)
        # (a gathered block wider than 128 whose chunks would exceed the kernels' block count — gMuS-GNN's 2H-wide latents after an
        # up-sampling — is hoisted at any size: its products are formed from 128-wide chunks, the launch itself keeps one block)
        too_many = (sum((s.width + 127) // 128 for s in plain) > _lib.MAX_SRC and any(s.width > 128 for s in plain)
                    and ops.mlp_precision() in ("bf16x6", "f16x3") and not ops.grad_mode())
        if (n_rows < HOIST_MIN_ROWS or (ops.mlp_precision() == "bf16" and not HOIST_BF16) or ops.grad_mode()) and products is None and not too_many:
            return self.run_coded(plain, n_rows, act_code, **kw)
        kw_widths = [s.width for s in k_sources]
        off = sum(kw_widths)
        adds = []
        for j, (t, idx) in enumerate(gathered):
            w_t = int(t.size(1))
            if products is not None:
                part = products[j]
            else:
                chunks = [(c, min(128, w_t - c)) for c in range(0, w_t, 128)] if (w_t > 128 and not ops.grad_mode()) else [(0, w_t)]
                pk1 = self._packed_cols("hoist1", off, off + w_t, [w for _, w in chunks], [False] * len(chunks), True)
                part16 = (torch.empty((int(t.size(0)), 128), dtype=torch.bfloat16, device=t.device)
                          if (ops.mlp_precision() == "bf16" and PRODUCTS_BF16 and pk1.n_out == 128 and pk1.precision == "bf16") else None)
                part = ops.mlp_forward(pk1, [Source(t, col0=c, width=w) for c, w in chunks], int(t.size(0)), out=part16)
            adds.append(Source(part, index=idx, additive=True))
            off += w_t
        if off != self.input_size:
            raise ValueError(f"MLP expects {self.input_size} input columns, got {off}")
        pk = self._packed_cols("hoist", 0, sum(kw_widths), kw_widths, [s.negate for s in k_sources], False)
        return ops.mlp_forward(pk, list(k_sources) + adds, n_rows, act_code, **kw)

    # -- rounded-bf16 mode: the update MLP of such a layer on the row-split update kernel -------------------------------------
    def _rs2_format(self, sources: Sequence[Source], n_rows: int, act_code: int, kw: dict, rs_rows: bool = True) -> int:
        """0, or the stream format (ops.PackedMLP rs2: 4 / 5) of mlp_rs2_kernel when this launch fits it: [aggregate | e] as two whole
        128-wide bf16 blocks — the aggregate in the row-split order (blocks.AGGREGATE_BF16), e in that order (4) or in feature order (5) —
        two layers 256 -> 128 -> 128, LayerNorm, activation none / SELU, plain fp32 or bf16 output rows."""
        if not (ROW_SPLIT_BF16 and UPDATE_ROW_SPLIT) or ops.mlp_precision() != "bf16" or ops.grad_mode() or n_rows < RS1_MIN_ROWS:
            return 0
        if len(sources) != 2 or act_code not in (_lib.ACT_NONE, _lib.ACT_SELU) or not rs_rows or not set(kw) <= {"out"}:
            return 0
        for s_ in sources:
            if (s_.width != 128 or s_.col0 != 0 or s_.negate or s_.index is not None or s_.segments is not None or s_.additive
                    or s_.pre_act != _lib.ACT_NONE or s_.tensor.dtype != torch.bfloat16 or s_.tensor.stride(0) % 8 or s_.tensor.data_ptr() % 16):
                return 0
        if not isinstance(sources[0].tensor, ops.RsOrderedRows):
            return 0
        lin = self._linears()
        if (len(lin) != 2 or tuple(lin[0].weight.shape) != (128, 256) or tuple(lin[1].weight.shape) != (128, 128)
                or getattr(self.MLP, "layer_norm", None) is None):
            return 0
        out = kw.get("out")
        if out is not None and (out.dtype not in (torch.float32, torch.bfloat16) or out.stride(1) != 1 or out.data_ptr() % 16
                                or out.stride(0) % (8 if out.dtype == torch.bfloat16 else 4)):
            return 0
        return 4 if isinstance(sources[1].tensor, ops.RsOrderedRows) else 5

    def _rs2_packed(self, fmt: int, consumer: Optional["MLP"], k_cols: int, widths) -> Optional[ops.PackedMLP]:
        key = ("rs2", fmt, None if consumer is None else id(consumer), k_cols, tuple(widths))
        sig = (self._signature(), None if consumer is None else consumer._signature())
        hit = self._packed.get(key)
        if hit is None or hit[0] != sig:
            lin = self._linears()
            ln = self.MLP.layer_norm
            heads = []
            if consumer is not None:
                w1 = consumer._linears()[0].weight.detach()
                if int(w1.size(0)) != 128 or tuple(int(w) for w in widths) != (128, 128):
                    return None
                heads = [w1[:, k_cols + 128 * j: k_cols + 128 * (j + 1)].contiguous() for j in range(2)]      # (rows in feature order: the kernel's stores order them)
            pk = ops.PackedMLP([l.weight for l in lin], [l.bias for l in lin], (ln.weight, ln.bias, ln.eps), [128, 128], [False, False],
                               heads=heads, precision="bf16", site=self._site, rs2=fmt)
            self._packed[key] = (sig, pk)
            hit = self._packed[key]
        return hit[1]

    # -- rounded-bf16 mode: uniform-degree message launches on the row-split kernel ------------------------------------------
    def rs1_ready(self, n_rows: int, csr) -> bool:
        """Whether a message launch of this MLP over `csr` (its rows grouped by receiver) is one the row-split kernel takes in the
        rounded-bf16 mode (mlp_rs.hip: mlp_rs1_kernel): [e | s[row] | r[col]] with three 128-wide blocks, two or three 128 x 128 layers
        behind the first, LayerNorm, receivers of one uniform in-degree 4 .. 8, the aggregation fused.  The launch that produces
        this MLP's hoisted products asks the same question (MLP.run_with_heads `rs_rows`): products and compact message rows of such
        launches are bf16 rows in that kernel's column order (ops.RsOrderedRows)."""
        if not (ROW_SPLIT_BF16 and HOIST_BF16 and PRODUCTS_BF16 and ops.FUSE_AGG) or ops.mlp_precision() != "bf16" or ops.grad_mode():
            return False
        lin = self._linears()
        if (len(lin) not in (2, 3) or tuple(lin[0].weight.shape) != (128, 384) or any(tuple(l.weight.shape) != (128, 128) for l in lin[1:])
                or getattr(self.MLP, "layer_norm", None) is None or not self.fits_one_launch()):
            return False
        return n_rows >= RS1_MIN_ROWS and n_rows == csr.n and 4 <= csr.uniform_deg <= 8 and csr.tiles() is not None

    def _rs1_takes(self, k_sources, gathered, n_rows, act_code, products, kw) -> bool:
        agg = kw.get("agg")
        if agg is None or act_code != _lib.ACT_NONE or not set(kw) <= {"agg", "store_rows", "rows_dtype", "rows_act"}:
            return False
        if not self.rs1_ready(n_rows, agg[0]) or len(k_sources) != 1 or len(gathered) != 2:
            return False
        x = k_sources[0]
        if (x.width != 128 or x.col0 != 0 or x.negate or x.index is not None or x.segments is not None or x.additive
                or x.pre_act not in (_lib.ACT_NONE, _lib.ACT_SELU)):
            return False
        if x.tensor.dtype == torch.bfloat16 and not (isinstance(x.tensor, ops.RsOrderedRows) and x.pre_act == _lib.ACT_NONE):
            return False
        if any(int(t.size(1)) != 128 for t, _ in gathered):
            return False
        if products is None and any(t.dtype not in (torch.float32, torch.bfloat16) for t, _ in gathered):
            return False          # (the product launches read them; tagged rows are restored to feature order there)
        if products is not None and not (len(products) == 2 and all(isinstance(t, ops.RsOrderedRows) for t in products)):
            return False
        if kw.get("rows_dtype") == torch.bfloat16 and kw.get("store_rows", True) and kw.get("rows_act", _lib.ACT_NONE) != _lib.ACT_SELU:
            return False
        return (agg[1].dtype in (torch.float32, torch.bfloat16) and agg[1].stride(1) == 1 and agg[1].data_ptr() % 16 == 0
                and agg[1].stride(0) % (8 if agg[1].dtype == torch.bfloat16 else 4) == 0)

    def _run_rs1(self, x: Source, gathered, n_rows: int, products, kw) -> Optional[Tensor]:
        adds = []
        for j, (t, idx) in enumerate(gathered):
            if products is not None:
                part = products[j]
            else:        # this block's product W1[:, 128 (j + 1) : 128 (j + 2)] t, bf16 rows in the kernel's column order
                tagged = isinstance(t, ops.RsOrderedRows)
                pk1 = self._packed_cols("hoist1_rs" + ("/rs_in" if tagged else ""), 128 * (j + 1), 128 * (j + 2), [128], [False], True,
                                        rs_rows=True, rs_in=tagged)
                part = torch.empty((int(t.size(0)), 128), dtype=torch.bfloat16, device=t.device)
                ops.mlp_forward(pk1, [Source(t)], int(t.size(0)), out=part)
                part = ops.RsOrderedRows.tag(part)
            adds.append(Source(part, index=idx, additive=True))
        pk = self._packed_cols("hoist_rs", 0, 128, [128], [False], False, rs_order=True)
        y = ops.mlp_forward(pk, [x] + adds, n_rows, _lib.ACT_NONE, **kw)
        return ops.RsOrderedRows.tag(y) if (y is not None and y.dtype == torch.bfloat16) else y

    def forward(self, x: Tensor) -> Tensor:
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.size(-1))
        y = self.run([Source(x2)], int(x2.size(0)))
        return y.reshape(*lead, y.size(-1))


# ------------------------------------------------------------------------------------- helpers
def _rs_blocks(sources: Sequence[Source]) -> Tuple[bool, ...]:
    """Per weighted input block: its rows are ops.RsOrderedRows (128 wide, whole) — the pack then takes that block's weight columns in the
    same order (ops.PackedMLP rs_blocks) instead of the rows being copied back to feature order."""
    return tuple(isinstance(s.tensor, ops.RsOrderedRows) and s.width == 128 and s.col0 == 0 and s.segments is None
                 for s in sources if not s.additive)


def scatter(src: Tensor, index: Tensor, dim: int = 0, dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    """Drop-in for `torch_geometric.utils.scatter(src, index, dim=0, dim_size, reduce)` as the
    reference uses it (nn/blocks.py:46-47,183,231,330,378): a CSR segmented reduction on a cached plan."""
    if dim != 0:
        raise NotImplementedError("scatter along dim != 0")
    if reduce not in ("sum", "add", "mean"):
        raise ValueError(f"unsupported reduce {reduce!r}")
    csr = plan.segments_of_sorted(index, dim_size)
    squeeze = src.dim() == 1
    out = ops.segment_reduce(src.reshape(src.size(0), -1), csr, reduce == "mean")
    return out.reshape(-1) if squeeze else out.reshape((csr.n_seg,) + tuple(src.shape[1:]))


def restriction(graph: Graph, coarse_mask: Tensor, edge_attr: Tensor, edge_index: Tensor, num_nodes: int,
                device: torch.device) -> None:
    r"""Restricts a graph to a subset of nodes (gMuS-GNN models; reference: nn/blocks.py:9-32).
    Pure index bookkeeping, static per mesh. The graph is modified in-place."""
    lut = torch.full((num_nodes,), -1, dtype=torch.long, device=device)
    lut[coarse_mask] = torch.arange(graph.field.size(0), dtype=torch.long, device=device)
    graph.edge_index = lut[edge_index]
    graph.edge_attr = edge_attr


def knn_interpolate(x: Tensor, y_idx: Tensor, x_idx: Tensor, weights: Tensor,
                    *, out: Optional[Tensor] = None, out_idx32: Optional[Tensor] = None) -> Tensor:
    r"""Inverse-distance interpolation with precomputed indices and weights (reference: nn/blocks.py:34-48).
    `y_idx` must be sorted (the layout `get_knn_interpolate_weights` produces)."""
    csr = plan.segments_of_sorted(y_idx)
    return ops.weighted_segment_mean(x, plan.index32(x_idx), weights, csr, out, out_idx32)


def pool_edge(idxHR_to_idxLR: Tensor, edge_index: Tensor, edge_attr: Tensor, aggr: str = "mean") -> Tuple[Tensor, Tensor]:
    r"""Pools the edges of a graph through the idxHR_to_idxLR mapping (reference: nn/blocks.py:51-68).
    The topology (remap, self-loop removal, coalescing order) comes from the static plan; only the
    feature reduction runs per call."""
    if aggr not in ("mean", "sum", "add"):
        raise ValueError(f"unsupported aggr {aggr!r}")
    pp = plan.pool_edge_plan(idxHR_to_idxLR, edge_index)
    return pp.edge_index, ops.segment_reduce(edge_attr, pp.csr, aggr == "mean")


def lstsq(A: Tensor, B: Tensor) -> Tensor:
    """Solves the least squares problem AX=B for X (reference: nn/blocks.py:71-85)."""
    return torch.linalg.pinv(A) @ B


def _num_nodes_of(edge_index: Tensor) -> int:
    return plan.segments_of_sorted(edge_index[1]).n_seg


def edgeScalarToNodeVector(edge_attr: Tensor, edge_index: Tensor, edgeUnitVector: Optional[Tensor] = None,
                           edgeUnitVectorInverse: Optional[Tensor] = None, coarse_mask: Optional[Tensor] = None) -> Tensor:
    r"""For each node j the least-squares vector from its k incoming edge scalars (reference:
    nn/blocks.py:88-114). Edges must be grouped by receiver with constant in-degree k. Returns [|V|, 2 Fe]."""
    assert (edgeUnitVector is None) != (edgeUnitVectorInverse is None), \
        "Either edgeUnitVector or edgeUnitVectorInverse must be provided."
    if edgeUnitVectorInverse is None:
        n = int(coarse_mask.sum()) if coarse_mask is not None else _num_nodes_of(edge_index)
        edgeUnitVectorInverse = torch.linalg.pinv(edgeUnitVector.reshape(n, -1, 2))
    n, k = int(edgeUnitVectorInverse.size(0)), int(edgeUnitVectorInverse.size(2))
    if edge_attr.size(0) != n * k:
        raise ValueError(f"{edge_attr.size(0)} edges cannot be viewed as {n} nodes x {k} incoming edges")
    return ops.edge_scalar_to_node_vector(edge_attr, edgeUnitVectorInverse, n, k)


# ------------------------------------------------------------------------------------- MP
# Rounded-bf16 mode only (BASELINE config 3): EdgeMP's message rows — read again only by the next EdgeMP's message launch, which applies
# SELU and rounds them to bf16 — are stored by the launch that fuses the aggregation as bf16(SELU(row)) (G4C_DTYPE_BF16_SELU): half the
# bytes of the largest tensors of a REMuS-GNN step, and bit for bit the operand the reader would have formed from fp32 rows (the
# aggregate still sees the un-activated fp32 rows).  Round 2 stored bf16(row) and let the reader apply SELU: two roundings, max
# deviation from the fp32 reference 3.6e-2 -> 6.5e-2 at 20k nodes, which is why it was opt-in then.
COMPACT_MESSAGES = True
# Rounded-bf16 mode: the first layer of a message MLP is hoisted as in the other arithmetics (products of the bf16-rounded node rows
# with the bf16-rounded weights, fp32 accumulate: the operands the unhoisted launch forms, added in another order).  Measured on
# config 3 (REMuS-GNN, 100k nodes): the level-1 angle launch is not HBM-bound on its gathers (reading them from bf16 copies of the
# sender rows: 1160 -> 1130 us) but on per-tile latency and vector work; hoisted, with the products from the producer's launch
# (g4c_mlp_forward_heads_bf16) and the message launch on mlp_ws_kernel<SP = 1>, it takes 1160 -> 800 us.
HOIST_BF16 = True
# Round 5, rounded-bf16 mode: those hoisted first-layer products are STORED as bf16 (g4c_mlp_forward_heads_bf16_out / _bf16_out) and
# widened when the message launch adds them: REMuS-GNN's level-1 angle launch gathers 2 x 2.5 M product rows per EdgeMP — its
# largest stream — at half the bytes.  One more rounding (relative 2^-9) of a pre-activation term whose operands were rounded to
# bf16 already; error against the fp32 reference restatement: scripts/remus_bf16_err.py and the 20k-node REMuS parity test.
PRODUCTS_BF16 = True
# Round 6, rounded-bf16 mode: message launches over receivers of one uniform in-degree (REMuS-GNN's angle launches: every edge of a
# k-nearest-neighbour graph receives k angles) run on the row-split kernel (mlp_rs.hip, mlp_rs1_kernel): a wave owns 16 rows through all
# layers, the weights stay in LDS, the aggregation is a segmented scan over the lanes of a row group.  Its bf16 rows — compact messages
# and hoisted products — are in its own column order (ops.RsOrderedRows), 16 contiguous bytes per lane: the level-1 angle launch of
# config 3 is bound by the NUMBER of memory instructions and 32-byte pieces, not by HBM bytes (2.5 M rows: 655 us on
# mlp_ws_kernel<SP = 1>, 622 us on this kernel with natural-order 8-byte pieces, 520 us with 16-byte pieces).
ROW_SPLIT_BF16 = __import__('os').environ.get('G4C_ROW_SPLIT_BF16', '1') != '0'      # (environment: same-box A/B runs)
# ... and the aggregate of such a launch is stored as bf16 rows in the same order (G4C_AGG_OUT_BF16) when its reader is the layer's update
# MLP in one launch: that reader rounds it to bf16 on load, so the operand is the same and two launches move half its bytes (2.5M-row
# angle launch 521 -> 493 us, the 500k-row edge update 284 -> 254 us).
AGGREGATE_BF16 = __import__('os').environ.get('G4C_AGGREGATE_BF16', '1') != '0'
# ... and the edge latents BETWEEN consecutive EdgeMPs of a level — read only by the next update MLP, which rounds them to bf16 on load —
# are stored as bf16 rows by the update launch (g4c_mlp_forward_heads_bf16_rows): the same operand, half the bytes in both launches.
COMPACT_LATENTS = __import__('os').environ.get('G4C_COMPACT_LATENTS', '1') != '0'
# ... and the update MLP of such a layer — [bf16 aggregate | bf16 e] -> two layers -> LayerNorm -> SELU -> e' (+ the next layer's two
# product heads) — runs on the row-split UPDATE kernel (mlp_rs.hip, mlp_rs2_kernel: the five 128 x 128 weight blocks are all of a CU's
# LDS, the bias / LayerNorm vectors live in registers), e' as bf16 rows in the kernel's order between consecutive EdgeMPs.
UPDATE_ROW_SPLIT = __import__('os').environ.get('G4C_UPDATE_ROW_SPLIT', '1') != '0'
RS1_MIN_ROWS = 20000


# One launch per MP layer (round 5, ops.mp_layer_forward / g4c_mp_layer_forward_bx6): message MLP + aggregation + node MLP (+ the next
# layer's products) in the same persistent workgroups.  For launches whose time is the dependent chain inside each kernel — the coarse
# levels of a multi-scale model, whole small meshes (a rank's share of a partitioned mesh) — not for throughput-bound ones: at
# FUSE_LAYER_MAX_ROWS edge rows and beyond the node update of 100k nodes runs faster as its own chip-filling launch.
FUSE_LAYER = os.environ.get("G4C_FUSE_LAYER", "1") != "0"
FUSE_LAYER_MIN_ROWS = 2048
FUSE_LAYER_MAX_ROWS = 120_000


def will_fuse_layer(msg_mlp: MLP, upd_mlp: MLP, edge_index: Tensor, n_nodes: int) -> bool:
    """The part of the one-launch-per-MP-layer decision that the PRODUCER of the layer's node input can evaluate too (the
    model's `_launch_for` emits the layer's products exactly when the layer will fuse — one predicate, not two copies): arithmetic,
    mode, the level's size, rows in target order in tiles of whole segments, both MLPs two or three 128-wide layers."""
    if not FUSE_LAYER or ops.grad_mode() or ops.mlp_precision() != "f16x3":
        return False
    if not (FUSE_LAYER_MIN_ROWS <= int(edge_index.size(1)) < FUSE_LAYER_MAX_ROWS):
        return False
    lm, lu = msg_mlp._linears(), upd_mlp._linears()
    if len(lm) != len(lu) or len(lm) not in (2, 3) or any(l.out_features != 128 for l in lm + lu):
        return False
    if msg_mlp.input_size != 384 or upd_mlp.input_size != 256:
        return False
    _, csr = plan.edge_csr(edge_index, n_nodes)
    return csr.perm is None and csr.tiles() is not None


def _can_fuse_layer(msg_mlp: MLP, upd_mlp: MLP, v: Tensor, e, edge_index: Tensor, csr, v_src, n_targets, v_out, compact_messages: bool) -> bool:
    if v_src is not None or n_targets is not None or v_out is not None or compact_messages:
        return False
    if not will_fuse_layer(msg_mlp, upd_mlp, edge_index, int(v.size(0))):
        return False
    e_t = e.tensor if isinstance(e, Source) else e
    if isinstance(e, Source) and (e.index is not None or e.col0 != 0 or e.negate or e.segments is not None or e.additive):
        return False
    return int(v.size(1)) == 128 and int(e_t.size(1)) == 128 and e_t.dtype == torch.float32


def _fused_layer(msg_mlp: MLP, upd_mlp: MLP, v: Tensor, e_src: Source, ep, csr, mean: bool, act_code: int, products,
                 next_msg: Optional[MLP], keep_e: bool):
    """The MP layer as one launch.  Returns (v', e' or None, next products or None)."""
    H = 128
    dev = v.device
    n_t = int(v.size(0))
    if products is None:          # the node-side products of this layer's first layer: W1[:, H:2H] v, W1[:, 2H:3H] v (MLP.run_hoisted)
        products = [ops.mlp_forward(msg_mlp._packed_cols("hoist1", H * (1 + j), H * (2 + j), [H], [False], True), [Source(v)], n_t) for j in range(2)]
    pk_msg = msg_mlp._packed_cols("hoist", 0, H, [H], [e_src.negate], False)
    srcs = [e_src, Source(products[0], index=ep.row, additive=True), Source(products[1], index=ep.col, additive=True)]
    heads = None
    if next_msg is not None and next_msg.input_size == 3 * H and next_msg._linears()[0].out_features == H:
        pk_upd = upd_mlp._heads_packed([H, H], next_msg, next_msg.input_size - 2 * H, [H, H])
        if pk_upd is not None:
            heads = [torch.empty((n_t, H), dtype=torch.float32, device=dev) for _ in range(2)]
    if heads is None:
        pk_upd = upd_mlp.packed([H, H], [False, False])
    e_new, v_new, _ = ops.mp_layer_forward(pk_msg, srcs, ep.n_edges, csr, mean, pk_upd, v, act_code, store_rows=keep_e, head_outs=heads)
    return v_new, e_new, heads


def _mp_step(msg_mlp: MLP, upd_mlp: MLP, v: Tensor, e: Tensor, index: Tensor, aggr: str, act_code: int,
             e_pre_act: int = _lib.ACT_NONE, v_src: Optional[Tensor] = None,
             products: Optional[Sequence[Tensor]] = None, next_msg: Optional[MLP] = None, keep_e: bool = True,
             n_targets: Optional[int] = None, v_out: Optional[Tensor] = None, compact_messages: bool = False,
             next_graph: Optional[Tuple[int, "plan.CsrPlan"]] = None, compact_v: bool = False):
    """Shared body of GNBlock / EdgeMP / DownEdgeMP (nn/blocks.py:175-186,322-333,360-381):
        e' = msg_mlp([e | s[row] | v[col]]);  agg = reduce(e' -> col);  v' = act(upd_mlp([agg | v])).
    Returns (v', e') where e' is stored WITHOUT the activation: the aggregation consumes the raw
    messages, and the consumer of e' applies the activation while loading (`e_pre_act` here is that
    pending activation of the incoming `e`).  `v_src` (DownEdgeMP) gathers sender rows from another tensor.
    `keep_e=False`: the caller discards e' (the last MP layer of a level, nn/mus_gnn.py:199-200,211-212): when the edge launch
    can reduce its own rows they are then not stored and None is returned for e'.
    `products` = (W1[:, H:2H] v, W1[:, 2H:3H] v) of msg_mlp's first layer when the launch that produced `v` already
    multiplied them; `next_msg` = the message MLP of the MP layer that will consume v' on the SAME graph: the node
    launch then emits its products as well and a third value (those products, or None) is returned.
    `compact_messages` (EdgeMP: the returned e' is only ever read by the next EdgeMP's message launch): in the rounded-bf16 mode
    the launch that fuses the aggregation stores bf16(SELU(e')) (ops.mlp_forward rows_dtype / rows_act) — exactly the operand that
    reader forms; REMuS-GNN's angle launches are HBM-bound on these rows.  A bf16 result is ALREADY ACTIVATED: its reader passes
    `e_pre_act = ACT_NONE` (see `pending_act`).
    `next_graph` = (rows, receiver CSR) of the launch `next_msg` will run when it is NOT over this step's graph (DownEdgeMP: the
    consumer is the coarse level's first EdgeMP): decides whether the consumer hoists and which column order its products take.
    `compact_v` (rounded-bf16 mode; the caller guarantees that v' is only ever read as an input block of the next layer's update MLP, which
    rounds it to bf16 on load — EdgeMP's edge latents between consecutive EdgeMPs of a level): v' comes back as bf16 rows (feature order),
    the same operand at half the bytes in both launches.
    `n_targets` / `v_out` (partitioned sub-meshes, partition_remus.py): only the first `n_targets` rows of `v` are targets (the
    rows behind them are halo rows, read as senders only); v' for those rows is written into `v_out`."""
    if aggr not in ("mean", "sum", "add"):
        raise ValueError(f"unsupported aggr {aggr!r}")
    n_t = int(v.size(0)) if n_targets is None else int(n_targets)
    if n_targets is not None and next_msg is not None:
        raise NotImplementedError("next_msg with n_targets")
    ep, csr = plan.edge_csr(index, n_t)
    senders = v if v_src is None else v_src
    mean = aggr == "mean"
    e_src = e if isinstance(e, Source) else Source(e, pre_act=e_pre_act)
    if next_graph is None and _can_fuse_layer(msg_mlp, upd_mlp, v, e, index, csr, v_src, n_targets, v_out, compact_messages):
        v_new, e_new, nxt = _fused_layer(msg_mlp, upd_mlp, v, e_src, ep, csr, mean, act_code, products, next_msg, keep_e)
        return (v_new, e_new, nxt) if next_msg is not None else (v_new, e_new)
    if ops.can_fuse_aggregation(csr, msg_mlp.output_size):
        # the edge launch reduces the rows it has just computed (whole CSR segments per row tile, g4c_mlp_forward_bx6_agg):
        # no second pass over the messages; with keep_e=False (the model discards e', nn/mus_gnn.py:199-200) they are not
        # even written
        kw = dict(store_rows=keep_e, rows_dtype=torch.bfloat16 if compact_messages and COMPACT_MESSAGES else None,
                  rows_act=_lib.ACT_SELU if compact_messages and COMPACT_MESSAGES else _lib.ACT_NONE)
        gathered = [(senders, ep.row), (v, ep.col)]
        agg = None
        if (AGGREGATE_BF16 and ops.mlp_precision() == "bf16" and msg_mlp.output_size == 128 and int(v.size(1)) == 128
                and upd_mlp.fits_one_launch() and ops.effective_precision([128, 128]) == "bf16"):
            # rounded-bf16 mode, message launch on the row-split kernel: the aggregate's one reader — this layer's update MLP — rounds it
            # to bf16 on load, so it is stored that way (in the kernel's column order; the update MLP's pack takes the block's columns in
            # the same order): the same operand, half the bytes in both launches
            agg16 = torch.empty((csr.n_seg, 128), dtype=torch.bfloat16, device=v.device)
            if msg_mlp._rs1_takes([e_src], gathered, ep.n_edges, _lib.ACT_NONE, products, dict(kw, agg=(csr, agg16, mean))):
                agg = agg16
        if agg is None:
            agg = torch.empty((csr.n_seg, msg_mlp.output_size), dtype=torch.float32, device=v.device)
        e_new = msg_mlp.run_hoisted([e_src], gathered, ep.n_edges, products=products, agg=(csr, agg, mean), **kw)
        agg_src = Source(ops.RsOrderedRows.tag(agg) if agg.dtype == torch.bfloat16 else agg)
    elif ops.can_aggregate_on_load(csr, msg_mlp.output_size, [msg_mlp.output_size, int(v.size(1))]):
        # the node launch averages each target's messages while it gathers its input (g4c_src_t.seg_off): no separate
        # aggregation pass, no aggregate written to / re-read from HBM
        e_new = msg_mlp.run_hoisted([e_src], [(senders, ep.row), (v, ep.col)], ep.n_edges, products=products)
        agg_src = Source(e_new, segments=csr, seg_mean=mean)
    elif ops.grad_mode():
        e_new = msg_mlp.run_hoisted([e_src], [(senders, ep.row), (v, ep.col)], ep.n_edges)
        agg_src = Source(ops.segment_reduce(e_new, csr, mean))
    else:
        agg = torch.empty((csr.n_seg, msg_mlp.output_size), dtype=torch.float32, device=v.device)
        e_new = msg_mlp.run_hoisted([e_src], [(senders, ep.row), (v, ep.col)], ep.n_edges,
                                    products=products, agg=(csr, agg, mean))
        agg_src = Source(agg)
    v16 = (compact_v and COMPACT_LATENTS and PRODUCTS_BF16 and HOIST_BF16 and ops.mlp_precision() == "bf16" and upd_mlp.output_size == 128 and upd_mlp.fits_one_launch()
           and n_targets is None and v_out is None and not ops.grad_mode() and ops.effective_precision([128, int(v.size(1))]) == "bf16")
    if next_msg is not None:
        nxt = None
        nx_rows, nx_csr = (ep.n_edges, csr) if next_graph is None else next_graph
        if nx_rows >= HOIST_MIN_ROWS:        # the consumer will hoist: give it its node-side terms from this launch
            w = upd_mlp.output_size          # (width of v', the node input of the next layer's message MLP)
            nxt = upd_mlp.run_with_heads([agg_src, Source(v)], int(v.size(0)), act_code, next_msg,
                                         next_msg.input_size - 2 * w, [w, w], rs_rows=next_msg.rs1_ready(nx_rows, nx_csr),
                                         out=torch.empty((int(v.size(0)), 128), dtype=torch.bfloat16, device=v.device) if v16 else None)
        if nxt is None:
            out16 = torch.empty((int(v.size(0)), 128), dtype=torch.bfloat16, device=v.device) if v16 else None
            return upd_mlp.run_coded([agg_src, Source(v)], int(v.size(0)), act_code, out=out16), e_new, None
        return nxt[0], e_new, nxt[1]
    v_new = upd_mlp.run_coded([agg_src, Source(v)], n_t, act_code,
                              out=torch.empty((n_t, 128), dtype=torch.bfloat16, device=v.device) if v16 else v_out)
    return v_new, e_new


def compact_latents_now(width: int = 128) -> bool:
    """Whether a launch whose output rows are read by nothing but the next update MLP may store them as bf16 (COMPACT_LATENTS)."""
    return bool(COMPACT_LATENTS and PRODUCTS_BF16 and HOIST_BF16 and ops.mlp_precision() == "bf16" and width == 128 and not ops.grad_mode()
                and ops.effective_precision([128, 128]) == "bf16")


def pending_act(e: Optional[Tensor]) -> int:
    """Activation still to be applied to the messages `_mp_step` returned: SELU for fp32 rows (stored raw, the aggregation needed them
    so), none for compact bf16 rows (stored activated)."""
    return _lib.ACT_NONE if (e is not None and torch.is_tensor(e) and e.dtype == torch.bfloat16) else _lib.ACT_SELU


def _public_mp(msg_mlp: MLP, upd_mlp: MLP, v, e, index, aggr, activation, v_src=None):
    code = _lib.act_code(activation)
    v_new, e_new = _mp_step(msg_mlp, upd_mlp, v, e, index, aggr, _lib.ACT_NONE if code is None else code, v_src=v_src)
    if activation is not None:
        if code is None:
            v_new, e_new = activation(v_new), activation(e_new)
        elif ops.grad_mode():                  # (out of place: e_new is an autograd output)
            e_new = torch.nn.functional.selu(e_new) if code == _lib.ACT_SELU else (torch.tanh(e_new) if code == _lib.ACT_TANH else e_new)
        else:
            ops.activation_(e_new, code)
    return v_new, e_new


class GNBlock(nn.Module):
    r"""Graph-network block (Battaglia et al. 2018; reference: nn/blocks.py:147-186).

    Args:
        edge_mlp_args (Tuple): Arguments for the MLP updating the edge features.
        node_mlp_args (Tuple): Arguments for the MLP updating the node features.
        aggr (str, optional): 'mean' or 'sum'. Defaults to 'mean'.
    """

    def __init__(self, edge_mlp_args: Tuple, node_mlp_args: Tuple, aggr: str = 'mean'):
        super().__init__()
        self.edge_mlp = MLP(*edge_mlp_args)
        self.node_mlp = MLP(*node_mlp_args)
        self.aggr = aggr

    def reset_parameters(self):
        for m in (self.node_mlp, self.edge_mlp):
            if m is not None and hasattr(m, 'reset_parameters'):
                m.reset_parameters()

    def step(self, v: Tensor, e: Tensor, edge_index: Tensor, act_code: int, e_pre_act: int = _lib.ACT_NONE,
             products: Optional[Sequence[Tensor]] = None, next_msg: Optional[MLP] = None, keep_e: bool = True):
        """Internal form used by the model programs: returns (act(v'), raw e') — and, when `next_msg` (the edge MLP of
        the next MP layer on the same graph) is given, a third value: that layer's `products` or None (see _mp_step)."""
        return _mp_step(self.edge_mlp, self.node_mlp, v, e, edge_index, self.aggr, act_code, e_pre_act,
                        products=products, next_msg=next_msg, keep_e=keep_e)

    def forward(self, v: Tensor, e: Tensor, edge_index: Tensor, *, activation=None) -> Tuple[Tensor, Tensor]:
        return _public_mp(self.edge_mlp, self.node_mlp, v, e, edge_index, self.aggr, activation)


# Alias for GNBlock
MP = GNBlock


class DownMP(nn.Module):
    r"""DownMP from Lino et al. (2022) (reference: nn/blocks.py:193-237).

    Args:
        down_mlp_args (Tuple): Arguments for the MLP of the downsampling edge-model.
        hr_graph_idx (int): The index of the high-resolution graph.
    """

    def __init__(self, down_mlp_args: Tuple, hr_graph_idx: int):
        super().__init__()
        self.down_mlp = MLP(*down_mlp_args)
        self.hr_graph_idx = hr_graph_idx
        self.lr_graph_idx = hr_graph_idx + 1
        self.reset_parameters()

    def reset_parameters(self):
        for item in [self.down_mlp]:
            if hasattr(item, 'reset_parameters'):
                item.reset_parameters()

    def pool(self, graph: Graph, field: Tensor, edge_index: Tensor, edge_attr: Tensor, activation=None,
             e_pre_act: int = _lib.ACT_NONE, target_major: bool = False):
        """Functional core: returns (field_l, edge_index_l, edge_attr_l) without touching the Graph.
        `e_pre_act`: activation still pending on `edge_attr` (applied while pooling)."""
        h, l = self.hr_graph_idx, self.lr_graph_idx
        rel = getattr(graph, f'e_{h}{l}')
        csr = plan.cluster_plan(getattr(graph, f'cluster_{l}'), getattr(graph, f'mask_{l}'))
        m = self.down_mlp.run([Source(rel), Source(field)], int(field.size(0)))
        code = _lib.act_code(activation)
        pooled = ops.segment_reduce(m, csr, True, _lib.ACT_NONE if code is None else code)
        pooled = _finish(pooled, activation, code)
        # (target_major: the models' internal coarse edge order, see plan.pool_edge_plan; the public forward keeps `coalesce` order)
        pp = plan.pool_edge_plan(getattr(graph, f'idx{h}_to_idx{l}'), edge_index, target_major)
        ea_l = ops.segment_reduce(edge_attr, pp.csr, True, src_act=e_pre_act)
        return pooled, pp.edge_index, ea_l

    def forward(self, graph: Graph, activation: Optional[Callable] = None) -> Graph:
        field, ei, ea = self.pool(graph, graph.field, graph.edge_index, graph.edge_attr, activation)
        graph.pos = getattr(graph, f'pos_{self.lr_graph_idx}')
        graph.field, graph.edge_index, graph.edge_attr = field, ei, ea
        return graph


class UpMP(nn.Module):
    r"""UpMP from Lino et al. (2022) (reference: nn/blocks.py:240-290).

    Args:
        up_mlp_args (Tuple): Arguments for the MLP of the upsampling edge-model.
        lr_graph_idx (int): The index of the low-resolution graph.
    """

    def __init__(self, up_mlp_args: Tuple, lr_graph_idx: int):
        super().__init__()
        self.up_mlp = MLP(*up_mlp_args)
        self.lr_graph_idx = lr_graph_idx
        self.hr_graph_idx = lr_graph_idx - 1
        self.reset_parameters()

    def reset_parameters(self):
        for item in [self.up_mlp]:
            if hasattr(item, 'reset_parameters'):
                item.reset_parameters()

    def sources(self, graph: Graph, field_lr: Tensor, field_hr_old: Tensor) -> List[Source]:
        """Input blocks of up_mlp: [-e_hl | field_l[parent] | field_hr_old]; the sign flip is folded into the packed weights."""
        h, l = self.hr_graph_idx, self.lr_graph_idx
        parent = plan.index32(getattr(graph, f'idx{h}_to_idx{l}'))
        return [Source(getattr(graph, f'e_{h}{l}'), negate=True), Source(field_lr, parent), Source(field_hr_old)]

    def unpool(self, graph: Graph, field_lr: Tensor, field_hr_old: Tensor, activation=None) -> Tensor:
        return self.up_mlp.run(self.sources(graph, field_lr, field_hr_old), int(field_hr_old.size(0)), activation=activation)

    def forward(self, graph: Graph, field_hr_old: Tensor, pos_hr: Tensor, activation: Optional[Callable] = None) -> Graph:
        graph.field = self.unpool(graph, graph.field, field_hr_old, activation)
        graph.pos = pos_hr
        return graph


# ------------------------------------------------------------------------------------- REMuS
class EdgeMP(nn.Module):
    r"""EdgeMP from Lino et al. (2022) (reference: nn/blocks.py:293-333): a GNBlock with
    (edges, angles) in the roles of (nodes, edges)."""

    def __init__(self, angle_mlp_args: Tuple, edge_mlp_args: Tuple, aggr: str = "mean"):
        super().__init__()
        self.angle_mlp = MLP(*angle_mlp_args)
        self.edge_mlp = MLP(*edge_mlp_args)
        self.aggr = aggr
        self.reset_parameters()

    def reset_parameters(self):
        for item in [self.edge_mlp, self.angle_mlp]:
            if hasattr(item, 'reset_parameters'):
                item.reset_parameters()

    def step(self, e: Tensor, a: Tensor, angle_index: Tensor, act_code: int, a_pre_act: int = _lib.ACT_NONE,
             products: Optional[Sequence[Tensor]] = None, next_msg: Optional[MLP] = None, keep_e: bool = True, compact_e: bool = False):
        """Internal form: returns (act(e'), raw a') (+ the next EdgeMP's `products` when `next_msg` is given, see GNBlock.step).
        `compact_e`: e' is read by nothing but the next EdgeMP.step of the level (_mp_step compact_v).
        The model feeds a' only to the next EdgeMP.step of the level, so it may come back as bf16 in the rounded-bf16 mode
        (_mp_step compact_messages); the public forward always returns fp32."""
        return _mp_step(self.angle_mlp, self.edge_mlp, e, a, angle_index, self.aggr, act_code, a_pre_act,
                        products=products, next_msg=next_msg, keep_e=keep_e, compact_messages=True, compact_v=compact_e)

    def forward(self, e: Tensor, a: Tensor, angle_index: Tensor, *, activation=None) -> Tuple[Tensor, Tensor]:
        return _public_mp(self.angle_mlp, self.edge_mlp, e, a, angle_index, self.aggr, activation)


class DownEdgeMP(nn.Module):
    r"""DownEdgeMP from Lino et al. (2022) (reference: nn/blocks.py:336-381)."""

    def __init__(self, angle_mlp_args: Tuple, edge_mlp_args: Tuple):
        super().__init__()
        self.angle_mlp = MLP(*angle_mlp_args)
        self.edge_mlp = MLP(*edge_mlp_args)
        self.reset_parameters()

    def reset_parameters(self):
        for item in [self.edge_mlp, self.angle_mlp]:
            if hasattr(item, 'reset_parameters'):
                item.reset_parameters()

    def forward(self, e1: Tensor, e2: Tensor, a12: Tensor, angle_index12: Tensor, *, activation=None) -> Tensor:
        return self.step(e1, e2, a12, angle_index12, activation)[0]

    def step(self, e1: Tensor, e2: Tensor, a12: Tensor, angle_index12: Tensor, activation=None, next_msg: Optional[MLP] = None,
             next_graph=None) -> Tuple[Tensor, Optional[Sequence[Tensor]]]:
        """Internal form: (e2', products) — `next_msg` = the angle MLP of the coarse level's EdgeMP that reads e2' next, `next_graph` its
        (angle rows, receiver CSR): its hoisted first-layer products then come out of this block's edge launch (see GNBlock.step)."""
        code = _lib.act_code(activation)
        if next_msg is None or code is None:
            e2_new, _ = _mp_step(self.angle_mlp, self.edge_mlp, e2, a12, angle_index12, "mean",
                                 _lib.ACT_NONE if code is None else code, v_src=e1, keep_e=False)          # (a12' is read by nobody)
            return _finish(e2_new, activation, code), None
        e2_new, _, prods = _mp_step(self.angle_mlp, self.edge_mlp, e2, a12, angle_index12, "mean", code, v_src=e1, keep_e=False,
                                    next_msg=next_msg, next_graph=next_graph, compact_v=True)      # (e2' is read by that EdgeMP only)
        return e2_new, prods


class UpEdgeMP(nn.Module):
    r"""UpEdgeMP from Lino et al. (2022) (reference: nn/blocks.py:384-456)."""

    def __init__(self, up_mlp_args: Tuple):
        super().__init__()
        self.up_mlp = MLP(*up_mlp_args)
        self.reset_parameters()

    def reset_parameters(self):
        for item in [self.up_mlp]:
            if hasattr(item, 'reset_parameters'):
                item.reset_parameters()

    def forward(self, pos: Tensor, y_idx_21: Tensor, x_idx_21: Tensor, weights_21: Tensor, edge_attr2: Tensor,
                edge_index2: Tensor, edgeUnitVectorInverse2: Tensor, coarse_mask2: Tensor, edge_attr1: Tensor,
                edge_index1: Tensor, edgeUnitVector1: Tensor, coarse_mask1=None, *, activation=None) -> Tensor:
        return self.step(pos, y_idx_21, x_idx_21, weights_21, edge_attr2, edge_index2, edgeUnitVectorInverse2, coarse_mask2, edge_attr1,
                         edge_index1, edgeUnitVector1, coarse_mask1, activation=activation)[0]

    def step(self, pos: Tensor, y_idx_21: Tensor, x_idx_21: Tensor, weights_21: Tensor, edge_attr2: Tensor,
             edge_index2: Tensor, edgeUnitVectorInverse2: Tensor, coarse_mask2: Tensor, edge_attr1: Tensor,
             edge_index1: Tensor, edgeUnitVector1: Tensor, coarse_mask1=None, *, activation=None, next_msg: Optional[MLP] = None,
             next_graph=None) -> Tuple[Tensor, Optional[Sequence[Tensor]]]:
        """Internal form: (e1', products) — with `next_msg` / `next_graph` (the fine level's next EdgeMP, see DownEdgeMP.step) the
        per-edge launch also emits that EdgeMP's hoisted first-layer products."""
        n_total, nfeat = int(pos.size(0)), int(edge_attr2.size(1))
        # 1- edge scalars of level 2 -> node vectors [|V_2|, 2F]
        v2 = edgeScalarToNodeVector(edge_attr2, edge_index2, edgeUnitVectorInverse=edgeUnitVectorInverse2,
                                    coarse_mask=coarse_mask2)
        # 2- interpolate to the nodes of level 1, written at their level-1 numbering
        if ops.grad_mode() and v2.requires_grad:       # (recorded for autograd: the interpolation returns the full tensor)
            from .. import autograd as _ag
            v1 = _ag.weighted_segment_mean(v2, plan.index32(x_idx_21), weights_21, plan.segments_of_sorted(y_idx_21), n_total,
                                           None if coarse_mask1 is None else plan.mask_index32(coarse_mask1))
        elif coarse_mask1 is None:
            v1 = torch.empty(n_total, 2 * nfeat, dtype=torch.float32, device=pos.device)
            knn_interpolate(v2, y_idx_21, x_idx_21, weights_21, out=v1)
        else:
            v1 = torch.zeros(n_total, 2 * nfeat, dtype=torch.float32, device=pos.device)
            knn_interpolate(v2, y_idx_21, x_idx_21, weights_21, out=v1, out_idx32=plan.mask_index32(coarse_mask1))
        # 3- project node vectors on the level-1 edges [|E_1|, F]
        ep = plan.edge_plan(edge_index1)
        e1 = ops.project_to_edges(v1, ep.col, edgeUnitVector1, ep.n_edges, nfeat)
        # 4- skip connection + per-edge MLP
        code = _lib.act_code(activation)
        if next_msg is not None and code is not None and next_graph is not None and next_graph[0] >= HOIST_MIN_ROWS:
            w = self.up_mlp.output_size
            out16 = torch.empty((ep.n_edges, 128), dtype=torch.bfloat16, device=pos.device) if compact_latents_now(w) else None
            got = self.up_mlp.run_with_heads([Source(e1), Source(edge_attr1)], ep.n_edges, code, next_msg, next_msg.input_size - 2 * w, [w, w],
                                             rs_rows=next_msg.rs1_ready(*next_graph), out=out16)      # (e1' is read by that EdgeMP only)
            if got is not None:
                return got
        return self.up_mlp.run([Source(e1), Source(edge_attr1)], ep.n_edges, activation=activation), None

"""Training path: autograd for the fused blocks (design and measurements in DESIGN.md §7).

The reference trains through torch autograd over `cat` / index / `nn.Linear` / SELU / LayerNorm / `scatter`
(GNN.fit, nn/model.py:152-301; blocks nn/blocks.py:117-290), keeping every intermediate of every block alive between
the passes (per MP layer at 100k nodes: the [E, 3H] concatenation, three [E, H] hidden tensors, ...).

Here the forward of a block stays ONE fused launch, recorded by a `torch.autograd.Function`:

  * forward: g4c_mlp_forward_bx6_save — the fused kernel also writes each layer's output rows (SELU(hidden), pre-LayerNorm
    rows), so the backward recomputes no product (SAVE_ACTIVATIONS; False keeps only block inputs / outputs and
    re-forms the hidden layers in the backward with single-layer launches of the same kernel);
  * weight + bias gradients of the 128-wide layers: g4c_weight_grad (one pass over dZ and A, fp32 MFMA, partial tiles added in a
    fixed order); other widths: a split-row strided-batched rocBLAS GEMM + g4c_colsum;
  * hidden layers of an edge MLP's backward: one launch of the fused kernel with a multiplicative SELU-slope epilogue
    (`backward_chain`); other input-gradient products: single-layer launches (`linear`), rocBLAS below 64k rows;
  * the first layer of a block gathered through an index is differentiated on the gathered tensor's rows (the adjoint of the
    inference path's hoisting): segmented sums of dZ by the index, then node-sized products;
  * LayerNorm / activation adjoints, input columns, bias sums: train_ops.hip (g4c_layernorm_grad, g4c_act_grad,
    g4c_train_gather, g4c_colsum); the adjoint of an index gather is g4c_segment_reduce on the CSR plan of the index (the
    reference's autograd does this scatter with atomics), of an aggregation g4c_segment_broadcast.

No atomics anywhere: two runs of a training step produce bit-identical gradients.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import Tensor

from . import _lib, ops, plan
from .ops import Source, _ld


# ------------------------------------------------------------------------------------- optional phase timing
PROFILE = None       # set to {} (scripts/bench_train.py --phases): per-phase lists of HIP event pairs of the backward pass


class _phase:
    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if PROFILE is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if PROFILE is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            PROFILE.setdefault(self.name, []).append((self.a, b))
        return False


def profile_summary() -> dict:
    torch.cuda.synchronize()
    return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in (PROFILE or {}).items()}


# ------------------------------------------------------------------------------------- kernel wrappers
def _buf(rows: int, cols: int, dev) -> Tensor:
    return torch.empty((rows, cols), dtype=torch.float32, device=dev)


def train_gather(src: Tensor, dst: Tensor, dcol0: int, scol0: int, width: int, idx32: Optional[Tensor], pre_act: int,
                 negate: bool, n_rows: int, accumulate: bool = False) -> None:
    lib = _lib.load()
    dev = _lib.require_hip(src, dst, idx32)
    _lib.check(lib.g4c_train_gather(_lib.ptr(src), _ld(src), scol0, _lib.ptr(idx32), pre_act, 1 if negate else 0, _lib.ptr(dst),
                                    _ld(dst), dcol0, width, n_rows, 1 if accumulate else 0, _lib.stream_handle(dev)))


def act_grad(dy: Tensor, ref: Tensor, act: int, from_input: bool, out: Optional[Tensor] = None) -> Tensor:
    """dy * act'(.) with `ref` the activation's output (from_input False) or input (True)."""
    if act == _lib.ACT_NONE:
        return dy
    lib = _lib.load()
    dev = _lib.require_hip(dy, ref)
    if out is None:
        out = _buf(int(dy.size(0)), int(dy.size(1)), dev)
    _lib.check(lib.g4c_act_grad(_lib.ptr(dy), _ld(dy), _lib.ptr(ref), _ld(ref), 1 if from_input else 0, act, _lib.ptr(out),
                                _ld(out), int(dy.size(1)), int(dy.size(0)), _lib.stream_handle(dev)))
    return out


def colsum(x: Tensor) -> Tensor:
    lib = _lib.load()
    dev = _lib.require_hip(x)
    rows, width = int(x.size(0)), int(x.size(1))
    scratch = _buf(int(lib.g4c_colsum_partials(rows)), width, dev)
    out = torch.empty(width, dtype=torch.float32, device=dev)
    _lib.check(lib.g4c_colsum(_lib.ptr(x), _ld(x), width, rows, _lib.ptr(scratch), _lib.ptr(out), _lib.stream_handle(dev)))
    return out


def layernorm_grad(z: Tensor, gamma: Tensor, dy: Tensor, eps: float):
    """(dz, dgamma, dbeta) of y = LayerNorm(z) * gamma + beta."""
    lib = _lib.load()
    dev = _lib.require_hip(z, gamma, dy)
    rows, width = int(z.size(0)), int(z.size(1))
    dz = _buf(rows, width, dev)
    partial = _buf(int(lib.g4c_layernorm_grad_partials(rows)), 2 * width, dev)
    _lib.check(lib.g4c_layernorm_grad(_lib.ptr(z), _ld(z), _lib.ptr(gamma), _lib.ptr(dy), _ld(dy), _lib.ptr(dz), _ld(dz),
                                      _lib.ptr(partial), width, rows, float(eps), _lib.stream_handle(dev)))
    gb = colsum(partial)
    return dz, gb[:width], gb[width:]


def segment_broadcast(dout: Tensor, csr, mean: bool, n_src_rows: int) -> Tensor:
    lib = _lib.load()
    dev = _lib.require_hip(dout, csr.off, csr.perm)
    width = int(dout.size(1))
    full = csr.perm is None and csr.n == n_src_rows
    dsrc = _buf(n_src_rows, width, dev) if full else torch.zeros((n_src_rows, width), dtype=torch.float32, device=dev)
    _lib.check(lib.g4c_segment_broadcast(_lib.ptr(dout), _ld(dout), _lib.ptr(csr.off), _lib.ptr(csr.perm), csr.n_seg, width,
                                         1 if mean else 0, _lib.ptr(dsrc), _ld(dsrc), _lib.stream_handle(dev)))
    return dsrc


# keep the hidden activations of every fused MLP from the forward launch (g4c_mlp_forward_bx6_save) instead of recomputing them
# in the backward pass: ~3 x [rows, 128] fp32 more per MLP between the passes (100k nodes, 3-scale: 7.3 -> 27.5 GB peak), no recompute
# GEMMs.  SAVE_ACTIVATIONS = False: the memory-light recompute path.
SAVE_ACTIVATIONS = True
FUSED_LINEAR = True
HOIST_MIN_ROWS = 32768             # below: the step is host-bound, and hoisting a block costs five more launches than it saves
FUSED_LINEAR_MIN_ROWS = 65536      # below: packing the weights for one launch costs more host time than the fusion saves


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int = _lib.ACT_NONE) -> Tensor:
    """act(x weight^T + bias) for the products of the backward pass (recomputed hidden layers, input gradients), ANY shape, as
    launches of the fused-MLP kernel (one Linear layer each: the forward's arithmetic — exact operand split on the matrix pipe, fp32
    accumulate — with bias and activation in the epilogue): the [M, 128 j] x [<= 128, 128 j] shapes of the published architectures
    are one launch; more than 128 outputs are computed 128 columns at a time, more than four 128-column input blocks group by group,
    each group adding to the previous group's partial sums (an additive source); narrow / ragged input blocks are padded by the
    pack.  No library GEMM (round 5: `torch.mm / addmm` went to rocBLAS for every other shape and below 64k rows)."""
    n_out, k = int(weight.size(0)), int(weight.size(1))
    M = int(x.size(0))
    x = _dense(x)
    if M == 0 or n_out == 0:
        return torch.zeros((M, n_out), dtype=torch.float32, device=x.device)
    # (this is the backward pass: x may hold gradient rows — 1e-6 .. 1e-9 per row for a mean loss over 1e5+ nodes, i.e. fp16's
    # subnormal range, where the two-way fp16 split keeps a few bits only and nothing below 1.5e-11.  Like backward_chain, these
    # products use the three-way bf16 split, which has fp32's exponent range, whatever the forward's arithmetic is.)
    blocks = [(k0, min(128, k - k0)) for k0 in range(0, k, 128)]
    prec = ops.effective_precision([w for _, w in blocks])
    prec = "bf16x6" if prec in ("f16x3", "bf16") else prec          # (the rounded-bf16 forward too — as backward_chain does)
    out = None if n_out <= 128 else torch.empty((M, n_out), dtype=torch.float32, device=x.device)
    w_d, b_d = weight.detach(), (None if bias is None else bias.detach())
    y = None
    for c in range(0, n_out, 128):
        c1 = min(c + 128, n_out)
        partial, i = None, 0
        while i < len(blocks):
            grp = blocks[i:i + (_lib.MAX_SRC if partial is None else _lib.MAX_SRC - 1)]
            i += len(grp)
            last = i >= len(blocks)
            k0, k1 = grp[0][0], grp[-1][0] + grp[-1][1]
            pk = ops.PackedMLP([w_d[c:c1, k0:k1]], [b_d[c:c1] if (b_d is not None and partial is None) else None], None,
                               [w for _, w in grp], [False] * len(grp), precision=prec)
            pk.params = None                                    # (never differentiated through)
            srcs = [Source(x, col0=q0, width=w) for q0, w in grp] + ([Source(partial, additive=True)] if partial is not None else [])
            dst = out[:, c:c1] if (last and out is not None) else None
            partial = ops.mlp_forward(pk, srcs, M, act if last else _lib.ACT_NONE, out=dst)
        y = partial
    return y if out is None else out


FUSED_CHAIN = True


def backward_chain(g: Tensor, weights: Sequence[Tensor], acts: Sequence[Tensor], w_dense: Tensor):
    """Hidden layers of an MLP backward in ONE launch of the fused kernel (g4c_mlp_forward_bx6_save with `mul`):
        D[l] = (D[l+1] W[l]) * selu'(acts[l])   for l = L-1 .. 1,      gX = D[1] W_dense
    with g = D[L] [M, 128], W[l] [128, 128], acts[l] the SELU outputs [M, 128], W_dense [128, 128] the first layer's columns of
    the dense input block.  Returns ({l: D[l]}, gX).  The transposed weights are packed last layer first; every D[l] leaves
    the launch through `save`, the tile itself goes from layer to layer through LDS — instead of a product launch and an
    elementwise pass (read, read, write) per layer."""
    L = len(weights)                                  # weights[l] for l = 0 .. L-1 (weights[0] unused: W_dense replaces it)
    M, dev = int(g.size(0)), g.device
    ws = [weights[l].t() for l in range(L - 1, 0, -1)] + [w_dense.t()]
    pk = ops.PackedMLP(ws, [None] * len(ws), None, [128], [False], precision="bf16x6")
    pk.params = None
    D = {l: _buf(M, 128, dev) for l in range(L - 1, 0, -1)}
    save = [D[l] for l in range(L - 1, 0, -1)] + [None]
    mul = [acts[l] for l in range(L - 1, 0, -1)] + [None]
    gx = ops.mlp_forward(pk, [Source(g)], M, _lib.ACT_NONE, save=save, mul=mul)
    return D, gx


def _wgrad_tile(g: Tensor, blk: Tensor, M: int, with_bias: bool, scratch: Tensor) -> Tensor:
    """One g4c_weight_grad launch: [128 x 128 | 128] = (g^T blk | column sums of g) for 128-wide g and a 128-column window `blk`."""
    lib = _lib.load()
    out = torch.empty(128 * 128 + 128, dtype=torch.float32, device=g.device)
    _lib.check(lib.g4c_weight_grad(_lib.ptr(g), _ld(g), _lib.ptr(blk), _ld(blk), M, _lib.ptr(scratch), _lib.ptr(out),
                                   1 if with_bias else 0, _lib.stream_handle(g.device)))
    return out


def _pad128(t: Tensor, c0: int, c1: int) -> Tensor:
    """Columns [c0, c1) of t as a 128-wide, 16-byte aligned window: a view when they already are one, else a zero-padded copy."""
    w = c1 - c0
    if w == 128 and t.stride(1) == 1 and _ld(t) % 4 == 0 and (t.data_ptr() + 4 * c0) % 16 == 0:
        return t[:, c0:c1]
    p = torch.zeros((int(t.size(0)), 128), dtype=torch.float32, device=t.device)
    ops.copy_cols(t if t.stride(1) == 1 else t.contiguous(), p, 0, scol0=c0, width=w)
    return p


def weight_bias_grad(g: Tensor, a: Tensor, want_bias: bool = True):
    """(dW [N, K], db [N] or None) = (g^T a, column sums of g) by g4c_weight_grad (one pass over both operands, fp32 MFMA, a
    deterministic sum of the workgroups' partial tiles; the bias gradient rides on the first block's pass), 128 x 128 tiles of dW
    at a time.  128-wide g and 128-column blocks of a — every hidden layer of every published architecture — are read in place;
    any other shape (narrow encoder inputs, a 3-wide decoder output, wide generic MLPs, few rows) through zero-padded 128-wide
    copies of the odd blocks (round 5: those went to a split-row rocBLAS GEMM)."""
    M, N, K = int(g.size(0)), int(g.size(1)), int(a.size(1))
    dev = _lib.require_hip(g, a)
    if M == 0:
        return torch.zeros((N, K), dtype=torch.float32, device=dev), (torch.zeros(N, dtype=torch.float32, device=dev) if want_bias else None)
    lib = _lib.load()
    scratch = torch.empty(int(lib.g4c_weight_grad_scratch_floats(M)), dtype=torch.float32, device=dev)
    single = N == 128 and K == 128
    dW = None if single else torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev) if (want_bias and N != 128) else None
    a_blocks = {}
    for n0 in range(0, N, 128):
        n1 = min(n0 + 128, N)
        gp = _pad128(g, n0, n1)
        for j, k0 in enumerate(range(0, K, 128)):
            k1 = min(k0 + 128, K)
            if k0 not in a_blocks:
                a_blocks[k0] = _pad128(a, k0, k1)
            with_bias = want_bias and j == 0
            out = _wgrad_tile(gp, a_blocks[k0], M, with_bias, scratch)
            tile = out[:128 * 128].view(128, 128)
            if single:
                dW = tile
            else:
                dW[n0:n1, k0:k1] = tile[:n1 - n0, :k1 - k0]
            if with_bias:
                if N == 128:
                    db = out[128 * 128:]
                else:
                    db[n0:n1] = out[128 * 128:128 * 128 + n1 - n0]
    return dW, db


def weight_grad(g: Tensor, a: Tensor) -> Tensor:
    """dW = g^T a for g [M, N], a [M, K] (weight_bias_grad without the bias)."""
    return weight_bias_grad(g, a, False)[0]


def _dense(t: Tensor) -> Tensor:
    return t if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous()


# ------------------------------------------------------------------------------------- fused MLP
class _Spec:
    """Non-tensor arguments of one fused-MLP call."""
    __slots__ = ("packed", "meta", "n_rows", "act", "resid_col0", "has_resid", "n_layers", "has_ln", "eps")


class _FusedMLP(torch.autograd.Function):
    """y = act(LN(MLP(cat(sources)))) (+ resid): forward = the fused launch, backward = recompute (module docstring)."""

    @staticmethod
    def forward(ctx, spec: _Spec, *tensors: Tensor):
        n_src = len(spec.meta)
        srcs = []
        for t, m in zip(tensors[:n_src], spec.meta):
            srcs.append(Source(t, index=m["index"], col0=m["col0"], width=m["width"], negate=m["negate"], pre_act=m["pre_act"],
                               segments=m["segments"], seg_mean=m["seg_mean"]))
        resid = tensors[n_src] if spec.has_resid else None
        saves = None
        if SAVE_ACTIVATIONS and spec.packed.precision == "bf16x6" and spec.n_rows > 0:
            dev = tensors[0].device
            saves = [_buf(spec.n_rows, 128, dev) for _ in range(spec.n_layers - 1)] + \
                    [_buf(spec.n_rows, 128, dev) if spec.has_ln else None]
        y = ops.mlp_forward(spec.packed, srcs, spec.n_rows, spec.act, resid=resid, resid_col0=spec.resid_col0, save=saves)
        ctx.spec, ctx.saves = spec, saves
        ctx.save_for_backward(y, *tensors)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        spec: _Spec = ctx.spec
        y, *tensors = ctx.saved_tensors
        ops.bump_weights_epoch()          # a training step is under way: packed weight images are stale after it
        n_src, L = len(spec.meta), spec.n_layers
        src_t = tensors[:n_src]
        pos = n_src
        resid = None
        if spec.has_resid:
            resid = tensors[pos]
            pos += 1
        W = tensors[pos:pos + L]
        b = tensors[pos + L:pos + 2 * L]
        gamma = tensors[pos + 2 * L] if spec.has_ln else None
        dev = dy.device
        M = spec.n_rows
        dy = _dense(dy)
        needs = ctx.needs_input_grad          # index 0 is `spec`

        # ---- input blocks: dense ones are concatenated (X_d); a block gathered through an index from a tensor with fewer rows
        # than the launch is HOISTED — W1_j t[idx] = (t W1_j^T)[idx], so its share of the first layer, of dW1 and of the input
        # gradient is computed on the tensor's rows (the adjoint of the inference path's first-layer hoisting):
        #   z1 += (tt_j W1_j^T)[idx_j],   dW1_j = segsum_idx(g)^T tt_j,   d tt_j = segsum_idx(g) W1_j
        cols, c0 = [], 0
        for m in spec.meta:
            cols.append(c0)
            c0 += m["width"]
        hoisted = [j for j, (t, m) in enumerate(zip(src_t, spec.meta))
                   if m["index"] is not None and int(t.size(0)) < M and M >= HOIST_MIN_ROWS]
        dense = [j for j in range(n_src) if j not in hoisted]
        kd = sum(spec.meta[j]["width"] for j in dense)
        W1 = W[0]
        N1 = int(W1.size(0))
        X = _buf(M, kd, dev) if kd else None
        with _phase("recompute: gather"):
            d0 = 0
            for j in dense:
                t, m = src_t[j], spec.meta[j]
                if m["segments"] is not None:     # aggregation on load: the block is the segment mean of the source's rows
                    blk = ops.segment_reduce(t, m["segments"], m["seg_mean"], src_act=m["pre_act"])
                    train_gather(blk, X, d0, 0, m["width"], None, _lib.ACT_NONE, m["negate"], M)
                else:
                    train_gather(t, X, d0, m["col0"], m["width"], m["index"], m["pre_act"], m["negate"], M)
                d0 += m["width"]
            tt = {}
            for j in hoisted:
                t, m = src_t[j], spec.meta[j]
                tt[j] = _buf(int(t.size(0)), m["width"], dev)
                train_gather(t, tt[j], 0, m["col0"], m["width"], None, m["pre_act"], m["negate"], int(t.size(0)))
        if not hoisted:
            W_d = W1
        elif kd:
            W_d = torch.cat([W1[:, cols[j]:cols[j] + spec.meta[j]["width"]] for j in dense], 1)
        saves, ctx.saves = ctx.saves, None            # (released with this call, not with the whole graph)
        # acts[l] = input rows of layer l+1 (acts[0] stands for the virtual concatenation and is never formed)
        acts: List[Optional[Tensor]] = [None]
        z_last = None
        if saves is not None:         # kept by the forward launch
            for l in range(L - 1):
                acts.append(saves[l][:, :int(W[l].size(0))])
            if spec.has_ln:
                z_last = saves[L - 1][:, :int(W[L - 1].size(0))]
        else:
            with _phase("recompute: GEMM"):
                z1 = linear(X, W_d, b[0]) if kd else b[0].expand(M, N1).contiguous()
                prods = {j: linear(tt[j], W1[:, cols[j]:cols[j] + spec.meta[j]["width"]].contiguous(), None) for j in hoisted}
            with _phase("recompute: gather"):
                for j in hoisted:
                    train_gather(prods[j], z1, 0, 0, N1, spec.meta[j]["index"], _lib.ACT_NONE, False, M, accumulate=True)
            del prods
            with _phase("recompute: SELU"):
                ops.activation_(z1, _lib.ACT_SELU)
            acts.append(z1)
            for l in range(1, L):
                last = l == L - 1
                if not last or spec.has_ln:
                    with _phase("recompute: GEMM"):           # hidden layers: bias + SELU in the launch's epilogue
                        out = linear(acts[-1], W[l], b[l], _lib.ACT_NONE if last else _lib.ACT_SELU)
                    if last:
                        z_last = out
                    else:
                        acts.append(out)
        # ---- output side: activation, residual, LayerNorm
        g = dy
        d_resid = None
        if spec.has_resid and needs[1 + n_src]:
            d_resid = torch.zeros_like(resid)
            d_resid[:, spec.resid_col0:spec.resid_col0 + dy.size(1)] = dy
        if spec.act != _lib.ACT_NONE:
            out_act = y if resid is None else y - resid[:, spec.resid_col0:spec.resid_col0 + y.size(1)]
            with _phase("activation adjoint"):
                g = act_grad(g, _dense(out_act), spec.act, False)
        d_gamma = d_beta = None
        if spec.has_ln:
            with _phase("LayerNorm adjoint"):
                g, d_gamma, d_beta = layernorm_grad(z_last, gamma, g, spec.eps)
        # ---- layers, last to second
        dW: List[Optional[Tensor]] = [None] * L
        db: List[Optional[Tensor]] = [None] * L
        need_dx_dense = any(needs[1 + j] for j in dense)
        gX = None
        chain = (FUSED_CHAIN and ops.mlp_precision() in ("bf16x6", "f16x3") and M >= FUSED_LINEAR_MIN_ROWS and kd == 128 and need_dx_dense
                 and int(g.size(1)) == 128 and all(tuple(W[l].shape) == (128, 128) for l in range(1, L)) and N1 == 128
                 and all(a is not None and a.stride(1) == 1 for a in acts[1:]))
        if chain:
            with _phase("dX chain (one launch)"):
                D, gX = backward_chain(g if g.is_contiguous() else g.contiguous(), W, acts, W_d)
            D[L] = g
            with _phase("dW + db"):
                for l in range(L - 1, 0, -1):
                    dW[l], db[l] = weight_bias_grad(D[l + 1], acts[l])
            g = D[1]
        else:
            for l in range(L - 1, 0, -1):
                with _phase("dW + db"):
                    dW[l], db[l] = weight_bias_grad(g, acts[l])
                with _phase("dX GEMM"):
                    g = linear(g, W[l].t().contiguous(), None)
                with _phase("activation adjoint"):
                    g = act_grad(g, acts[l], _lib.ACT_SELU, False, out=g)
        # ---- first layer
        dW1 = torch.empty_like(W1)
        d_src: List[Optional[Tensor]] = [None] * n_src

        def finish(j: int, gt: Tensor) -> Tensor:
            """Gradient rows of block j's tensor -> sign, activation-on-load slope, column window."""
            t, m = src_t[j], spec.meta[j]
            w = m["width"]
            if m["negate"]:
                gt = -gt
            if m["pre_act"] != _lib.ACT_NONE:
                gt = act_grad(_dense(gt), t[:, m["col0"]:m["col0"] + w], m["pre_act"], True)
            if m["col0"] == 0 and w == int(t.size(1)):
                return gt
            full = torch.zeros_like(t)
            full[:, m["col0"]:m["col0"] + w] = gt
            return full

        if kd:
            with _phase("dW + db"):
                dWd, db[0] = weight_bias_grad(g, X)
                if hoisted:
                    d0 = 0
                    for j in dense:
                        w = spec.meta[j]["width"]
                        dW1[:, cols[j]:cols[j] + w] = dWd[:, d0:d0 + w]
                        d0 += w
                else:
                    dW1 = dWd
            if need_dx_dense:
                if gX is None:
                    with _phase("dX GEMM"):
                        gX = linear(g, W_d.t().contiguous(), None)
                with _phase("input adjoint: gather / aggregation"):
                    d0 = 0
                    for j in dense:
                        t, m = src_t[j], spec.meta[j]
                        w = m["width"]
                        if needs[1 + j]:
                            gx = gX[:, d0:d0 + w]
                            if m["segments"] is not None:
                                gt = segment_broadcast(gx, m["segments"], m["seg_mean"], int(t.size(0)))
                            elif m["index"] is not None:        # (an index into a tensor with at least as many rows: not hoisted)
                                gt = ops.segment_reduce(gx, plan.gather_csr(m["index"], int(t.size(0))), False)
                            else:
                                gt = gx
                            d_src[j] = finish(j, gt)
                        d0 += w
        if db[0] is None:
            with _phase("dW + db"):
                db[0] = colsum(g)
        for j in hoisted:
            t, m = src_t[j], spec.meta[j]
            w = m["width"]
            with _phase("input adjoint: gather / aggregation"):
                G = ops.segment_reduce(g, plan.gather_csr(m["index"], int(t.size(0))), False)       # [rows(t), N1]
            with _phase("dW + db"):
                dW1[:, cols[j]:cols[j] + w] = weight_bias_grad(G, tt[j], False)[0]
            if needs[1 + j]:
                with _phase("dX GEMM"):
                    gt = linear(G, W1[:, cols[j]:cols[j] + w].t().contiguous(), None)
                with _phase("input adjoint: gather / aggregation"):
                    # (sign and slope were applied when tt was formed from t: chain rule through them)
                    d_src[j] = finish(j, gt)
        dW[0] = dW1
        db = [g_ if b_ is not None else None for g_, b_ in zip(db, b)]          # (a layer without a bias: no gradient slot to fill)
        grads = [None] + d_src + ([d_resid] if spec.has_resid else []) + dW + db
        if spec.has_ln:
            grads += [d_gamma, d_beta]
        return tuple(grads)


def wants_grad(packed, sources: Sequence[Source], resid: Optional[Tensor]) -> bool:
    if not torch.is_grad_enabled():
        return False
    if any(s.tensor.requires_grad for s in sources) or (resid is not None and resid.requires_grad):
        return True
    params = getattr(packed, "params", None)
    if params is None:
        return False
    weights, biases, ln = params
    flat = list(weights) + list(biases) + ([ln[0], ln[1]] if ln is not None else [])
    return any(p is not None and p.requires_grad for p in flat)


def mlp(packed, sources: Sequence[Source], n_rows: int, act: int, resid: Optional[Tensor], resid_col0: int) -> Tensor:
    """Differentiable form of ops.mlp_forward (called by it when gradients are wanted)."""
    if any(s.additive for s in sources):
        raise NotImplementedError("pre-multiplied (additive) input blocks are an inference-only optimisation")
    if packed.params is None or packed.heads_params:
        raise NotImplementedError("this packed MLP carries no parameter references / has heads: inference only")
    weights, biases, ln = packed.params
    spec = _Spec()
    spec.packed, spec.n_rows, spec.act = packed, int(n_rows), int(act)
    spec.resid_col0, spec.has_resid = int(resid_col0), resid is not None
    spec.n_layers, spec.has_ln, spec.eps = len(weights), ln is not None, (float(ln[2]) if ln is not None else 0.0)
    spec.meta = [dict(index=s.index, col0=s.col0, width=s.width, negate=s.negate, pre_act=s.pre_act, segments=s.segments,
                      seg_mean=s.seg_mean) for s in sources]
    tensors = [s.tensor for s in sources] + ([resid] if resid is not None else []) + list(weights) + list(biases)
    if ln is not None:
        tensors += [ln[0], ln[1]]
    return _FusedMLP.apply(spec, *tensors)


# ------------------------------------------------------------------------------------- segmented reduction
class _SegmentReduce(torch.autograd.Function):
    """out[s] = act(sum | mean of src_act(src[perm[p]]) over segment s) — scatter(…, reduce) of nn/blocks.py:183,231 and the
    feature part of pool_edge (:67)."""

    @staticmethod
    def forward(ctx, src: Tensor, csr, mean: bool, act: int, src_act: int):
        out = ops.segment_reduce(src, csr, mean, act, src_act=src_act)
        ctx.csr, ctx.mean, ctx.act, ctx.src_act = csr, mean, act, src_act
        ctx.save_for_backward(src, out)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        src, out = ctx.saved_tensors
        g = act_grad(_dense(dout), out, ctx.act, False)
        gs = segment_broadcast(_dense(g), ctx.csr, ctx.mean, int(src.size(0)))
        if ctx.src_act != _lib.ACT_NONE:
            gs = act_grad(gs, _dense(src), ctx.src_act, True, out=gs)
        return gs, None, None, None, None


def segment_reduce(src: Tensor, csr, mean: bool, act: int, src_act: int) -> Tensor:
    return _SegmentReduce.apply(src, csr, mean, act, src_act)


# ------------------------------------------------------------------------------------- gathers / interpolation / projections
# (gMuS-GNN and REMuS-GNN: nn/mugs_gnn.py restriction + knn_interpolate, nn/blocks.py:34-48,88-114,408-456).  Forward: the
# inference kernels.  The adjoints are linear maps with static coefficients: an elementwise product (torch) followed, where rows
# were gathered, by the deterministic segmented sum on the CSR plan of the gather index (g4c_segment_reduce).
class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, idx32: Tensor):
        out = _buf(int(idx32.numel()), int(x.size(1)), x.device)
        ops.copy_cols(x, out, 0, idx32=idx32)
        ctx.idx32, ctx.n = idx32, int(x.size(0))
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        return ops.segment_reduce(_dense(dout), plan.gather_csr(ctx.idx32, ctx.n), False), None


def gather_rows(x: Tensor, idx32: Tensor) -> Tensor:
    return _GatherRows.apply(x, idx32)


def _segment_coefficients(w: Tensor, csr) -> Tensor:
    """w_p / sum of w over p's segment, [n, 1] (static per interpolation plan; cached on the plan)."""
    hit = getattr(csr, "_coef", None)
    if hit is None or hit[0] != (w.data_ptr(), w._version):
        deg = (csr.off[1:] - csr.off[:-1]).long()
        seg = torch.repeat_interleave(torch.arange(csr.n_seg, device=w.device), deg)
        wf = w.reshape(-1).float()
        tot = torch.zeros(csr.n_seg, dtype=torch.float32, device=w.device).index_add_(0, seg, wf)
        csr._coef = ((w.data_ptr(), w._version), (wf / tot[seg]).reshape(-1, 1).contiguous())
        hit = csr._coef
    return hit[1]


class _WeightedSegmentMean(torch.autograd.Function):
    """knn_interpolate (nn/blocks.py:34-48): out[s] = sum_p w_p x[x_idx[p]] / sum_p w_p over segment s; with `out_idx32` the
    result rows land at those positions of a zero [n_out, F] tensor (UpEdgeMP's masked write, nn/blocks.py:437-441)."""

    @staticmethod
    def forward(ctx, x: Tensor, x_idx32: Tensor, w: Tensor, csr, n_out: int, out_idx32: Optional[Tensor]):
        width = int(x.size(1))
        out = (torch.zeros if out_idx32 is not None else torch.empty)((n_out, width), dtype=torch.float32, device=x.device)
        ops.weighted_segment_mean(x, x_idx32, w, csr, out, out_idx32)
        ctx.args = (x_idx32, w, csr, out_idx32, int(x.size(0)))
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        x_idx32, w, csr, out_idx32, n_x = ctx.args
        d = _dense(dout)
        if out_idx32 is not None:
            rows = _buf(csr.n_seg, int(d.size(1)), d.device)
            ops.copy_cols(d, rows, 0, idx32=out_idx32)
            d = rows
        tmp = segment_broadcast(d, csr, False, csr.n)
        tmp.mul_(_segment_coefficients(w, csr))
        return ops.segment_reduce(tmp, plan.gather_csr(x_idx32, n_x), False), None, None, None, None, None


def weighted_segment_mean(x: Tensor, x_idx32: Tensor, w: Tensor, csr, n_out: Optional[int] = None,
                          out_idx32: Optional[Tensor] = None) -> Tensor:
    return _WeightedSegmentMean.apply(x, x_idx32, w, csr, csr.n_seg if n_out is None else int(n_out), out_idx32)


class _ProjectToEdges(torch.autograd.Function):
    """(v[node].view(E, F, 2) * unit[:, None]).sum(-1) (nn/blocks.py:449-451)."""

    @staticmethod
    def forward(ctx, v: Tensor, node32: Optional[Tensor], unit: Tensor, n_edges: int, n_feat: int):
        ctx.args = (node32, unit, n_edges, n_feat, int(v.size(0)))
        return ops.project_to_edges(v, node32, unit, n_edges, n_feat)

    @staticmethod
    def backward(ctx, dout: Tensor):
        node32, unit, n_edges, n_feat, n_v = ctx.args
        tmp = (dout.unsqueeze(2) * unit.unsqueeze(1)).reshape(n_edges, 2 * n_feat)
        dv = tmp if node32 is None else ops.segment_reduce(tmp, plan.gather_csr(node32, n_v), False)
        return dv, None, None, None, None


def project_to_edges(v: Tensor, node32: Optional[Tensor], unit: Tensor, n_edges: int, n_feat: int) -> Tensor:
    return _ProjectToEdges.apply(v, node32, unit, n_edges, n_feat)


class _EdgeScalarToNodeVector(torch.autograd.Function):
    """Uinv [n, 2, k] @ e.view(n, k, F), feature-major flatten (nn/blocks.py:111-114)."""

    @staticmethod
    def forward(ctx, e: Tensor, unit_inv: Tensor, n_nodes: int, k: int):
        ctx.args = (unit_inv, n_nodes, k, int(e.size(1)))
        return ops.edge_scalar_to_node_vector(e, unit_inv, n_nodes, k)

    @staticmethod
    def backward(ctx, dout: Tensor):
        unit_inv, n, k, f = ctx.args
        # d e[(n, j), f] = sum_c Uinv[n, c, j] dout[n, 2 f + c]: the projection of the node vectors dout[n] on the edges' "unit vectors"
        # Uinv[n, :, j] — g4c_project_to_edges with node = edge // k (no batched library GEMM)
        unit = unit_inv.transpose(1, 2).reshape(n * k, 2).contiguous()
        node32 = torch.arange(n, dtype=torch.int32, device=dout.device).repeat_interleave(k)
        return ops.project_to_edges(_dense(dout), node32, unit, n * k, f), None, None, None


def edge_scalar_to_node_vector(e: Tensor, unit_inv: Tensor, n_nodes: int, k: int) -> Tensor:
    return _EdgeScalarToNodeVector.apply(e, unit_inv, n_nodes, k)

"""mlp_rs1_kernel (row-split kernel, rounded-bf16 mode: weights resident in LDS, no barriers) against mlp_ws_kernel<SP = 1> — parity within
the mode's tolerance and time — in the forms BASELINE config 3's angle launches use: hoisted message form, fp32 or bf16(SELU) input rows,
bf16 product tables, bf16(SELU) output rows, fused mean over uniform segments of K rows."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
lib = _lib.load(); dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("bf16")

def bench(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(8):
        s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(reps): fn()
        t_.record(); torch.cuda.synchronize()
        ts.append(s_.elapsed_time(t_) / reps * 1e3)
    return statistics.median(ts), min(ts)

for layers in (2, 3):
    for n, K in ((3, 5), (2999, 5), (1001, 4), (777, 8), (500, 7), (500000, 5)):
        torch.manual_seed(n + layers)
        E = K * n
        blk = B.GNBlock((3 * H, (H,) * layers, True), (2 * H, (H,) * layers, True)).to(dev)
        e32, pr, pc = torch.randn(E, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
        ei = torch.stack([torch.randint(0, n, (E,)), torch.arange(n).repeat_interleave(K)]).to(dev)
        ep, csr = plan.edge_csr(ei, n)
        assert csr.uniform_deg == K
        pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
        lin = blk.edge_mlp._linears(); ln = blk.edge_mlp.MLP.layer_norm
        pk_rs = ops.PackedMLP([lin[0].weight.detach()[:, :H].contiguous()] + [l.weight for l in lin[1:]], [l.bias for l in lin],
                              (ln.weight, ln.bias, ln.eps), [H], [False], precision="bf16", rs_order=True)
        e16 = torch.nn.functional.selu(e32).to(torch.bfloat16)
        pr16, pc16 = pr.to(torch.bfloat16), pc.to(torch.bfloat16)
        order = ops._rs_k_order(dev)
        rs = lambda t: ops.RsOrderedRows.tag(t[:, order].contiguous())          # the same rows in the row-split kernel's column order
        def for_rs(src):
            return [ops.Source(rs(s_.tensor) if s_.tensor.dtype == torch.bfloat16 else s_.tensor, index=s_.index, pre_act=s_.pre_act, additive=s_.additive) for s_ in src]
        forms = {
            "fp32 rows, fp32 tables, fp32 out": ([ops.Source(e32, pre_act=_lib.ACT_SELU), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)], {}),
            "fp32 rows, bf16 tables, bf16(SELU) out": ([ops.Source(e32, pre_act=_lib.ACT_SELU), ops.Source(pr16, index=ep.row, additive=True), ops.Source(pc16, index=ep.col, additive=True)],
                                                        dict(rows_dtype=torch.bfloat16, rows_act=_lib.ACT_SELU)),
            "bf16 rows, bf16 tables, bf16(SELU) out": ([ops.Source(e16), ops.Source(pr16, index=ep.row, additive=True), ops.Source(pc16, index=ep.col, additive=True)],
                                                        dict(rows_dtype=torch.bfloat16, rows_act=_lib.ACT_SELU)),
        }
        for name, (src, kw) in forms.items():
            for with_agg in (False, True):
                if not with_agg and kw:
                    continue        # (compact rows only exist with a fused aggregation)
                res = {}
                for tag, pack in (("ws", pk), ("rs", pk_rs)):
                    agg = torch.full((n, H), float("nan"), device=dev)
                    y = ops.mlp_forward(pack, for_rs(src) if tag == "rs" else src, E, agg=(csr, agg, True) if with_agg else None, **kw)
                    if tag == "rs" and y.dtype == torch.bfloat16:
                        y = ops.rs_rows_to_natural(ops.RsOrderedRows.tag(y))
                    res[tag] = (y.float(), agg, int(lib.g4c_mlp_last_kernel()))
                d = (res["rs"][0] - res["ws"][0]).abs()
                line = f"layers {layers} rows {E:8d} K {K}  {name:40s} agg {int(with_agg)}: kernels {res['ws'][2]}/{res['rs'][2]}  rows max {d.max().item():.2e} mean {d.mean().item():.2e}"
                if with_agg:
                    da = (res["rs"][1] - res["ws"][1]).abs()
                    # the aggregate of the rs kernel's OWN fp32 rows, where they are available
                    line += f"  agg max {da.max().item():.2e} mean {da.mean().item():.2e} finite {bool(torch.isfinite(res['rs'][1]).all())}"
                    if not kw:
                        own = ops.segment_reduce(res["rs"][0], csr, True)
                        line += f"  |agg - reduce(own rows)| {(own - res['rs'][1]).abs().max().item():.2e}"
                print(line)
                if n == 500000 and with_agg:
                    for tag, pack in (("mlp_ws_kernel<SP=1>", pk), ("mlp_rs1_kernel", pk_rs)):
                        agg = torch.empty((n, H), device=dev)
                        srcs = for_rs(src) if pack is pk_rs else src
                        med, mn = bench(lambda: ops.mlp_forward(pack, srcs, E, agg=(csr, agg, True), **kw))
                        print(f"      {tag:22s} kernel {int(lib.g4c_mlp_last_kernel())}: median {med:7.1f} us  min {mn:7.1f}")

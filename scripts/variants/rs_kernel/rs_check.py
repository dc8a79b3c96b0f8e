"""mlp_rs_kernel (row-split persistent kernel, mlp_rs.hip) against the tile kernel on the hoisted three-layer message form; timing
(20 launches back to back) against mlp_ws_kernel.  Usage: python scripts/rs_check.py [--time]"""
import argparse, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
ap = argparse.ArgumentParser(); ap.add_argument("--time", action="store_true"); ap.add_argument("--agg", action="store_true")
a = ap.parse_args()
torch.set_grad_enabled(False)
lib = _lib.load(); dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("f16x3")
bad = 0
for rows6 in (1, 7, 100, 2999, 20011, 100000):
    torch.manual_seed(rows6)
    n = rows6; E = 6 * n
    blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
    e, pr, pc = torch.randn(E, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    col = torch.arange(n).repeat_interleave(6)
    ei = torch.stack([torch.randint(0, n, (E,)), col]).to(dev)
    ep, csr = plan.edge_csr(ei, n)
    pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    lin = blk.edge_mlp._linears(); ln = blk.edge_mlp.MLP.layer_norm
    pk_rs = ops.PackedMLP([lin[0].weight.detach()[:, :H].contiguous(), lin[1].weight, lin[2].weight], [l.bias for l in lin],
                          (ln.weight, ln.bias, ln.eps), [H], [False], precision="f16x3", rs_order=True)
    for pre in (_lib.ACT_SELU, _lib.ACT_NONE):
        src = [ops.Source(e, pre_act=pre), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
        lib.g4c_mlp_ws_enable(0)
        ref = ops.mlp_forward(pk, src, E)
        got = ops.mlp_forward(pk_rs, src, E)
        k = int(lib.g4c_mlp_last_kernel())
        d = (got - ref).abs().max().item()
        ok = k == 5 and d < 2e-5 and bool(torch.isfinite(got).all())
        bad += not ok
        print(f"{'ok  ' if ok else 'FAIL'} rows {E:7d} pre_act {pre}: kernel {k}  max|rs - tile| {d:.2e}")
    lib.g4c_mlp_ws_enable(1)
print("all checks passed" if not bad else f"{bad} FAILED")
if a.time:
    rows = 600000; n = rows // 6
    torch.manual_seed(0)
    blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
    e, pr, pc = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    ei = torch.stack([torch.randint(0, n, (rows,)), torch.arange(n).repeat_interleave(6)]).to(dev)
    ep, csr = plan.edge_csr(ei, n)
    lin = blk.edge_mlp._linears(); ln = blk.edge_mlp.MLP.layer_norm
    pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
    pk_rs = ops.PackedMLP([lin[0].weight.detach()[:, :H].contiguous(), lin[1].weight, lin[2].weight], [l.bias for l in lin],
                          (ln.weight, ln.bias, ln.eps), [H], [False], precision="f16x3", rs_order=True)
    src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
    out = torch.empty(rows, H, device=dev)
    for name, fn in (("mlp_ws_kernel (no aggregation)", lambda: ops.mlp_forward(pk, src, rows, out=out)),
                     ("mlp_rs_kernel (no aggregation)", lambda: ops.mlp_forward(pk_rs, src, rows, out=out))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for r in range(8):
            s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(10): fn()
            t_.record(); torch.cuda.synchronize()
            ts.append(s_.elapsed_time(t_) / 10 * 1e3)
        print(f"{name:36s} kernel {int(lib.g4c_mlp_last_kernel())}  rows {rows}: median {statistics.median(ts):7.1f} us  min {min(ts):7.1f}")

"""mlp_rs2_kernel (the row-split UPDATE kernel of the rounded-bf16 mode) against the tile kernel on the same launch — parity within the
mode's rounding, and time: [bf16 aggregate | bf16 e] -> 256 -> 128 -> 128 -> LayerNorm -> SELU -> e' (+ two product heads)."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
lib = _lib.load(); dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("bf16")
order = ops._rs_k_order(dev)
rs = lambda t: ops.RsOrderedRows.tag(t[:, order].contiguous())
nat = lambda t: ops.rs_rows_to_natural(t) if isinstance(t, ops.RsOrderedRows) else t

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
    return statistics.median(ts)

for n in (20001, 33333, 100000, 500000):
    torch.manual_seed(n)
    blk = B.GNBlock((3 * H, (H, H), True), (2 * H, (H, H), True)).to(dev)
    nxt = B.GNBlock((3 * H, (H, H), True), (2 * H, (H, H), True)).to(dev)
    agg = torch.randn(n, H, device=dev).to(torch.bfloat16)
    e = torch.nn.functional.selu(torch.randn(n, H, device=dev)).to(torch.bfloat16)
    for e_tagged in (True, False):
        for heads in (True, False):
            for out16 in (True, False):
                res = {}
                for on in (False, True):
                    B.UPDATE_ROW_SPLIT = on
                    src = [ops.Source(rs(agg)), ops.Source(rs(e) if e_tagged else e)]
                    out = torch.empty(n, H, device=dev, dtype=torch.bfloat16 if out16 else torch.float32)
                    if heads:
                        f = lambda: blk.node_mlp.run_with_heads(src, n, _lib.ACT_SELU, nxt.edge_mlp, H, [H, H], out=out, rs_rows=True)
                        y, hs = f()
                    else:
                        f = lambda: blk.node_mlp.run_coded(src, n, _lib.ACT_SELU, out=out)
                        y, hs = f(), []
                    k = int(lib.g4c_mlp_last_kernel())
                    t = bench(f) if n == 500000 else 0.0
                    res[on] = (nat(y).float().clone(), [nat(h).float().clone() for h in hs], k, t)
                (y0, h0, k0, t0), (y1, h1, k1, t1) = res[False], res[True]
                d = (y1 - y0).abs()
                line = (f"rows {n:7d} e {'rs ' if e_tagged else 'nat'} heads {int(heads)} out {'bf16' if out16 else 'fp32'}: kernels {k0}/{k1}  "
                        f"e' max {d.max().item():.2e} mean {d.mean().item():.2e}")
                for a, b in zip(h0, h1):
                    dh = (b - a).abs()
                    line += f"  head max {dh.max().item():.2e} mean {dh.mean().item():.2e}"
                if n == 500000:
                    line += f"   tile {t0:6.1f} us  rs2 {t1:6.1f} us"
                print(line)

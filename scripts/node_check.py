"""mlp_node_kernel (the node update's persistent pair-pipelined launch, mlp_ws.hip) against the 32-row tile kernel: v' and the heads
over row counts from 1 to 100 000 (remainders of every kind: < 32, 33 - 64 rows, odd tiles per workgroup), two / three layers,
0 / 2 heads, SELU / tanh / no activation, with / without LayerNorm; then the level-1 node launch of the headline workload timed on
both kernels (same process, interleaved).  Usage: python scripts/node_check.py [--time]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B
ap = argparse.ArgumentParser(); ap.add_argument("--time", action="store_true")
a = ap.parse_args()
torch.set_grad_enabled(False)
lib = _lib.load()
dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("f16x3")
bad = []


def both(fn):
    lib.g4c_mlp_node_enable(0); ref = fn(); k0 = int(lib.g4c_mlp_last_kernel())
    lib.g4c_mlp_node_enable(2)
    try:
        got = fn(); k1 = int(lib.g4c_mlp_last_kernel())
    finally:
        lib.g4c_mlp_node_enable(0)
    assert k0 == 2 and k1 == 5, (k0, k1)
    return ref, got


def cmp(name, ref, got, tol=2e-5):
    d = (ref - got).abs().max().item() if ref.numel() else 0.0
    ok = d <= tol and bool(torch.isfinite(got).all())
    print(f"{'ok  ' if ok else 'FAIL'} {name:70s} max|node - tile| = {d:.2e} (tol {tol:g})")
    if not ok: bad.append(name)


for layers in (3, 2):
    for ln in (True, False):
        torch.manual_seed(layers * 2 + ln)
        hid = (H,) * layers
        node = B.MLP(2 * H, hid, ln).to(dev)
        nxt = B.MLP(3 * H, hid, True).to(dev)
        for rows in (100000, 12511, 4097, 97, 65, 64, 33, 32, 31, 1):
            agg, v = torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev)
            for act in (_lib.ACT_SELU, _lib.ACT_TANH, _lib.ACT_NONE):
                tag = f"[{layers} layers, LayerNorm {ln}] rows={rows} act={act}: "
                r, g_ = both(lambda: node.run_with_heads([ops.Source(agg), ops.Source(v)], rows, act, nxt, H, [H, H]))
                cmp(tag + "v' (with heads)", r[0], g_[0])
                for j in range(2): cmp(tag + f"head {j}", r[1][j], g_[1][j])
                r, g_ = both(lambda: node.run_coded([ops.Source(agg), ops.Source(v)], rows, act))
                cmp(tag + "v' (no heads)", r, g_)
print("all node-kernel checks passed" if not bad else "FAILED: " + ", ".join(bad))
if a.time:
    torch.manual_seed(0)
    node = B.MLP(2 * H, (H, H, H), True).to(dev); nxt = B.MLP(3 * H, (H, H, H), True).to(dev)
    for rows in (100000, 50000, 25000, 12500):
        agg, v = torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev)
        fn = lambda: node.run_with_heads([ops.Source(agg), ops.Source(v)], rows, _lib.ACT_SELU, nxt, H, [H, H])
        times = {0: [], 2: []}
        for k in times:
            lib.g4c_mlp_node_enable(k); fn(); fn()
        torch.cuda.synchronize()
        for r in range(15):
            for k in times:
                lib.g4c_mlp_node_enable(k)
                s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(); fn(); fn(); fn(); t_.record(); torch.cuda.synchronize()
                times[k].append(s_.elapsed_time(t_) / 3 * 1e3)
        lib.g4c_mlp_node_enable(0)
        print(f"node launch, {rows} rows, 3 layers + 2 heads: tile kernel median {statistics.median(times[0]):7.1f} us (min {min(times[0]):7.1f})   "
              f"node kernel median {statistics.median(times[2]):7.1f} us (min {min(times[2]):7.1f})")

"""Standalone timing of the two dominant launches at the C4 level-1 shape:
edge MLP (gather e|v[row]|v[col] -> 384-128-128-128 + LN) over E = k*N rows and the CSR mean.
Usage: python scripts/bench_kernels.py [--nodes N] [--reps R] [--what mlp|reduce|node|all]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import ops, plan, synthetic as S, _lib
from graphs4cfd_amd.nn import blocks as B

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=100_000)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--what", default="all")
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = S.mus_graph(a.nodes, levels=1, seed=0).to(dev)
n, E, H = a.nodes, g.edge_index.size(1), 128
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
v, e = torch.randn(n, H, device=dev), torch.randn(E, H, device=dev)
ep, csr = plan.edge_csr(g.edge_index, n)


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a_.record()
    for _ in range(reps):
        fn()
    b_.record(); torch.cuda.synchronize()
    return a_.elapsed_time(b_) * 1e-3 / reps


if a.what in ("mlp", "all"):
    out = torch.empty(E, H, device=dev)
    srcs = [ops.Source(e), ops.Source(v, ep.row), ops.Source(v, ep.col)]
    t = timeit(lambda: blk.edge_mlp.run_hoisted([ops.Source(e)], [(v, ep.row), (v, ep.col)], E, 0, out=out), a.reps)
    fl = E * 2.0 * (128 * 128 + 128 * 128 + 128 * 128) + 2 * n * 2.0 * 128 * 128   # executed FLOP (first layer hoisted)
    print(f"edge MLP  E={E}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TFLOP/s  ({fl / t / 157.3e12 * 100:.1f}% of fp32 MFMA peak)")
if a.what in ("node", "all"):
    agg = torch.randn(n, H, device=dev)
    out = torch.empty(n, H, device=dev)
    srcs = [ops.Source(agg), ops.Source(v)]
    t = timeit(lambda: blk.node_mlp.run_coded(srcs, n, 1, out=out), a.reps)
    fl = n * 2.0 * (256 * 128 + 128 * 128 + 128 * 128)
    print(f"node MLP  N={n}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TFLOP/s  ({fl / t / 157.3e12 * 100:.1f}% of fp32 MFMA peak)")
if a.what in ("reduce", "all"):
    out = torch.empty(n, H, device=dev)
    t = timeit(lambda: ops.segment_reduce(e, csr, True, out=out), a.reps)
    by = 4.0 * (E * H + n * H + n + 1)
    print(f"seg mean  E={E}: {t * 1e6:8.1f} us  {by / t / 1e9:6.0f} GB/s  ({by / t / 8e12 * 100:.1f}% of 8 TB/s)")

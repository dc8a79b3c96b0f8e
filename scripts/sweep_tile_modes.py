"""Tuning: time of the hoisted edge MLP, the node MLP and the one-layer node-product launch vs row count for each
kernel variant behind g4c_mlp_forward_rows (ops.mlp_forward(..., tile_mode=...)), interleaved in one process.
Usage: python scripts/sweep_tile_modes.py [rows,rows,...] [--modes 324,325]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import ops
from graphs4cfd_amd.nn import blocks as B

ap = argparse.ArgumentParser()
ap.add_argument("rows", nargs="?", default="192,768,3072,6144,12500,25000,50000,75000,150000,600000")
ap.add_argument("--modes", default="64,32,322,324,325")
a = ap.parse_args()
modes = [int(m) for m in a.modes.split(",")]
dev = torch.device("cuda", 0); H = 128
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
pk_e = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
pk_v = blk.node_mlp.packed([H, H], [False, False])
pk_p = blk.edge_mlp._packed_cols("hoist1", H, 2 * H, [H], [False], True)
pk_full = blk.edge_mlp.packed([H, H, H], [False, False, False])


def timeit(f, reps):
    for _ in range(2): f()
    ts = []
    for _ in range(5):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); [f() for _ in range(reps)]; t.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(t) / reps * 1e3)
    return statistics.median(ts)


print("case      rows " + " ".join(f"{m:>9d}" for m in modes) + "   (us per launch, median of 5 x reps)")
for rows in [int(x) for x in a.rows.split(",")]:
    n = max(rows // 6, 32)
    v, e, agg = torch.randn(n, H, device=dev), torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev)
    row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
    col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
    pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    out_e, out_v = torch.empty(rows, H, device=dev), torch.empty(rows, H, device=dev)
    src_e = [ops.Source(e), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
    vv, aa = torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev)
    cases = {"edge": lambda m: ops.mlp_forward(pk_e, src_e, rows, 0, out=out_e, tile_mode=m),
             "node": lambda m: ops.mlp_forward(pk_v, [ops.Source(aa), ops.Source(vv)], rows, 1, out=out_v, tile_mode=m),
             "edge3src": lambda m: ops.mlp_forward(pk_full, [ops.Source(e), ops.Source(v, index=row), ops.Source(v, index=col)], rows, 0, out=out_e, tile_mode=m),
             "prod(n)": lambda m: ops.mlp_forward(pk_p, [ops.Source(v)], n, 0, out=out_v, tile_mode=m),
             "product": lambda m: ops.mlp_forward(pk_p, [ops.Source(vv)], rows, 0, out=out_v, tile_mode=m)}
    reps = 20 if rows <= 100000 else 5
    for name, f in cases.items():
        print(f"{name:8s} {rows:7d} " + " ".join(f"{timeit(lambda: f(m), reps):9.1f}" for m in modes), flush=True)

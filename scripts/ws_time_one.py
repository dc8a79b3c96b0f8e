"""One library build (G4C_LIB_PATH), the level-1 message launch of the headline workload (600k rows, first layer hoisted, fused
aggregation) on mlp_ws_kernel: correctness against the same library's tile kernel (e' to 2e-5, aggregate bit-exact against
g4c_segment_reduce of the kernel's own rows) and the median / minimum launch time over 40 launches.  One line of output; the driver
(scripts/ws_ab_variants.sh) interleaves the builds.  Usage: G4C_LIB_PATH=... python scripts/ws_time_one.py [--rows N] [--layers 3]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=600000); ap.add_argument("--layers", type=int, default=3)
ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
torch.set_grad_enabled(False)
lib = _lib.load()
dev = torch.device("cuda", 0); H = 128
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H,) * a.layers, True), (2 * H, (H,) * a.layers, True)).to(dev)
rows = a.rows; n = rows // 6
e, v = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev)
colh = torch.arange(n).repeat_interleave(6)
ei = torch.stack([torch.randint(0, n, (rows,)), colh]).to(dev)
ep, csr = plan.edge_csr(ei, n)
pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
out, agg = torch.empty(rows, H, device=dev), torch.empty(n, H, device=dev)
fn = lambda: ops.mlp_forward(pk, src, rows, 0, out=out, agg=(csr, agg, True))
lib.g4c_mlp_bx6i_enable(0)
lib.g4c_mlp_ws_enable(0); fn(); ref = out.clone()
lib.g4c_mlp_ws_enable(2); fn()
d = (out - ref).abs().max().item()
sr = ops.segment_reduce(out, csr, True)
bit = bool(torch.equal(sr, agg))
rel = ((sr - agg).abs().max() / sr.abs().max()).item()
for _ in range(5): fn()
torch.cuda.synchronize()
# 20 launches back to back between two events (the launch overhead of an eager launch — ~10 us — amortises: the figure is the kernel's)
ts = []
for _ in range(a.reps // 4):
    s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(20): fn()
    t_.record(); torch.cuda.synchronize()
    ts.append(s_.elapsed_time(t_) * 1e3 / 20)
print(f"{os.path.basename(os.environ.get('G4C_LIB_PATH', 'shipped')):28s} rows {rows} median {statistics.median(ts):7.1f} us  min {min(ts):7.1f} us   "
      f"max|ws - tile| {d:.2e}  aggregate bit-exact {bit} (max rel {rel:.1e})")

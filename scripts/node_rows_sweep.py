"""Level-1 node launch (3 layers + LayerNorm + SELU + 2 heads, tile kernel) timed over row counts around the workgroup-round
boundaries of a 256-CU chip (4 workgroups per CU = 1024 slots of one 32-row tile): does the launch pay for whole rounds?"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
lib = _lib.load(); dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("f16x3")
torch.manual_seed(0)
node = B.MLP(2 * H, (H, H, H), True).to(dev); nxt = B.MLP(3 * H, (H, H, H), True).to(dev)
for rows in (32768, 65536, 90112, 98304, 100000, 102400, 114688, 131072, 163840):
    agg, v = torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev)
    fn = lambda: node.run_with_heads([ops.Source(agg), ops.Source(v)], rows, _lib.ACT_SELU, nxt, H, [H, H])
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(8):
        s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(10): fn()
        t_.record(); torch.cuda.synchronize()
        ts.append(s_.elapsed_time(t_) / 10 * 1e3)
    print(f"rows {rows:7d}  tiles {rows // 32:5d} = {rows / 32 / 1024:5.2f} rounds of 1024  median {statistics.median(ts):7.1f} us  min {min(ts):7.1f}  ns/row {1e3 * statistics.median(ts) / rows:6.3f}")

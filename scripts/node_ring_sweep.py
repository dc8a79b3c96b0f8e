"""Level-1 node launch (tile kernel) with the chip-filling instantiation (weight ring of 2 steps, 4 workgroups per CU) against the deep-ring
one (ring of 8, 2 workgroups per CU: g4c_mlp_small_launch_tiles forces it at any size)."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
lib = _lib.load(); dev = torch.device("cuda", 0); H = 128
ops.set_mlp_precision("f16x3")
torch.manual_seed(0)
node = B.MLP(2 * H, (H, H, H), True).to(dev); nxt = B.MLP(3 * H, (H, H, H), True).to(dev)
for rows in (100000, 25000):
    agg, v = torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev)
    fn = lambda: node.run_with_heads([ops.Source(agg), ops.Source(v)], rows, _lib.ACT_SELU, nxt, H, [H, H])
    for lim, name in ((512, "ring 2, 4 WG/CU (default at this size)"), (1 << 30, "deep ring, 2 WG/CU (forced)")):
        old = lib.g4c_mlp_small_launch_tiles(lim)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for r in range(8):
            s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(10): fn()
            t_.record(); torch.cuda.synchronize()
            ts.append(s_.elapsed_time(t_) / 10 * 1e3)
        lib.g4c_mlp_small_launch_tiles(old)
        print(f"rows {rows:6d}  {name:42s} median {statistics.median(ts):7.1f} us  min {min(ts):7.1f}")

"""Every timed launch of one eager rollout step of a bench workload, in order (HIP events on the launch stream), plus the hipGraph step time.
Usage: python scripts/step_breakdown.py [--workload headline]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
import graphs4cfd_amd as gfd
from graphs4cfd_amd import ops, synthetic as S
from graphs4cfd_amd.nn.model import Rollout
dev = torch.device("cuda", 0)
ops.set_mlp_precision(args.precision)
g, model, nf = bench.build_workload(args, gfd, S, dev)
ro = Rollout(model, g.clone().to(dev), 40, capture=False)
ro.run(2); torch.cuda.synchronize()
with ops.KernelTimer() as kt:
    ro.run(1)
torch.cuda.synchronize()
tot = 0.0
for kind, flops, nbytes, a, b in kt.records:
    us = a.elapsed_time(b) * 1e3; tot += us
    print(f"{kind:26s} {us:8.1f} us   {flops / 1e9:8.2f} GFLOP  {nbytes / 1e6:8.1f} MB alg   {flops / max(us, 1e-3) / 1e6:7.1f} TFLOP/s  {nbytes / max(us, 1e-3) / 1e3:7.0f} GB/s")
print(f"sum of timed launches {tot / 1e3:.3f} ms over {len(kt.records)} launches")
ro.close()
cap = Rollout(model, g.clone().to(dev), 60, capture=True)
cap.run(5); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); cap.run(40); e.record(); torch.cuda.synchronize()
print(f"hipGraph step {s.elapsed_time(e) / 40:.3f} ms")

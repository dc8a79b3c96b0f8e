"""Per-launch-shape timing of the fused MLP / segment-reduce kernels for one rollout step
(HIP events around each launch).  Usage: python scripts/profile_shapes.py [--nodes N] [--model M]"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import ops, synthetic as S
from graphs4cfd_amd.nn.model import Rollout

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=100_000)
ap.add_argument("--model", default="NsThreeScaleGNN")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
levels = {"NsOneScaleGNN": 1, "NsTwoScaleGNN": 2, "NsThreeScaleGNN": 3, "NsFourScaleGNN": 4}[a.model]
g = S.mus_graph(a.nodes, levels=levels, seed=0).to(dev)
torch.manual_seed(0)
model = getattr(gfd.nn, a.model)(arch=S.mus_arch(a.model, 128), device=dev)
ro = Rollout(model, g, a.reps + 4, capture=False)
ro.run(2); torch.cuda.synchronize()
with ops.KernelTimer() as kt:
    ro.run(a.reps)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, flops, nbytes, s, e in kt.records:
    d = agg.setdefault((kind, flops, nbytes), [0, 0.0])
    d[0] += 1; d[1] += s.elapsed_time(e) * 1e-3
tot = 0.0
print(f"{'kind':16s} {'launches/step':>13s} {'GFLOP':>9s} {'MB':>9s} {'avg us':>9s} {'TFLOP/s':>8s} {'GB/s':>8s} {'ms/step':>8s}")
for (kind, flops, nbytes), (n, sec) in agg.items():
    avg = sec / n
    tot += sec / a.reps
    print(f"{kind:16s} {n / a.reps:13.1f} {flops / 1e9:9.2f} {nbytes / 1e6:9.2f} {avg * 1e6:9.1f} {flops / avg / 1e12:8.1f} {nbytes / avg / 1e9:8.0f} {1e3 * sec / a.reps:8.3f}")
print(f"sum of timed kernels per step: {tot * 1e3:.3f} ms")

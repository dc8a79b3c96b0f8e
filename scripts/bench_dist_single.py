"""Single-GPU check of the partitioned code path's overhead: DistributedRollout with world = 1 (no exchange) vs Rollout on
the same mesh.  Usage: python scripts/bench_dist_single.py [--nodes 12500]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import partition as P, synthetic as S
from graphs4cfd_amd.nn.model import Rollout
ap = argparse.ArgumentParser(); ap.add_argument("--nodes", type=int, default=12500); ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = S.mus_graph(a.nodes, levels=3, seed=0)
torch.manual_seed(0)
model = gfd.nn.NsThreeScaleGNN(arch=S.mus_arch("NsThreeScaleGNN", 128), device=dev)
for name, ro in (("Rollout", Rollout(model, g.clone().to(dev), a.steps + 6, capture=True)),
                 ("DistributedRollout(world=1)", P.DistributedRollout(model, g, a.steps + 6, 0, 1, dev, capture=True))):
    ro.run(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); ro.run(a.steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:28s} {a.nodes} nodes: {a.steps / dt:8.1f} steps/s ({1e3 * dt / a.steps:.3f} ms/step)")

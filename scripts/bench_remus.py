"""REMuS-GNN (3-scale, k=5) rollout timing on a synthetic mesh. Usage: python scripts/bench_remus.py [--nodes N] [--steps K]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import synthetic as S
from graphs4cfd_amd.nn.model import Rollout
ap = argparse.ArgumentParser(); ap.add_argument("--nodes", type=int, default=100_000); ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
t0 = time.perf_counter(); g = S.remus_graph(a.nodes, k=5, seed=0); print(f"graph build {time.perf_counter() - t0:.1f} s: E={g.edge_index.size(1)} A={g.angle_index.size(1)} "
      f"E2={g.edge_index2.size(1)} E3={g.edge_index3.size(1)}")
torch.manual_seed(0)
model = gfd.nn.NsRotEquiTreeScaleGNN(arch=S.remus_arch(128), device=dev)
ro = Rollout(model, g.to(dev), a.steps + 4, capture=True)
ro.run(3); torch.cuda.synchronize()
t0 = time.perf_counter(); ro.run(a.steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"REMuS 3-scale, {a.nodes} nodes: {a.steps / dt:.2f} steps/s ({1e3 * dt / a.steps:.2f} ms/step), finite={bool(torch.isfinite(ro.outputs).all())}, "
      f"mem={torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

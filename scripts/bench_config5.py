"""BASELINE config 5 on ONE GPU: 1M-node 3-D mesh, 4-scale MuS-GNN, hipGraph-captured rollout (the 8-GPU form is
`bench.py --gpus 8 --nodes 1000000 --model NsFourScaleGNN` on a 2-D mesh; this script checks the 3-D data path and the
memory / index ranges at full size).  Usage: python scripts/bench_config5.py [--nodes 1000000] [--steps 10]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import synthetic as S
from graphs4cfd_amd.nn.model import Rollout
ap = argparse.ArgumentParser(); ap.add_argument("--nodes", type=int, default=1_000_000); ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
g = S.mus_graph(a.nodes, levels=4, dim=3, seed=0)
print(f"graph build {time.perf_counter() - t0:.1f} s: N={a.nodes} E={g.edge_index.size(1)} levels: "
      + ", ".join(str(int(getattr(g, f'pos_{l}').size(0))) for l in (2, 3, 4)), flush=True)
torch.manual_seed(0)
model = gfd.nn.NsFourScaleGNN(arch=S.mus_arch("NsFourScaleGNN", 128, dim=3), device=dev)
ro = Rollout(model, g.to(dev), a.steps + 4, capture=True)
ro.run(3); torch.cuda.synchronize()
t0 = time.perf_counter(); ro.run(a.steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"NsFourScaleGNN, {a.nodes} nodes (3-D): {a.steps / dt:.2f} steps/s ({1e3 * dt / a.steps:.1f} ms/step), finite={bool(torch.isfinite(ro.outputs).all())}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

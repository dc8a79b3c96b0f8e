"""REMuS-GNN (config 3, rounded-bf16 operands) on the same random point set numbered (a) as generated, (b) along a Morton curve BEFORE
BuildRemusGraph builds edges / angles / interpolation tables: hipGraph-replayed steps/s, same process, interleaved."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import ops, synthetic as S
from graphs4cfd_amd.reorder import morton_order
from graphs4cfd_amd.nn.model import Rollout
dev = torch.device("cuda", 0); n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ops.set_mlp_precision(sys.argv[2] if len(sys.argv) > 2 else "bf16")
gen = torch.Generator().manual_seed(0)
pos = torch.rand(n, 2, generator=gen)
graphs = {"as generated": S.remus_graph(n, k=5, seed=0, pos=pos.clone(), device=dev),
          "morton": S.remus_graph(n, k=5, seed=0, pos=pos[morton_order(pos)].contiguous(), device=dev)}
torch.manual_seed(0)
model = gfd.nn.NsRotEquiTreeScaleGNN(arch=S.remus_arch(128), device=dev); model.eval()
res = {k: [] for k in graphs}
for rd in range(3):
    for k, g in graphs.items():
        ro = Rollout(model, g, 48, capture=True)
        ro.run(6); torch.cuda.synchronize()
        t0 = time.perf_counter(); ro.run(30); torch.cuda.synchronize()
        res[k].append(30 / (time.perf_counter() - t0)); ro.close()
print(f"REMuS-GNN {n} nodes, {ops.mlp_precision()}: " + "   ".join(f"{k}: {statistics.median(v):.2f} steps/s" for k, v in res.items()))

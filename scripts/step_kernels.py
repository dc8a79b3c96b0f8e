"""Per-kernel-family time of an eager headline step (ops.KernelTimer) under the library in G4C_LIB_PATH; prints the level-1 CSR's uniform degree."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import ops, plan, synthetic as S
from graphs4cfd_amd.nn.model import Rollout
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = S.mus_graph(n, levels=3, seed=0, device=dev)
torch.manual_seed(0)
model = gfd.nn.NsThreeScaleGNN(arch=S.mus_arch("NsThreeScaleGNN", 128), device=dev); model.eval()
ro = Rollout(model, g, 12, capture=False)
ro.run(3); torch.cuda.synchronize()
gg = ro.graph if hasattr(ro, "graph") else g
ep, csr = plan.edge_csr(gg.edge_index, gg.num_nodes)
print("level-1 csr: uniform_deg", csr.uniform_deg, "max_deg", csr.max_deg, "perm", csr.perm is None, "rows", csr.n)
with ops.KernelTimer() as kt:
    ro.run(4)
torch.cuda.synchronize()
for k, m in sorted(kt.summary().items(), key=lambda kv: -kv[1]["seconds"]):
    print(f"{k:24s} launches/step {m['launches'] / 4:6.1f}  total ms/step {1e3 * m['seconds'] / 4:7.3f}  avg us {1e6 * m['seconds'] / m['launches']:8.1f}")

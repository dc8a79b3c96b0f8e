"""Training iterations over FRESH batches (every batch a new collated graph, as in GNN.fit with a DataLoader): the static-plan
builders run once per batch on the host, so what matters is that they never read index tensors back from the device
(plan.remember_host) and are O(n).  A/B: python scripts/bench_fit_batches.py --no-host-copies"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import graphs4cfd_amd as gfd
from graphs4cfd_amd import plan, synthetic as S
if "--no-host-copies" in sys.argv:
    plan._REMEMBER_HOST = False
dev = torch.device("cuda", 0)
n = 7000
h = 2.0 * n ** -0.5
data = []
for i in range(32):
    g = S.mus_graph(n, levels=1, seed=i)
    g.target = torch.randn(n, 3)
    data.append(g)
coarsen = gfd.transforms.GridClustering([2 * h, 4 * h])
loader = gfd.DataLoader(data, batch_size=8, shuffle=False, transform=coarsen)
model = gfd.nn.NsThreeScaleGNN(arch=S.mus_arch("NsThreeScaleGNN", 128), device=dev)
crit = gfd.nn.GraphLoss(0.25)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
model.train()
def step(batch):
    pred = model.forward(batch, 0); loss = crit(batch, pred, batch.target); loss.backward(); opt.step(); opt.zero_grad(); return loss
times = []
for ep in range(4):
    for b in loader:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step(b.to(dev)); torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
import statistics
print(f"fresh batches (8 x {n} nodes), upload + step, median of iterations 5..16: {1e3*statistics.median(times[4:]):.1f} ms (min {1e3*min(times[4:]):.1f})")
t0 = time.perf_counter()
for _ in range(5): step(b)
torch.cuda.synchronize(); print(f"same batch again (plans cached): {1e3*(time.perf_counter()-t0)/5:.1f} ms / iteration")

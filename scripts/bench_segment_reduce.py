import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import ops, plan, synthetic as S
dev = torch.device("cuda", 0)
n = 100000; g = S.mus_graph(n, levels=1, seed=0).to(dev)
ep, csr = plan.edge_csr(g.edge_index, n)
e = torch.randn(6 * n, 128, device=dev); out = torch.empty(n, 128, device=dev)
f = lambda: ops.segment_reduce(e, csr, True, out=out)
for _ in range(5): f()
ts = []
for _ in range(9):
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); [f() for _ in range(10)]; t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t) / 10 * 1e3)
print(os.environ.get("G4C_SEG_LPR", "32"), f"median {statistics.median(ts):.1f} us  -> {358.8e6 / statistics.median(ts) / 1e6:.2f} TB/s")

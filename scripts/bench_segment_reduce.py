"""The CSR segment reduction on level-1 messages (600k x 128 fp32 into 100k targets) in rotating buffers (4 x 307 MB > the 256 MiB
Infinity Cache), with the plan's permutation and without.  (Round 3 tried a streaming form — a 32-lane group owning 2 / 4 / 8 / 16
consecutive segments, offsets in one load, the next segment's rows requested before the current one is added up: 4.65 / 4.19 / 3.42 /
3.83 TB/s against 4.71 for one segment per group: what this pattern needs is many independent groups in flight, not depth per group.)"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import ops, plan, synthetic as S
dev = torch.device("cuda", 0)
n = 100000; g = S.mus_graph(n, levels=1, seed=0).to(dev)
ep, csr = plan.edge_csr(g.edge_index, n)
bufs = [torch.randn(6 * n, 128, device=dev) for _ in range(4)]
out = torch.empty(n, 128, device=dev)
# a permuted plan over the same rows (pool_edge's shape: rows gathered through the plan's permutation)
idx = torch.randint(0, n, (6 * n,), device=dev)
csr_p = plan.build_csr(idx, n, dev)
nbytes = 4.0 * (6 * n * 128 + n * 128 + n + 1)
for name, c in (("sorted (no permutation)", csr), ("through a permutation", csr_p)):
    k = [0]
    def f():
        ops.segment_reduce(bufs[k[0] % 4], c, True, out=out); k[0] += 1
    for _ in range(8): f()
    ts = []
    for _ in range(9):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); [f() for _ in range(12)]; t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t) / 12 * 1e3)
    extra = 4.0 * 6 * n if c.perm is not None else 0.0
    print(f"{name:26s} median {statistics.median(ts):7.1f} us  -> {(nbytes + extra) / statistics.median(ts) / 1e6:.2f} TB/s algorithmic")

"""The in-step segment reductions of the headline step on the real plans of a 100k-node 3-level mesh, standalone, on rotating buffers:
pool_edge's feature mean (SELU on load, through the coarse-edge permutation; DownMP, reference nn/blocks.py:63-67) at both levels
and DownMP's cluster mean.  Usage: python scripts/bench_pool_edge.py [--nodes 100000]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan, synthetic as S
from graphs4cfd_amd.reorder import reorder_nodes
ap = argparse.ArgumentParser(); ap.add_argument("--nodes", type=int, default=100000)
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
g = S.mus_graph(a.nodes, levels=3, seed=0).to(dev)
re = reorder_nodes(g)
if re is not None:
    g = re[0]
H = 128
ei = g.edge_index
for lvl in (1, 2):
    pp = plan.pool_edge_plan(getattr(g, f"idx{lvl}_to_idx{lvl + 1}"), ei, True)
    csr = pp.csr
    E = int(ei.size(1))
    bufs = [torch.randn(E, H, device=dev) for _ in range(4 if lvl == 1 else 16)]
    out = torch.empty(csr.n_seg, H, device=dev)
    for it in range(3): ops.segment_reduce(bufs[it % len(bufs)], csr, True, src_act=_lib.ACT_SELU, out=out)
    torch.cuda.synchronize()
    ts = []
    for it in range(20):
        s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record(); ops.segment_reduce(bufs[it % len(bufs)], csr, True, src_act=_lib.ACT_SELU, out=out); t_.record(); torch.cuda.synchronize()
        ts.append(s_.elapsed_time(t_) * 1e3)
    nbytes = 4.0 * (csr.n * H + csr.n_seg * H + csr.n_seg + 1 + csr.n)
    us = statistics.median(ts)
    print(f"pool_edge level {lvl} -> {lvl + 1}: {E} fine edges, {csr.n} kept, {csr.n_seg} coarse edges (mean {csr.n / max(csr.n_seg, 1):.2f} rows, max {csr.max_deg}): "
          f"{us:7.1f} us (min {min(ts):.1f}) = {nbytes / us / 1e6:.2f} TB/s algorithmic ({nbytes / 1e6:.1f} MB)")
    ei = pp.edge_index

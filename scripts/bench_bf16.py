"""fp32 vs opt-in bf16-MFMA MLPs: per-launch time of the level-1 edge / node MLPs and whole-rollout rate.
Usage: python scripts/bench_bf16.py [--nodes N] [--model M | --remus]"""
import argparse, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import ops, synthetic as S
from graphs4cfd_amd.nn import blocks as B
from graphs4cfd_amd.nn.model import Rollout
ap = argparse.ArgumentParser(); ap.add_argument("--nodes", type=int, default=100_000); ap.add_argument("--model", default="NsThreeScaleGNN")
ap.add_argument("--remus", action="store_true"); ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0); H = 128
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
rows, n = 6 * a.nodes, a.nodes
e, v, agg = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
out_e, out_v = torch.empty(rows, H, device=dev), torch.empty(n, H, device=dev)


def t(f, reps=5):
    f(); f(); ts = []
    for _ in range(7):
        s, x = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); [f() for _ in range(reps)]; x.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(x) / reps * 1e3)
    return statistics.median(ts)


for prec in ("fp32", "bf16x6", "bf16"):
    ops.set_mlp_precision(prec)
    te = t(lambda: blk.edge_mlp.run_hoisted([ops.Source(e)], [(v, row), (v, col)], rows, 0, out=out_e))
    t3 = t(lambda: blk.edge_mlp.run_coded([ops.Source(e), ops.Source(v, index=row), ops.Source(v, index=col)], rows, 0, out=out_e))
    tn = t(lambda: blk.node_mlp.run_coded([ops.Source(agg), ops.Source(v)], n, 1, out=out_v))
    print(f"{prec}: edge MLP hoisted (incl. 2 product launches) {te:8.1f} us | plain 3-block {t3:8.1f} us | node MLP {tn:8.1f} us", flush=True)
    if a.remus:
        g = S.remus_graph(a.nodes, k=5, seed=0).to(dev); model = gfd.nn.NsRotEquiTreeScaleGNN(arch=S.remus_arch(128), device=dev)
    else:
        levels = {"NsOneScaleGNN": 1, "NsTwoScaleGNN": 2, "NsThreeScaleGNN": 3, "NsFourScaleGNN": 4}[a.model]
        g = S.mus_graph(a.nodes, levels=levels, seed=0).to(dev); model = getattr(gfd.nn, a.model)(arch=S.mus_arch(a.model, 128), device=dev)
    ro = Rollout(model, g, a.steps + 4, capture=True)
    ro.run(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); ro.run(a.steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{prec}: {'REMuS' if a.remus else a.model} {a.nodes} nodes: {a.steps / dt:.2f} steps/s ({1e3 * dt / a.steps:.2f} ms/step) finite={bool(torch.isfinite(ro.outputs).all())}", flush=True)
    ro.close() if hasattr(ro, "close") else None
ops.set_mlp_precision("fp32")

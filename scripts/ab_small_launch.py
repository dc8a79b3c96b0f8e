"""Same-process A/B of g4c_mlp_small_launch_tiles (the tile kernel's deep weight ring for launches of few tiles): hipGraph-replayed
rollout steps/s of several meshes per limit, interleaved rounds; and the outputs of the limits compared bit for bit.
Usage: python scripts/ab_small_launch.py [--limits 0,512,1024] [--rounds 3]"""
import argparse, os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import _lib, ops, synthetic as S
from graphs4cfd_amd.nn.model import Rollout
ap = argparse.ArgumentParser(); ap.add_argument("--limits", default="0,256,512,1024,4096"); ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--precision", default="f16x3")
a = ap.parse_args()
limits = [int(x) for x in a.limits.split(",")]
lib = _lib.load(); dev = torch.device("cuda", 0)
ops.set_mlp_precision(a.precision)
cases = [("NsTwoScaleGNN", 10_000, 2, 400), ("NsThreeScaleGNN", 12_500, 3, 300), ("NsThreeScaleGNN", 25_000, 3, 200), ("NsThreeScaleGNN", 100_000, 3, 60)]
for name, nodes, levels, steps in cases:
    graph = S.mus_graph(nodes, levels=levels, dim=2, seed=0, device=dev)
    torch.manual_seed(0)
    model = getattr(gfd.nn, name)(arch=S.mus_arch(name, 128, dim=2), device=dev); model.eval()
    res = {l: [] for l in limits}; outs = {}
    for rd in range(a.rounds):
        for l in limits:
            lib.g4c_mlp_small_launch_tiles(l)
            ro = Rollout(model, graph, steps + 16, capture=True)
            ro.run(8); torch.cuda.synchronize()
            t0 = time.perf_counter(); ro.run(steps); torch.cuda.synchronize(); el = time.perf_counter() - t0
            res[l].append(steps / el)
            if rd == 0: outs[l] = ro.outputs[:, : 3 * 8].clone()
            ro.close()
    same = all(torch.equal(outs[limits[0]], outs[l]) for l in limits)
    print(f"{name} {nodes} nodes: " + "   ".join(f"limit {l}: {statistics.median(res[l]):7.1f} steps/s" for l in limits) + f"   outputs bit-identical across limits: {same}")
lib.g4c_mlp_small_launch_tiles(512)

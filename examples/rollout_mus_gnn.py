"""Drop-in use of the MI355X path with the reference's API: build a mesh with `gfd.transforms`, create (or load) a MuS-GNN,
roll it out with `solve`.  With a trained checkpoint of the reference: `gfd.nn.NsThreeScaleGNN(checkpoint="NsThreeScaleGNN.chk")`.

    python examples/rollout_mus_gnn.py [--nodes 20000] [--steps 50] [--checkpoint file.chk]
"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd          # instead of: import graphs4cfd as gfd

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=20000); ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--checkpoint", default=None)
a = ap.parse_args()
dev = torch.device("cuda")

# a synthetic flow domain: random points, the pre-processing pipeline of examples/training/NsMuSGNN/NsThreeScaleGNN.py
torch.manual_seed(0)
graph = gfd.Graph(pos=torch.rand(a.nodes, 2))
h = 2.0 * a.nodes ** -0.5
graph = gfd.transforms.Compose([
    gfd.transforms.ConnectKNN(6),
    gfd.transforms.ScaleEdgeAttr(h),
    gfd.transforms.GridClustering([2 * h, 4 * h]),
])(graph)
graph.field = torch.randn(a.nodes, 3)                      # u, v, p at the last time step
graph.glob = torch.rand(a.nodes, 1)                        # e.g. the Reynolds number
graph.omega = (torch.rand(a.nodes, 1) > 0.9).float()       # boundary marker

if a.checkpoint:
    model = gfd.nn.NsThreeScaleGNN(checkpoint=a.checkpoint, device=dev)
else:
    model = gfd.nn.NsThreeScaleGNN(arch=gfd.synthetic.mus_arch("NsThreeScaleGNN", 128), device=dev)   # random weights

model.solve(graph, 3)                                      # builds the static mesh plan, packs the weights, captures the step
torch.cuda.synchronize(); t0 = time.perf_counter()
out = model.solve(graph, a.steps)                          # [N, 3 * steps], device-resident rollout (hipGraph replay)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{a.steps} steps on {a.nodes} nodes: {a.steps / dt:.1f} steps/s, output {tuple(out.shape)}, finite={bool(torch.isfinite(out).all())}")

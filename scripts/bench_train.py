"""Training-step throughput of the autograd path: forward + GraphLoss + backward + Adam step of a
MuS-GNN on a synthetic mesh, on the GPU (fused forward, recompute backward: autograd.py) and — on a bounded sample — with
the oracle on the host cores (torch autograd over the op-for-op restatement = what the reference's fit() executes).
Usage: python scripts/bench_train.py [--nodes 100000] [--model NsThreeScaleGNN] [--steps 10] [--cpu-steps 1]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import synthetic as S

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=100_000)
ap.add_argument("--model", default="NsThreeScaleGNN", help="a MuS-GNN class, NsThreeGuillardScaleGNN (gMuS) or NsRotEquiTreeScaleGNN (REMuS)")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--cpu-steps", type=int, default=1)
ap.add_argument("--phases", action="store_true", help="HIP-event time per phase of the backward pass (one extra step)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
nf = 3
if a.model == "NsRotEquiTreeScaleGNN":
    g_cpu, arch, nf = S.remus_graph(a.nodes, k=5, seed=0), S.remus_arch(128), 2
elif "Guillard" in a.model:
    g_cpu, arch = S.mugs_graph(a.nodes, levels={"Two": 2, "Three": 3, "Four": 4}[a.model[2:].split("Guillard")[0]], seed=0), S.mugs_arch(a.model, 128)
else:
    levels = {"NsOneScaleGNN": 1, "NsTwoScaleGNN": 2, "NsThreeScaleGNN": 3, "NsFourScaleGNN": 4}[a.model]
    g_cpu, arch = S.mus_graph(a.nodes, levels=levels, seed=0), S.mus_arch(a.model, 128)
g_cpu.target = torch.randn(a.nodes, nf)
torch.manual_seed(0)
model = getattr(gfd.nn, a.model)(arch=arch, device=dev)
g = g_cpu.clone().to(dev)
crit = gfd.nn.GraphLoss(lambda_d=0.25)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)        # (as GNN.fit does on the GPU)
model.train()


def step():
    pred = model.forward(g, 0)
    loss = crit(g, pred, g.target)
    loss.backward()
    opt.step()
    opt.zero_grad()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
# forward only (same launches, recorded for autograd) and inference forward for scale
t0 = time.perf_counter()
for _ in range(a.steps):
    pred = None                      # (release the previous graph and the activations it keeps before building the next)
    pred = model.forward(g, 0)
torch.cuda.synchronize()
fwd = (time.perf_counter() - t0) / a.steps
pred = None
with torch.no_grad():
    model.forward(g, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        model.forward(g, 0)
    torch.cuda.synchronize()
    inf = (time.perf_counter() - t0) / a.steps
out = {"workload": f"{a.model} H=128 training step (forward + GraphLoss + backward + Adam) on a {a.nodes}-node synthetic 2D mesh",
       "gpu_ms_per_training_step": 1e3 * dt, "gpu_training_steps_per_s": 1 / dt, "gpu_ms_forward_recorded": 1e3 * fwd,
       "gpu_ms_forward_inference_eager": 1e3 * inf, "gpu_peak_memory_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
       "loss": float(loss)}
if a.phases:
    from graphs4cfd_amd import autograd as A
    A.PROFILE = {}
    t0 = time.perf_counter()
    step()
    ph = A.profile_summary()
    A.PROFILE = None
    out["backward_phases_ms"] = {k: round(v, 2) for k, v in sorted(ph.items(), key=lambda kv: -kv[1])}
    out["backward_phases_total_ms"] = round(sum(ph.values()), 2)
if a.cpu_steps > 0:
    from oracle import g4c_oracle as O
    import torch.nn.functional as F
    w = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    copt = torch.optim.Adam(list(w.values()), lr=1e-4)
    gd = g_cpu.to_dict()
    threads = min(16, os.cpu_count())          # (more threads are slower and erratic on the 2-socket host: profiles/r01_cpu_thread_sweep.log)
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    fwd = (lambda: O.remus_forward(gd, w)) if a.model == "NsRotEquiTreeScaleGNN" else \
          (lambda: O.mugs_forward(a.model, gd, w, 3)) if "Guillard" in a.model else (lambda: O.mus_forward(a.model, gd, w, 3))
    for _ in range(a.cpu_steps):
        closs = O.graph_loss(gd, fwd(), gd["target"], 0.25)
        closs.backward()
        copt.step()
        copt.zero_grad()
    cdt = (time.perf_counter() - t0) / a.cpu_steps
    out.update({"cpu_ms_per_training_step": 1e3 * cdt, "cpu_cores": threads, "cpu_kind": "port (oracle + torch autograd)",
                "cpu_sample": f"{a.cpu_steps} training step(s) of the same mesh and weights", "gpu_over_cpu": cdt / dt})
print(json.dumps(out))

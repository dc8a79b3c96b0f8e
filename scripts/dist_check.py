"""2+ rank end-to-end check of DistributedRollout against the single-process rollout.
Launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/dist_check.py [--backend nccl|gloo] [--same-gpu]
(--same-gpu: every rank uses cuda:0 with the gloo transport -- for single-GPU boxes)"""
import argparse, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import partition as P, synthetic as S
ap = argparse.ArgumentParser(); ap.add_argument("--backend", default="nccl"); ap.add_argument("--same-gpu", action="store_true")
ap.add_argument("--nodes", type=int, default=20000); ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--model", default="NsThreeScaleGNN"); ap.add_argument("--dim", type=int, default=2); ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--capture", type=int, default=-1, help="-1: capture the step in a hipGraph iff the transport is nccl")
ap.add_argument("--force-exchange", action="store_true", help="enter every halo collective even with one rank (zero-length splits): executes "
                "'hipGraph capture with an RCCL collective inside' on a single-GPU box")
ap.add_argument("--hoist-min-rows", type=int, default=None, help="blocks.HOIST_MIN_ROWS for this run (0: every MP layer hoists, i.e. every halo "
                "exchange carries first-layer products)")
ap.add_argument("--time", type=int, default=0, help="after the check: time this many steps of the captured (hipGraph-replayed) and of the uncaptured "
                "(eager launches + eager collectives) partitioned step — what a capture failure on the first real RCCL run would cost per rank")
a = ap.parse_args()
if a.hoist_min_rows is not None:
    from graphs4cfd_amd.nn import blocks as _B
    _B.HOIST_MIN_ROWS = a.hoist_min_rows
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0 if a.same_gpu else int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group(a.backend)
torch.manual_seed(2)
if a.model == "NsRotEquiTreeScaleGNN":           # REMuS-GNN: edge-latent halo (partition_remus.py)
    g = S.remus_graph(a.nodes, k=5, seed=1)
    model = gfd.nn.NsRotEquiTreeScaleGNN(arch=S.remus_arch(a.hidden), device=dev)
else:
    levels = {"NsOneScaleGNN": 1, "NsTwoScaleGNN": 2, "NsThreeScaleGNN": 3, "NsFourScaleGNN": 4}[a.model]
    g = S.mus_graph(a.nodes, levels=levels, dim=a.dim, seed=1)
    model = getattr(gfd.nn, a.model)(arch=S.mus_arch(a.model, a.hidden, dim=a.dim), device=dev)
dr = P.DistributedRollout(model, g, a.steps, rank, world, dev, capture=(a.backend == "nccl") if a.capture < 0 else bool(a.capture))
if a.force_exchange:
    dr.fwd.xch.force = True
dr.run(a.steps)
full = dr.gather_outputs()
if rank == 0:
    ref = model.solve(g.clone().to(dev), a.steps, capture=False)
    err = (full - ref).abs().max().item()
    x = dr.fwd.xch
    print(f"world={world} backend={a.backend} {a.model} dim={a.dim}: halo rows per level {dr.mesh.n_halo}, "
          f"{x.n_exchanges} exchanges, {x.bytes_sent} B sent, captured={dr.captured}, max|partitioned - single| = {err:.3e}", flush=True)
    assert err < 5e-4, err          # (the full-forward parity bar of tests/test_gpu_parity.py; measured 1e-6 .. 3e-6)
    if a.capture > 0:
        assert dr.captured, f"the partitioned step was not captured: {dr.capture_error}"
if a.time:
    import time
    res, per_step = {}, 0
    for name, cap in (("captured (hipGraph replay)", True), ("uncaptured (eager launches and collectives)", False)):
        r = P.DistributedRollout(model, g, a.time + 4, rank, world, dev, capture=cap)
        if a.force_exchange:
            r.fwd.xch.force = True
        r.run(3)                                   # eager step, capture step, one replay
        dist.barrier(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        r.run(a.time)
        dist.barrier(); torch.cuda.synchronize(dev)
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        res[name] = (1e3 * float(el.item()) / a.time, r.captured, r.capture_error)
        if not cap:
            per_step = r.fwd.xch.n_exchanges // (a.time + 3)          # (every step of the eager runner enters its collectives from Python)
        del r
    if rank == 0:
        for name, (ms, captured, note) in res.items():
            print(f"world={world} backend={a.backend} {a.model} {a.nodes} nodes, {per_step} exchanges per step: {name}: {ms:.3f} ms/step "
                  f"(captured={captured}{'' if note is None else ', ' + str(note)})", flush=True)
        c, u = res["captured (hipGraph replay)"][0], res["uncaptured (eager launches and collectives)"][0]
        print(f"eager fallback / captured = {u / c:.2f}x", flush=True)
dist.barrier()
dist.destroy_process_group()

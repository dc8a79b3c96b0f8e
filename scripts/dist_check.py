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
a = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0 if a.same_gpu else int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group(a.backend)
g = S.mus_graph(a.nodes, levels=3, seed=1)
torch.manual_seed(2)
model = gfd.nn.NsThreeScaleGNN(arch=S.mus_arch("NsThreeScaleGNN", 128), device=dev)
dr = P.DistributedRollout(model, g, a.steps, rank, world, dev, capture=(a.backend == "nccl"))
dr.run(a.steps)
full = dr.gather_outputs()
if rank == 0:
    ref = model.solve(g.clone().to(dev), a.steps, capture=False)
    err = (full - ref).abs().max().item()
    print(f"world={world} backend={a.backend}: halo rows per level {dr.mesh.n_halo}, max|partitioned - single| = {err:.3e}", flush=True)
    assert err < 2e-3, err
dist.barrier()
dist.destroy_process_group()

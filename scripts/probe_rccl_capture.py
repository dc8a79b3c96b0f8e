"""Probe: does an RCCL all_to_all_single survive torch.cuda.graph capture + replay on this stack?"""
import os, sys, torch, torch.distributed as dist
def P(*a): print(*a, flush=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
P("init...", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
mode = sys.argv[1] if len(sys.argv) > 1 else "lazy"
if mode == "eager":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group("nccl", rank=0, world_size=1)
P("init done")
a = torch.arange(8., device=dev); b = torch.zeros(8, device=dev)
dist.all_reduce(a); torch.cuda.synchronize(); P("allreduce ok")
dist.all_to_all_single(b, a, [8], [8]); torch.cuda.synchronize(); P("eager a2a ok", b.tolist())
g = torch.cuda.CUDAGraph()
a.mul_(2)
try:
    with torch.cuda.graph(g):
        dist.all_to_all_single(b, a, [8], [8])
        b.add_(1)
    g.replay(); torch.cuda.synchronize(); P("capture+replay ok", b.tolist())
    a.mul_(2); g.replay(); torch.cuda.synchronize(); P("replay 2 ok", b.tolist())
except Exception as e:
    P("capture FAILED:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()

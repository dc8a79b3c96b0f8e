"""Oracle (CPU baseline) step time vs torch thread count on this host."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import synthetic as S
import graphs4cfd_amd as gfd
from oracle import g4c_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
g = S.mus_graph(n, levels=3, seed=0).to_dict()
torch.manual_seed(0)
m = gfd.nn.NsThreeScaleGNN(arch=S.mus_arch("NsThreeScaleGNN", 128))
w = {k: v.detach() for k, v in m.state_dict().items()}
print("cpu_count", os.cpu_count())
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    with torch.no_grad():
        O.mus_forward("NsThreeScaleGNN", g, w, 3)
        t0 = time.perf_counter(); O.mus_forward("NsThreeScaleGNN", g, w, 3); dt = time.perf_counter() - t0
    print(f"threads {th:4d}: {dt:.2f} s/step at {n} nodes", flush=True)

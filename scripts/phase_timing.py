"""Debug: per-phase cycle stamps of the fused MLP kernel (needs a -DG4C_TIMING build of mlp_fused.hip)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import ops, plan, synthetic as S, _lib
from graphs4cfd_amd.nn import blocks as B
dev = torch.device("cuda", 0)
n = 100_000
g = S.mus_graph(n, levels=1, seed=0).to(dev)
E, H = g.edge_index.size(1), 128
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
v, e = torch.randn(n, H, device=dev), torch.randn(E, H, device=dev)
ep, csr = plan.edge_csr(g.edge_index, n)
out = torch.empty(E, H, device=dev)
srcs = [ops.Source(e), ops.Source(v, ep.row), ops.Source(v, ep.col)]
for _ in range(3):
    blk.edge_mlp.run_hoisted([ops.Source(e)], [(v, ep.row), (v, ep.col)], E, 0, out=out)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.g4c_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.g4c_debug_read_stamps(buf.ctypes.data, buf.size)
st = buf.reshape(4096, 16).astype(np.int64)
names = ["prologue(idx,ring fill,first gather)", "layer0 MFMA loop (12 chunks)", "epilogue 0", "layer1 loop (4 chunks)", "epilogue 1",
         "layer2 loop", "epilogue 2 (last)"]
idx = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7)]
print("rc", rc, "tiles", (st[:, 13] > 0).sum())
for nm, (a, b) in zip(names, idx):
    d = st[:, b] - st[:, a]
    print(f"{nm:40s} median {np.median(d):9.0f}  p90 {np.percentile(d, 90):9.0f} ticks")
d = st[:, 12] - st[:, 7]; print(f"{'LayerNorm/act pass':40s} median {np.median(d):9.0f}  p90 {np.percentile(d, 90):9.0f}")
d = st[:, 13] - st[:, 12]; print(f"{'store':40s} median {np.median(d):9.0f}  p90 {np.percentile(d, 90):9.0f}")
d = st[:, 13] - st[:, 0]; print(f"{'whole tile':40s} median {np.median(d):9.0f}  p90 {np.percentile(d, 90):9.0f}")

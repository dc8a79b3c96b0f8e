"""Phase timing of the persistent MLP kernel (needs a library built with -DG4C_PX_TIMING: scripts/build_px_timing.sh).
Prints, for workgroup 0, the cycle length of the M / X phases and the barrier waits of both wave groups."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
lib_path = sys.argv[1] if len(sys.argv) > 1 else "graphs4cfd_amd/lib/libg4c_pxtiming.so"
_lib.LIB_PATH = os.path.abspath(lib_path)
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
if len(sys.argv) > 2: ops.set_mlp_precision(sys.argv[2])
lib = _lib.load()
lib.g4c_mlp_px6_enable(1)
lib.g4c_px_read_stamps.restype = C.c_int; lib.g4c_px_read_stamps.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda", 0); H = 128
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
rows = 600000; n = rows // 6
e = torch.randn(rows, H, device=dev); pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
pk_e = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
src_e = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
out = torch.empty(rows, H, device=dev)
for _ in range(3):
    ops.mlp_forward(pk_e, src_e, rows, 0, out=out)
torch.cuda.synchronize()
st = np.zeros(2048, dtype=np.uint64)
lib.g4c_px_read_stamps(st.ctypes.data, 2048)
st = st.reshape(2, 1024).astype(np.int64)
t0 = st[0, 0]
names = ["matrix waves", "helper waves"]
for g in range(2):
    s = st[g] - t0
    print(f"{names[g]}: first stamp at {s[0]}")
    # two stamps per interval: begin of the role's work, end of it (then the workgroup barrier)
    for k in range(0, 120, 2):
        b, e = s[k:k + 2]
        nxt = s[k + 2]
        print(f"  interval {k // 2:3d}  work {e - b:6d}  barrier wait {nxt - e:6d}   (t = {b})")

sub = np.zeros(256 * 8, dtype=np.uint64)
lib.g4c_px_read_sub.restype = C.c_int; lib.g4c_px_read_sub.argtypes = [C.c_void_p, C.c_int]
lib.g4c_px_read_sub(sub.ctypes.data, 256 * 8)
sub = sub.reshape(256, 8).astype(np.int64)
print("helper sub-stamps per interval (cycles from interval begin): 0 start | 1 previous unit done (park / finish / head) | 2 presum done | 3 gathers issued | 4 rows requested | end")
for k in range(1, 40):
    b, e = st[1, 2 * k], st[1, 2 * k + 1]
    s_ = sub[k]
    def d(q):
        v = int(s_[q] - b)
        return f"{v:6d}" if 0 <= v < 100000 else "     -"
    print(f"  interval {k:3d}: " + "  ".join(d(q) for q in (0, 1, 2, 5, 3, 4)) + f"   end {int(e - b):6d}")

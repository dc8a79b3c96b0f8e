"""Cycle stamps of the first 256 workgroups of mlp_bx6i_kernel (wave 0), level-1 edge launch.  Needs a -DG4C_BX6I_TIMING build:
hipcc ... -DG4C_BX6I_TIMING -c mlp_bx6i.hip ; python scripts/bx6i_stamps.py <lib.so>"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B
lib = C.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, args) in _lib._SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
_lib._lib = lib
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0); H = 128; rows = 600000; n = rows // 6
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
e, pr, pc = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
lib.g4c_mlp_bx6i_enable(0); lib.g4c_mlp_bx6w_enable(2)
for _ in range(3): ops.mlp_forward(pk, src, rows)
torch.cuda.synchronize()
buf = np.zeros(256 * 32, dtype=np.uint64)
lib.g4c_bx6w_read_stamps.argtypes = [C.c_void_p, C.c_int]
lib.g4c_bx6w_read_stamps(buf.ctypes.data, buf.size)
st = buf.reshape(256, 32).astype(np.int64)
names = {0: "start", 1: "indices + ring fill, barrier", 2: "rows landed, park (64 rows)", 3: "start values (additive rows)", 4: "  barrier"}
for l in range(3):
    names[5 + 4 * l] = f"M({l}): 192 MFMAs"; names[6 + 4 * l] = "  barrier"; names[7 + 4 * l] = f"epilogue({l})"; names[8 + 4 * l] = "  barrier"
names[20] = "LayerNorm + stores"
keys = sorted(names)
prev = keys[0]
for k in keys[1:]:
    d = st[:, k] - st[:, prev]
    print(f"{names[k]:34s} median {int(np.median(d)):7d}  (p10 {int(np.percentile(d, 10)):7d}, p90 {int(np.percentile(d, 90)):7d})")
    prev = k
print("tile lifetime median", int(np.median(st[:, 20] - st[:, 0])))

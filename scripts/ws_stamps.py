"""Cycle stamps of the weight-stationary kernel (mlp_ws.hip), level-1 message launch with the fused aggregation: per workgroup the
SECOND pair of its range (steady state of the software pipeline), wave 0.  Needs a -DG4C_WS_TIMING build of mlp_ws.hip:
  bash scripts/build_ws_timing.sh ; python scripts/ws_stamps.py graphs4cfd_amd/lib/libg4c_ws_timing.so [bf16]
bf16: REMuS-GNN's level-1 angle launch instead (rounded-bf16 mode, 2 layers, 5 rows per target, bf16 rows in and out)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B
lib = C.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, args) in _lib._SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
_lib._lib = lib
torch.set_grad_enabled(False)
bf16 = len(sys.argv) > 2 and sys.argv[2].startswith("bf16")
f32in = len(sys.argv) > 2 and sys.argv[2] == "bf16-f32in"
dev = torch.device("cuda", 0); H = 128
deg = 5 if bf16 else 6
rows = 2_500_000 if bf16 else 600000; n = rows // deg
torch.manual_seed(0)
ops.set_mlp_precision("bf16" if bf16 else "f16x3")
hid = (H, H) if bf16 else (H, H, H)
blk = B.GNBlock((3 * H, hid, True), (2 * H, hid, True)).to(dev)
e, pr, pc = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
colh = torch.arange(n).repeat_interleave(deg)
rowh = (colh + torch.randint(-64, 65, (rows,))).clamp(0, n - 1) if bf16 else torch.randint(0, n, (rows,))
ei = torch.stack([rowh, colh]).to(dev)
ep, csr = plan.edge_csr(ei, n)
pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
src = [ops.Source(e) if f32in else ops.Source(e.to(torch.bfloat16)) if bf16 else ops.Source(e, pre_act=_lib.ACT_SELU),
       ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
out, agg = torch.empty(rows, H, device=dev), torch.empty(n, H, device=dev)
lib.g4c_mlp_ws_enable(2)
for _ in range(3):
    if bf16: ops.mlp_forward(pk, src, rows, 0, agg=(csr, agg, True), rows_dtype=torch.bfloat16, rows_act=_lib.ACT_SELU)
    else: ops.mlp_forward(pk, src, rows, 0, out=out, agg=(csr, agg, True))
torch.cuda.synchronize()
buf = np.zeros(256 * 32, dtype=np.uint64)
lib.g4c_ws_read_stamps.argtypes = [C.c_void_p, C.c_int]
lib.g4c_ws_read_stamps(buf.ctypes.data, buf.size)
st = buf.reshape(256, 32).astype(np.int64)
names = {0: "loop top", 1: "meta + table loads issued", 2: "M(A,0) + park B, barrier", 3: "tables -> LDS, gathers issued, M(B,0) + E(A,0), barrier",
         4: "M(A,1) + E(B,0), barrier", 5: "M(B,1) + E(A,1), barrier", 6: "M(A,2) + E(B,1), barrier", 7: "M(B,2) + F(A), F(B), barrier",
         8: "open next pair (park A', start values)", 9: "LayerNorm + row stores", 10: "[barrier,] aggregation"}
if bf16:
    names = {0: "loop top", 1: "meta + table loads issued", 2: "M(A,0) + park B, barrier", 3: "tables -> LDS, gathers issued, M(B,0) + E(A,0), barrier",
             4: "M(A,1) + E(B,0), barrier", 7: "M(B,1) + F(A), F(B), barrier", 8: "open next pair (park A', start values)",
             9: "LayerNorm + row stores", 10: "[barrier,] aggregation"}
keys = sorted(names)
prev = keys[0]
for k in keys[1:]:
    d = st[:, k] - st[:, prev]
    print(f"{names[k]:62s} median {int(np.median(d)):7d}  (p10 {int(np.percentile(d, 10)):7d}, p90 {int(np.percentile(d, 90)):7d})")
    prev = k
for a_, b_, nm in ((9, 15, "  of which: wait at the aggregation barrier"), (15, 16, "  of which: aggregation"), (16, 10, "  of which: tile B additive gathers issued")):
    if st[:, 15].any():
        d = st[:, b_] - st[:, a_]
        print(f"{nm:62s} median {int(np.median(d)):7d}  (p10 {int(np.percentile(d, 10)):7d}, p90 {int(np.percentile(d, 90)):7d})")
if st[:, 17].any():
    for a_, b_, nm in ((8, 17, "  LayerNorm: the fp32 rows read back from LDS"), (17, 18, "  LayerNorm: statistics (DPP), normalise[, activation]"),
                       (18, 9, "  LayerNorm: rows back to LDS + global row stores issued"), (15, 20, "  aggregation: segment offsets read, addresses"),
                       (20, 21, "  aggregation: rows read from LDS and added"), (21, 22, "  aggregation: mean"), (22, 16, "  aggregation: store issued, loop exit")):
        d = st[:, b_] - st[:, a_]
        print(f"{nm:62s} median {int(np.median(d)):7d}  (p10 {int(np.percentile(d, 10)):7d}, p90 {int(np.percentile(d, 90)):7d})")
print("pair period (loop top -> end of tail) median", int(np.median(st[:, 10] - st[:, 0])))
live = st[:, 14] > 0
per = (st[live, 13] - st[live, 12]) / st[live, 14]
print(f"whole launch, ticks per pair and workgroup: median {np.median(per):.0f}, p10 {np.percentile(per, 10):.0f}, p90 {np.percentile(per, 90):.0f}, max {per.max():.0f}")

"""Where a SMALL launch of the tile kernel (mlp_bx6_kernel, node shape: [aggregate | v] -> 3 layers -> LayerNorm -> SELU, + two heads)
spends its time: cycle stamps per phase, median over the launch's tiles.  The blocks rotate over 16 different MLPs, as in a step (a
launch's weights were last read a whole step ago).  Needs a -DG4C_TIMING build of mlp_fused.hip:
  bash scripts/build_variant.sh graphs4cfd_amd/lib/libg4c_timing.so -DG4C_TIMING ; G4C_LIB_PATH=graphs4cfd_amd/lib/libg4c_timing.so python scripts/small_launch_stamps.py [rows ...]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import ops, _lib
from graphs4cfd_amd.nn import blocks as B
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0); H = 128
lib = _lib.load()
lib.g4c_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
sizes = [int(a) for a in sys.argv[1:]] or [400, 1600, 3200, 12500, 100000]
blks = [B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev) for _ in range(16)]
names = {1: "indices / biases -> LDS, weight ring filled, first rows loaded, barrier", 2: "block 0 parked (split -> planes)", 3: "start values, barrier",
         4: "layer 0: two 128-k blocks (+ park of block 1)", 5: "epilogue 0 (SELU, split), barrier", 6: "layer 1 MFMAs, barrier", 7: "epilogue 1, barrier",
         8: "layer 2 MFMAs, barrier", 9: "last layer -> fp32 tile, barrier", 12: "(-)", 13: "LayerNorm / SELU / row stores", 14: "tile -> planes for the heads (2 barriers)",
         15: "two heads: MFMAs + stores"}
for n in sizes:
    agg, v = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    flush = torch.empty(64 << 20, device=dev)
    def one(b): return b.node_mlp.run_with_heads([ops.Source(agg), ops.Source(v)], n, _lib.ACT_SELU, b.edge_mlp, H, [H, H])
    for b in blks: one(b)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(4):
        for b in blks: one(b)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / 64
    flush.zero_(); one(blks[3]); torch.cuda.synchronize()           # (256 MB written in between: the weights come from HBM / MALL)
    buf = np.zeros(4096 * 16, dtype=np.uint64)
    lib.g4c_debug_read_stamps(buf.ctypes.data, buf.size)
    st = buf.reshape(4096, 16).astype(np.int64)[: min(4096, (n + 31) // 32)]
    print(f"== {n} rows ({(n + 31) // 32} tiles): {us:.1f} us per launch back to back (eager, 16 MLPs rotating); stamps of one launch after a 256 MB flush")
    prev = 0
    for k in sorted(names):
        if k == 12: prev = 12 if st[:, 12].any() else prev; continue
        d = st[:, k] - st[:, prev]
        print(f"   {names[k]:78s} median {int(np.median(d)):7d}  p90 {int(np.percentile(d, 90)):7d}")
        prev = k
    d = st[:, 15] - st[:, 0]
    print(f"   {'whole tile':78s} median {int(np.median(d)):7d}  p90 {int(np.percentile(d, 90)):7d}   first start -> last end {int(st[:, 15].max() - st[:, 0].min())}")

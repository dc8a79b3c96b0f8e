"""Debug: per-phase cycle stamps (wave 0 of each tile) of the column-split / small-launch MLP kernels.
Needs a -DG4C_TIMING build: bash scripts/build_variant.sh /tmp/libT.so -DG4C_TIMING ; python scripts/phase_timing_split.py /tmp/libT.so"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B
lib = C.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, args) in _lib._SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
_lib._lib = lib
lib.g4c_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
PREC = os.environ.get("G4C_MLP_PRECISION", "fp32")      # the stamps also exist in the bf16x6 kernel
dev = torch.device("cuda", 0); H = 128
torch.set_grad_enabled(False)
torch.manual_seed(0)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
pk_e = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
pk_v = blk.node_mlp.packed([H, H], [False, False])
names = ["prologue: indices, bias -> LDS, barrier", "first gather issued + acc-init gathers", "ring fill issue, first park, barrier",
         "layer 0 MFMA", "hidden store 0", "layer 1 MFMA", "hidden store 1", "layer 2 MFMA", "hidden store 2 (last)"]
for rows in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "192,3072,75000").split(",")]:
    n = max(rows // 6, 32)
    e, pr, pc = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
    col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
    aa, vv = torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev)
    src_e = [ops.Source(e), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
    flush = torch.empty(64 << 20, device=dev)     # 256 MB: push weights / inputs out of L2 and most of the Infinity Cache
    for mode in ((324,) if PREC == "fp32" else (None,)):
        for case, f in (("edge", lambda: ops.mlp_forward(pk_e, src_e, rows, 0, tile_mode=mode)),
                        ("node", lambda: ops.mlp_forward(pk_v, [ops.Source(aa), ops.Source(vv)], rows, 1, tile_mode=mode))):
            for cold in (False, True):
                f(); f()
                if cold: flush.fill_(1.0)
                torch.cuda.synchronize()
                s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); f(); t.record(); torch.cuda.synchronize()
                buf = np.zeros(4096 * 16, dtype=np.uint64)
                lib.g4c_debug_read_stamps(buf.ctypes.data, buf.size)
                st = buf.reshape(4096, 16).astype(np.int64)[: min(4096, (rows + 31) // 32)]
                d = np.diff(st[:, :10], axis=1)
                parts = " ".join(f"{int(np.median(d[:, k])):6d}" for k in range(9))
                fin = int(np.median(st[:, 13] - st[:, 12])); tot = int(np.median(st[:, 13] - st[:, 0]))
                span = int(st[:, 13].max() - st[:, 0].min())
                nx = ((rows + 31) // 32) // 8            # tiles of XCD 0 (contiguous tile range per XCD)
                if 0 < nx <= 4096:
                    x0 = st[:nx]
                    t0, t1 = int(x0[:, 0].min()), int(x0[:, 13].max())
                    conc = [int(((x0[:, 0] <= T) & (x0[:, 13] > T)).sum()) for T in np.linspace(t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0), 7)]
                    gaps = np.sort(x0[:, 0])
                    print(f"    XCD 0: {nx} tiles in {t1 - t0} ticks; tiles in flight at 7 sample times {conc} (32 CUs); "
                          f"sum of tile lifetimes / span = {float((x0[:, 13] - x0[:, 0]).sum()) / (t1 - t0):.1f}")
                print(f"rows {rows:6d} mode {mode} {case} {'cold' if cold else 'hot '}: event {s.elapsed_time(t) * 1e3:7.1f} us | phases {parts} | finish {fin:6d} | tile {tot:7d} | launch span {span:8d} ticks")
print("phases:", "; ".join(names))

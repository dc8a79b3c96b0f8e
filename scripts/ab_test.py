"""Within-process A/B timing of two builds of libg4c.so on the hoisted edge MLP / node MLP (interleaved rounds,
median + min), because run-to-run (box-to-box, DVFS) noise between separate invocations is ~5 %.
Usage: python scripts/ab_test.py libA.so libB.so [--rows 600000] [--rounds 15]"""
import argparse, ctypes as C, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B

ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--rows", type=int, default=600000)
ap.add_argument("--rounds", type=int, default=15); ap.add_argument("--inner", type=int, default=3)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--modes", default="", help="comma list of tile modes: each library is timed once per mode (default: the library's own policy)")
a = ap.parse_args()
_modes = [int(m) for m in a.modes.split(",")] if a.modes else [None]
a.libs, modes = [l for l in a.libs for _ in _modes], [m for _ in a.libs for m in _modes]


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib._SIGNATURES.items():
        if hasattr(lib, name):       # older builds lack the newest entry points
            fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    return lib


libs = [load(p) for p in a.libs]
ops.set_mlp_precision(a.precision)
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0); H = 128
torch.manual_seed(0)
_lib._lib = libs[0]
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
rows = a.rows; n = rows // 6
e = torch.randn(rows, H, device=dev); v = torch.randn(n, H, device=dev); agg = torch.randn(n, H, device=dev)
row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
pr, pc = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
out_e, out_v = torch.empty(rows, H, device=dev), torch.empty(n, H, device=dev)
def pack_for(lib):   # weights are packed by the library that consumes them (the stream layout may differ)
    _lib._lib = lib
    blk.edge_mlp._packed.clear(); blk.node_mlp._packed.clear()
    return blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False), blk.node_mlp.packed([H, H], [False, False])


packs = [pack_for(lib) for lib in libs]
cur = [0]
src_e = [ops.Source(e), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
src_v = [ops.Source(agg), ops.Source(v)]
cases = {"edge(hoisted)": lambda: ops.mlp_forward(packs[cur[0]][0], src_e, rows, 0, out=out_e, tile_mode=modes[cur[0]]),
         "node": lambda: ops.mlp_forward(packs[cur[0]][1], src_v, n, 1, out=out_v, tile_mode=modes[cur[0]])}
ref = {}
for cname, fn in cases.items():
    times = [[] for _ in libs]
    for li, lib in enumerate(libs):
        _lib._lib = lib; cur[0] = li; fn(); fn(); torch.cuda.synchronize()
        ref.setdefault(cname, []).append((out_e if cname.startswith("edge") else out_v).clone())
    for r in range(a.rounds):
        for li, lib in enumerate(libs):
            _lib._lib = lib; cur[0] = li
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); [fn() for _ in range(a.inner)]; t.record(); torch.cuda.synchronize()
            times[li].append(s.elapsed_time(t) / a.inner * 1e3)
    for li, p in enumerate(a.libs):
        d = (ref[cname][li] - ref[cname][0]).abs().max().item()
        print(f"{cname:14s} {os.path.basename(p) + ('' if modes[li] is None else f' mode {modes[li]}'):28s} median {statistics.median(times[li]):8.1f} us   min {min(times[li]):8.1f} us   max|out - out_A| {d:.1e}")

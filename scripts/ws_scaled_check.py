"""mlp_ws_kernel with the values to split carried as y * 2^11 (G4C_WS_SCALED, mlp_ws.hip) against the build without it
(-DG4C_WS_SCALED=0): outputs of the message MLP and its fused aggregate must be BIT-identical, also for rows scaled by 1e-4, 1e3
and 3e4 (fp16 subnormal low parts, values past the fp16 range end: both builds clip and flag them).
Usage: python scripts/ws_scaled_check.py <libg4c.so> <libg4c_unscaled.so>"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops, plan
from graphs4cfd_amd.nn import blocks as B


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    return lib


libs = [load(p) for p in sys.argv[1:3]]
torch.set_grad_enabled(False)
ops.set_mlp_precision("f16x3")
dev = torch.device("cuda", 0); H = 128; rows = 120000; n = rows // 6
torch.manual_seed(0)
_lib._lib = libs[0]
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
colh = torch.arange(n).repeat_interleave(6)
ei = torch.stack([torch.randint(0, n, (rows,)), colh]).to(dev)
ep, csr = plan.edge_csr(ei, n)
bad = 0
for scale in (1.0, 1e-4, 1e3, 3e4):
    e, pr, pc = torch.randn(rows, H, device=dev) * scale, torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    for pre in (_lib.ACT_SELU, _lib.ACT_NONE):
        res = []
        for lib in libs:
            _lib._lib = lib
            lib.g4c_mlp_bx6i_enable(0); lib.g4c_mlp_ws_enable(2)
            blk.edge_mlp._packed.clear()
            pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
            src = [ops.Source(e, pre_act=pre), ops.Source(pr, index=ep.row, additive=True), ops.Source(pc, index=ep.col, additive=True)]
            out, agg = torch.empty(rows, H, device=dev), torch.empty(n, H, device=dev)
            ops.f16_range_report(dev)          # (clear)
            ops.mlp_forward(pk, src, rows, 0, out=out, agg=(csr, agg, True))
            res.append((out.clone(), agg.clone(), bool(ops.f16_range_report(dev))))
        same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
        bad += not same
        print(f"{'ok  ' if same else 'FAIL'} scale {scale:g} pre_act {pre}: bitwise equal {same}, finite {bool(torch.isfinite(res[0][0]).all())}, "
              f"clip flagged {res[0][2]} / {res[1][2]}  (max |diff| {(res[0][0] - res[1][0]).abs().max().item():.2e})")
print("all bit-identical" if not bad else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)

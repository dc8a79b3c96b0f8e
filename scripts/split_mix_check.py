"""The four-instruction fp16 split (split_pair_f16 with v_fma_mixlo_f16 / v_fma_mixhi_f16, mlp_common.h) against the conversion /
subtraction form it replaced (-DG4C_SPLIT_MIX=0 build): outputs of the message MLP must be BIT-identical, also for rows scaled by
1e-4, 1e3 and 3e4 (fp16 subnormal low parts, values near the fp16 range end).
Usage: python scripts/split_mix_check.py <libg4c.so> <libg4c_nomix.so>"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    return lib


libs = [load(p) for p in sys.argv[1:3]]
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0); H = 128; rows = 60000; n = rows // 6
torch.manual_seed(0)
_lib._lib = libs[0]
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
bad = 0
for scale in (1.0, 1e-4, 1e3, 3e4):
    e, pr, pc = torch.randn(rows, H, device=dev) * scale, torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
    row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)
    col = (torch.arange(rows, device=dev) // 6).clamp(max=n - 1).to(torch.int32)
    outs = {}
    for mode in (0, 2):          # tile kernel, then every specialised kernel the library has
        res = []
        for lib in libs:
            _lib._lib = lib
            lib.g4c_mlp_bx6i_enable(mode)
            if hasattr(lib, "g4c_mlp_ws_enable"): lib.g4c_mlp_ws_enable(0)
            blk.edge_mlp._packed.clear()
            pk = blk.edge_mlp._packed_cols("hoist", 0, H, [H], [False], False)
            src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(pr, index=row, additive=True), ops.Source(pc, index=col, additive=True)]
            res.append(ops.mlp_forward(pk, src, rows).clone())
        same = torch.equal(res[0], res[1])
        bad += not same
        print(f"{'ok  ' if same else 'FAIL'} scale {scale:g} kernel mode {mode}: mix split == conversion split bitwise: {same}  (max |diff| {(res[0] - res[1]).abs().max().item():.2e})")
print("all bit-identical" if not bad else f"{bad} MISMATCHES")

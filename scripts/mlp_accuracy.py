"""Accuracy of the fused-MLP arithmetic modes against an fp64 evaluation of the same MLP (hoisted edge MLP + node MLP,
H = 128): max / mean |error| of fp32 MFMA, bf16x6 (three-way bf16 split, six products), f16x3 (two-way fp16 split, three
products) and plain bf16; then f16x3 on inputs scaled by 1e-4 / 1e3 (range behaviour of the fp16 split)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphs4cfd_amd import _lib, ops
from graphs4cfd_amd.nn import blocks as B
dev = torch.device("cuda", 0); H = 128; rows = 20000; n = rows // 6
torch.manual_seed(1)
blk = B.GNBlock((3 * H, (H, H, H), True), (2 * H, (H, H, H), True)).to(dev)
e, v = torch.randn(rows, H, device=dev), torch.randn(n, H, device=dev)
row = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32); col = torch.randint(0, n, (rows,), device=dev, dtype=torch.int32)


def f64(mlp, x):
    y = x.double()
    lin = mlp._linears()
    for li, l in enumerate(lin):
        y = y @ l.weight.detach().double().T + l.bias.detach().double()
        if li < len(lin) - 1:
            y = torch.selu(y)
    ln = mlp.MLP.layer_norm
    return torch.nn.functional.layer_norm(y, (H,), ln.weight.double(), ln.bias.double(), ln.eps)


ref = f64(blk.edge_mlp, torch.cat([torch.selu(e), v[row.long()], v[col.long()]], 1))
for prec in ("fp32", "bf16x6", "f16x3", "bf16"):
    ops.set_mlp_precision(prec)
    src = [ops.Source(e, pre_act=_lib.ACT_SELU), ops.Source(v, index=row), ops.Source(v, index=col)]
    y = blk.edge_mlp.run_coded(src, rows)
    d = (y.double() - ref).abs()
    yh = blk.edge_mlp.run_hoisted([ops.Source(e, pre_act=_lib.ACT_SELU)], [(v, row), (v, col)], rows)
    dh = (yh.double() - ref).abs()
    print(f"{prec:7s} 3-block: max {d.max().item():.3e} mean {d.mean().item():.3e} | hoisted: max {dh.max().item():.3e} mean {dh.mean().item():.3e}")
# range behaviour: the same MLP on scaled edge rows (the additive node terms unscaled), f16x3 against bf16x6
for scale in (1e-4, 1e3, 3e4):
    es = e * scale
    refs = f64(blk.edge_mlp, torch.cat([torch.selu(es), v[row.long()], v[col.long()]], 1))
    for prec in ("bf16x6", "f16x3"):
        ops.set_mlp_precision(prec)
        y = blk.edge_mlp.run_coded([ops.Source(es, pre_act=_lib.ACT_SELU), ops.Source(v, index=row), ops.Source(v, index=col)], rows)
        d = (y.double() - refs).abs()
        print(f"{prec:7s} edge rows x {scale:g}: max {d.max().item():.3e} mean {d.mean().item():.3e} finite {bool(torch.isfinite(y).all())}")
ops.set_mlp_precision("fp32")

"""Rounded-bf16 mode (BASELINE config 3), REMuS-GNN forward at 20k nodes against the fp32 oracle: the first-layer products stored as
fp32 (round 4) and as bf16 (round 5, blocks.PRODUCTS_BF16), same weights and mesh.  Usage: python scripts/remus_bf16_err.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import ops, synthetic as S
from graphs4cfd_amd.nn import blocks as B
from oracle import g4c_oracle as O
torch.set_grad_enabled(False)
DEV = torch.device("cuda", 0)
ops.set_mlp_precision("bf16")
for seed in (21, 31):
    g = S.remus_graph(20_000, k=5, seed=seed)
    torch.manual_seed(seed + 1)
    model = gfd.nn.NsRotEquiTreeScaleGNN(arch=S.remus_arch(128), device=DEV)
    ref = O.remus_forward(g.to_dict(), {k: v.cpu() for k, v in model.state_dict().items()})
    for on in (False, True):
        B.PRODUCTS_BF16 = on
        y = model.forward(g.clone().to(DEV)).cpu()
        d = (y - ref).abs()
        print(f"seed {seed} products {'bf16' if on else 'fp32'}: max {d.max().item():.3e} mean {d.mean().item():.3e} (reference scale {ref.abs().mean().item():.3e})")

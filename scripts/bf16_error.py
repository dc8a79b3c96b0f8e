"""Deviation of the opt-in bf16-MFMA MLP mode (and of the default fp32 mode) from the fp32 oracle, full forwards."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd
from graphs4cfd_amd import synthetic as S, ops
from oracle import g4c_oracle as O
DEV = torch.device("cuda", 0)
g = S.mus_graph(6000, levels=3, seed=3); arch = S.mus_arch("NsThreeScaleGNN", 128)
torch.manual_seed(5); model = gfd.nn.NsThreeScaleGNN(arch=arch, device=DEV)
w = {k: v.cpu() for k, v in model.state_dict().items()}
ref = O.mus_forward("NsThreeScaleGNN", g.to_dict(), w, 3)
for prec in ("fp32", "bf16"):
    ops.set_mlp_precision(prec)
    with torch.no_grad(): y = model.forward(g.clone().to(DEV)).cpu()
    d = (y - ref).abs(); print("MuS3", prec, "max", d.max().item(), "mean", d.mean().item(), "ref scale", ref.abs().mean().item())
g = S.remus_graph(1500, k=5, seed=4); torch.manual_seed(6)
model = gfd.nn.NsRotEquiTreeScaleGNN(arch=S.remus_arch(128), device=DEV)
w = {k: v.cpu() for k, v in model.state_dict().items()}
ref = O.remus_forward(g.to_dict(), w)
for prec in ("fp32", "bf16"):
    ops.set_mlp_precision(prec)
    with torch.no_grad(): y = model.forward(g.clone().to(DEV)).cpu()
    d = (y - ref).abs(); print("REMuS", prec, "max", d.max().item(), "mean", d.mean().item(), "ref scale", ref.abs().mean().item())

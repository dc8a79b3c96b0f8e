import sys, torch
sys.path.insert(0, "/root/repo")
import graphs4cfd_amd as gfd
from graphs4cfd_amd import ops, synthetic as S
from oracle import g4c_oracle as O
torch.set_grad_enabled(False)
DEV = torch.device("cuda", 0)
ops.set_mlp_precision("bf16")
g = S.remus_graph(20_000, k=5, seed=21)
torch.manual_seed(22)
model = gfd.nn.NsRotEquiTreeScaleGNN(arch=S.remus_arch(128), device=DEV)
ref = O.remus_forward(g.to_dict(), {k: v.cpu() for k, v in model.state_dict().items()})
y = model.forward(g.clone().to(DEV)).cpu()
d = (y - ref).abs()
print("max", d.max().item(), "mean", d.mean().item(), "ref scale", ref.abs().mean().item())

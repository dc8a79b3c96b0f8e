"""Training with the reference's API on the MI355X path — the structure of examples/training/NsMuSGNN/NsTwoScaleGNN.py:
TrainConfig, a per-sample transform pipeline (ConnectKNN, ScaleEdgeAttr), a batch-level transform (GridClustering),
gfd.DataLoader, model.fit.  The NsCircle HDF5 dataset is replaced by an in-memory list of synthetic trajectories (there is no
h5py / network here): a scalar-diffusion-like map on the kNN graph, so there is something to learn.

    python examples/train_mus_gnn.py [--samples 24] [--nodes 3000] [--epochs 8]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd          # instead of: import graphs4cfd as gfd

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=24); ap.add_argument("--nodes", type=int, default=3000)
ap.add_argument("--epochs", type=int, default=8); ap.add_argument("--folder", default="/tmp")
a = ap.parse_args()
h = 2.0 * a.nodes ** -0.5

train_config = gfd.nn.TrainConfig(
    name            = 'NsTwoScaleGNN_synthetic',
    folder          = a.folder,
    chk_interval    = 1,
    training_loss   = gfd.nn.losses.GraphLoss(lambda_d=0.25),
    validation_loss = gfd.nn.losses.GraphLoss(),
    epochs          = a.epochs,
    num_steps       = [1, 2, 3],
    add_steps       = {'tolerance': 0.05, 'loss': 'training'},
    batch_size      = 4,
    lr              = 1e-3,
    grad_clip       = {"epoch": 0, "limit": 1},
    scheduler       = {"factor": 0.5, "patience": 5, "loss": 'training'},
    stopping        = 1e-8,
    device          = torch.device('cuda'),
)

transform = gfd.transforms.Compose([gfd.transforms.ConnectKNN(6), gfd.transforms.ScaleEdgeAttr(h)])
batch_transform = gfd.transforms.Compose([gfd.transforms.GridClustering([2 * h])])


def sample(seed: int, n_out: int) -> gfd.Graph:
    gen = torch.Generator().manual_seed(seed)
    g = transform(gfd.Graph(pos=torch.rand(a.nodes, 2, generator=gen)))
    x, y = g.pos[:, 0:1], g.pos[:, 1:2]
    k = 1 + torch.randint(0, 3, (3,), generator=gen).float()
    g.field = torch.cat((torch.sin(6.28 * k[0] * x), torch.cos(6.28 * k[1] * y), torch.sin(6.28 * k[2] * (x + y))), 1)
    g.glob = torch.full((a.nodes, 1), float(torch.rand(1, generator=gen)))
    g.omega = ((x < 0.03) | (x > 0.97) | (y < 0.03) | (y > 0.97)).float()
    row, col = g.edge_index
    f, out = g.field, []
    for _ in range(n_out):           # one explicit smoothing step per time step, rate set by `glob`; boundary nodes held fixed
        mean = torch.zeros_like(f).index_add_(0, col, f[row]) / 6.0
        f = torch.where(g.omega.bool(), f, f + (0.2 + 0.6 * g.glob) * (mean - f))
        out.append(f)
    g.target = torch.cat(out, 1)
    return g


n_out = train_config['num_steps'][-1]
train_set = [sample(s, n_out) for s in range(a.samples)]
test_set = [sample(10_000 + s, n_out) for s in range(max(a.samples // 6, 2))]
train_loader = gfd.DataLoader(train_set, batch_size=train_config['batch_size'], shuffle=True, transform=batch_transform)
val_loader = gfd.DataLoader(test_set, batch_size=train_config['batch_size'], shuffle=False, transform=batch_transform)

model = gfd.nn.NsTwoScaleGNN(arch=gfd.synthetic.mus_arch("NsTwoScaleGNN", 128), device=train_config['device'])
print("Number of trainable parameters: ", model.num_params)
model.fit(train_config, train_loader, val_loader=val_loader)
first, last = model.history[0], model.history[-1]
print(f"training loss {first['training_loss']:.3e} -> {last['training_loss']:.3e}, validation {first['validation_loss']:.3e} -> "
      f"{last['validation_loss']:.3e}, rollout length {first['n_out']} -> {last['n_out']}")

"""Training with the reference's API on the MI355X path — examples/training/NsMuSGNN/NsTwoScaleGNN.py with its own structure
and transform lists: TrainConfig, gfd.datasets.NsCircle, the per-sample pipeline (periodic kNN connect, field / edge scaling,
random rotation / flip, noise), GridClustering as the batch-level transform, gfd.DataLoader, model.fit.

The NsCircle HDF5 file is replaced by a synthetic array in the same record layout (data[simulation, node, x | y | Re | boundary
code | u v p per time step], NaN-padded to the largest mesh): decaying travelling waves on random point clouds — there is no
network / h5py here; with the real file pass path="NsCircle.h5" instead of data=.

    python examples/train_mus_gnn.py [--samples 24] [--nodes 3000] [--epochs 8]
"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphs4cfd_amd as gfd          # instead of: import graphs4cfd as gfd

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=24); ap.add_argument("--nodes", type=int, default=3000)
ap.add_argument("--epochs", type=int, default=8); ap.add_argument("--folder", default="/tmp")
a = ap.parse_args()
T = 20                                                        # time steps per simulation

# Training configuration (as in the reference script; fewer epochs, larger lr: this is a demonstration)
train_config = gfd.nn.TrainConfig(
    name            = 'NsTwoScaleGNN_synthetic',
    folder          = a.folder,
    chk_interval    = 1,
    training_loss   = gfd.nn.losses.GraphLoss(lambda_d=0.25),
    validation_loss = gfd.nn.losses.GraphLoss(),
    epochs          = a.epochs,
    num_steps       = [1, 2, 3],
    add_steps       = {'tolerance': 0.02, 'loss': 'training'},
    batch_size      = 4,
    lr              = 1e-3,
    grad_clip       = {"epoch": 0, "limit": 1},
    scheduler       = {"factor": 0.5, "patience": 5, "loss": 'training'},
    stopping        = 1e-8,
    device          = torch.device('cuda'),
)


def synthetic_ns_circle(n_sim: int, n_max: int, seed: int) -> torch.Tensor:
    """Records in the NsCircle layout; meshes of different sizes (NaN padding), domain [0, 4] x [0, 1] periodic in y."""
    gen = torch.Generator().manual_seed(seed)
    data = torch.full((n_sim, n_max, 4 + 3 * T), float("nan"))
    for s in range(n_sim):
        n = n_max - int(torch.randint(0, n_max // 10, (1,), generator=gen))
        x, y = 4 * torch.rand(n, generator=gen), torch.rand(n, generator=gen)
        re = 500 + 500 * float(torch.rand(1, generator=gen))
        bound = torch.zeros(n)
        bound[x < 0.05] = 2; bound[x > 3.95] = 3                                        # inlet / outlet
        bound[((x - 1) ** 2 + (y - 0.5) ** 2) < 0.02] = 4                               # "cylinder" wall
        k = 1 + int(torch.randint(0, 3, (1,), generator=gen))
        cols = [x, y, torch.full((n,), re), bound]
        for t in range(T):
            decay, phase = math.exp(-0.02 * t * 1000 / re), 2 * math.pi * (k * y) - 0.3 * t
            cols += [1.0 + 0.8 * decay * torch.sin(phase) * torch.cos(0.5 * math.pi * x), 0.6 * decay * torch.cos(phase), -0.5 * decay * torch.sin(2 * phase)]
        data[s, :n] = torch.stack(cols, 1)
    return data


# Training datasets: the reference's per-sample and batch-level transform lists
transform = gfd.transforms.Compose([
    gfd.transforms.ConnectKNN(6, period=[None, "auto"]),
    gfd.transforms.ScaleNs({'u': (-2.1, 2.6), 'v': (-2.25, 2.1), 'p': (-3.7, 2.35), 'Re': (500, 1000)}, format='uvp'),
    gfd.transforms.ScaleEdgeAttr(0.1),
    gfd.transforms.RandomGraphRotation(eq='ns', format='uvp'),
    gfd.transforms.RandomGraphFlip(eq='ns', format='uvp'),
    gfd.transforms.AddUniformNoise(0.01),
])
batch_transform = gfd.transforms.Compose([
    gfd.transforms.GridClustering([0.15]),
])
info = {"n_in": 1, "n_out": train_config['num_steps'][-1], "step": 1, "T": T}
dataset = gfd.datasets.NsCircle(format='uvp', data=synthetic_ns_circle(a.samples + max(a.samples // 6, 2), a.nodes, seed=0), training_info=info, transform=transform)
train_set, test_set = torch.utils.data.random_split(dataset, [a.samples, len(dataset) - a.samples])
train_loader = gfd.DataLoader(train_set, batch_size=train_config['batch_size'], shuffle=True, transform=batch_transform)
val_loader = gfd.DataLoader(test_set, batch_size=train_config['batch_size'], shuffle=False, transform=batch_transform)

# Model definition (the published 2S-GNN arch)
model = gfd.nn.NsTwoScaleGNN(arch=gfd.synthetic.mus_arch("NsTwoScaleGNN", 128), device=train_config['device'])
print("Number of trainable parameters: ", model.num_params)

# Training
model.fit(train_config, train_loader, val_loader=val_loader)
first, last = model.history[0], model.history[-1]
print(f"training loss {first['training_loss']:.3e} -> {last['training_loss']:.3e}, validation {first['validation_loss']:.3e} -> "
      f"{last['validation_loss']:.3e}, rollout length {first['n_out']} -> {last['n_out']}")

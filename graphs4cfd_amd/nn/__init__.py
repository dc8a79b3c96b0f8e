"""`gfd.nn`: blocks, MuS-GNN, gMuS-GNN and REMuS-GNN models, the GNN base with the rollout loop."""
from . import blocks
from .mus_gnn import *
from .mugs_gnn import NsTwoGuillardScaleGNN, NsThreeGuillardScaleGNN, NsFourGuillardScaleGNN
from .remus_gnn import NsRotEquiTreeScaleGNN
from .model import GNN, TrainConfig, collate
from .losses import GraphLoss

"""Training losses (reference: graphs4cfd/nn/losses.py:5-16)."""
import torch.nn as nn
import torch.nn.functional as F


class GraphLoss(nn.Module):
    """MSE over all nodes, plus `lambda_d` times the L1 error on the Dirichlet-boundary nodes (`graph.omega[:, 0] == 1`) when
    `lambda_d > 0` and the batch has any (nn/losses.py:10-16).  The loss works on the [N, num_fields] prediction — a few
    hundred kB — and is ordinary torch arithmetic; its gradient enters the fused blocks through autograd.py."""

    def __init__(self, lambda_d=0):
        super().__init__()
        self.lambda_d = lambda_d

    def forward(self, graph, pred, target):
        loss = F.mse_loss(pred, target)
        if self.lambda_d > 0:
            dirichlet_boundary = (graph.omega[:, 0] == 1)
            if dirichlet_boundary.any():
                loss = loss + self.lambda_d * F.l1_loss(pred[dirichlet_boundary], target[dirichlet_boundary])
        return loss

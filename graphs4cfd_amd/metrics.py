"""`gfd.metrics` (reference: graphs4cfd/metrics.py:4-23)."""
import torch


def r2(pred: torch.Tensor, target: torch.Tensor) -> float:
    """Coefficient of determination between `pred` and `target` (one time-point [N] / [N, F] or a whole rollout [N, F*T]):
    1 - sum (target - pred)^2 / sum (target - mean(target))^2, over the entries where target differs from its mean."""
    if pred.dim() not in (1, 2):
        raise RuntimeError("r2 expects a time-point or a rollout: a 1-D or 2-D tensor")
    mean = target.mean()
    mask = target != mean
    res = ((target[mask] - pred[mask]) ** 2).sum()
    tot = ((target[mask] - mean) ** 2).sum()
    return float(1 - res / tot)

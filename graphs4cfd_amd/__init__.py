"""graphs4cfd_amd — MI355X-native implementation of graphs4cfd's message-passing hot path.

Drop-in for the hot-path slice of `import graphs4cfd as gfd`:

    import graphs4cfd_amd as gfd
    model = gfd.nn.NsThreeScaleGNN(arch=arch, device=torch.device('cuda'))
    out = model.solve(graph, n_out)          # graph: gfd.Graph with the reference's attribute layout

Compute runs in hand-written HIP kernels (libg4c.so, C-ABI in include/g4c.h); there is no CPU or
eager-torch fallback.  Training (`model.fit`, `gfd.nn.TrainConfig`, `gfd.nn.GraphLoss`, `gfd.DataLoader`) runs the same fused
forward recorded for autograd (autograd.py).  Out of scope: plotting.  The package is also importable under
the reference's name (`import graphs4cfd as gfd`, the alias package at the repository root).
"""
from .graph import Graph
from . import nn, plan, ops, synthetic, transforms, metrics, datasets
from .loader import DataLoader, Collater
from .ops import check_f16_range, f16_range_report     # clipped values of the default arithmetic are reported, never silent (ops.py)
from .nn.model import set_forward_validation            # bare forward() calls validate their own fp16 range (one flag read per call)
from .ops import mlp_precision, set_mlp_precision      # "f16x3" (default: fp32-class two-way fp16 split on the matrix pipe) | "bf16x6" | "fp32" | "bf16" (opt-in)

__version__ = "0.1.0"

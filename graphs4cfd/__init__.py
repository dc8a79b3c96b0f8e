"""`import graphs4cfd as gfd` — the reference's package name, bound to the MI355X implementation.

The reference's examples/ scripts import `graphs4cfd` (and `from graphs4cfd import nn, transforms, datasets, metrics`); with this
repository on PYTHONPATH in place of the reference they run on `graphs4cfd_amd` unchanged: every submodule is registered in
`sys.modules` under the reference's dotted name, so `import graphs4cfd.nn`, `from graphs4cfd.transforms import ConnectKNN`
and `graphs4cfd.nn.NsThreeScaleGNN` resolve to the same objects as their `graphs4cfd_amd` counterparts (no second copy of
any class — `isinstance` checks agree across the two names).  `graphs4cfd.plot` is out of scope (plotting)."""
import sys as _sys

import graphs4cfd_amd as _impl
from graphs4cfd_amd import *                                                                                # noqa: F401,F403
from graphs4cfd_amd import Graph, DataLoader, Collater, nn, transforms, metrics, datasets, plan, ops, synthetic      # noqa: F401

__version__ = _impl.__version__

for _name in ("graph", "loader", "nn", "transforms", "metrics", "datasets", "plan", "ops", "synthetic", "partition", "augment"):
    _mod = __import__(f"graphs4cfd_amd.{_name}", fromlist=["_"])
    _sys.modules[f"{__name__}.{_name}"] = _mod
    globals()[_name] = _mod
for _name in ("blocks", "losses", "model", "training", "mus_gnn", "mugs_gnn", "remus_gnn"):
    _sys.modules[f"{__name__}.nn.{_name}"] = __import__(f"graphs4cfd_amd.nn.{_name}", fromlist=["_"])


def __getattr__(name):
    if name == "plot":
        raise ImportError("graphs4cfd.plot is not part of the MI355X hot-path implementation (SURVEY.md §8: plotting is out of scope); "
                          "use the reference's plot.py on tensors moved to the CPU")
    raise AttributeError(f"module 'graphs4cfd' has no attribute {name!r}")

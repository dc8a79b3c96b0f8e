"""`Graph`: the attribute-bag data container of the hot path.

The reference's `graphs4cfd.graph.Graph` (graph.py:6-19) is a `torch_geometric.data.Data`; the hot
path only ever uses it as a bag of named tensors (`graph.field`, `graph.edge_index`, `graph.pos_2`,
`getattr(graph, f'idx{h}_to_idx{l}')`, `hasattr(graph, 'loc')`, `graph.num_nodes`, `graph.to(device)`;
nn/blocks.py:223-227, nn/mus_gnn.py:71, nn/model.py:311-312).  This class provides exactly that
surface without PyG, with the attribute layout of SURVEY.md §8(b).  Plotting helpers are out of
scope.
"""
from __future__ import annotations

from typing import Any, Dict, Iterator, List

import torch


class Graph:
    def __init__(self, **kwargs: Any):
        for k, v in kwargs.items():
            setattr(self, k, v)

    # -- PyG-compatible conveniences used by the reference's callers -------------------------
    @property
    def num_nodes(self) -> int:
        for key in ("pos", "x", "field", "batch"):
            v = self.__dict__.get(key)
            if v is not None:
                return int(v.size(0))
        ei = self.__dict__.get("edge_index")
        if ei is not None:
            return int(ei.max()) + 1
        raise AttributeError("Graph has no node attribute to infer num_nodes from")

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.size(1))

    def keys(self) -> List[str]:
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __contains__(self, key: str) -> bool:
        return key in self.__dict__

    def __getitem__(self, key: str) -> Any:
        return self.__dict__[key]

    def __setitem__(self, key: str, value: Any) -> None:
        setattr(self, key, value)

    def __iter__(self) -> Iterator:
        return iter((k, self.__dict__[k]) for k in self.keys())

    def to_dict(self) -> Dict[str, Any]:
        return {k: self.__dict__[k] for k in self.keys()}

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "Graph":
        return cls(**d)

    def to(self, device, non_blocking: bool = False) -> "Graph":
        """Moves every tensor attribute in place (like `Data.to`) and returns self."""
        from . import plan as _plan
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                moved = v.to(device, non_blocking=non_blocking)
                if v.device.type == "cpu" and moved.device.type != "cpu" and not v.dtype.is_floating_point:
                    # the static-plan builders run on the host: keep the host image of index tensors so they never
                    # have to read them back (a read-back waits for the whole launch queue)
                    _plan.remember_host(moved, v)
                self.__dict__[k] = moved
        return self

    def clone(self) -> "Graph":
        return Graph(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})

    def __repr__(self) -> str:
        parts = [f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}" for k, v in self.__dict__.items()
                 if not k.startswith("_")]
        return f"Graph({', '.join(parts)})"

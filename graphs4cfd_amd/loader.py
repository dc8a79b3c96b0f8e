"""`gfd.DataLoader` (reference: graphs4cfd/loader.py:7-75): a torch DataLoader whose collate function merges a list of `Graph`s
into one batch graph and then applies batch-level transforms.

The reference delegates the merge to `torch_geometric.data.Batch.from_data_list` after correcting the REMuS angle indices
(loader.py:17-55: `angle_index*` address EDGES, so they are offset by the running edge count of their level instead of the
running node count).  Here the merge is `nn.model.collate` (node-indexed attributes offset by the node count) with the same
correction applied up front."""
from typing import Callable, List, Optional, Sequence

import torch.utils.data

from .graph import Graph
from .nn.model import collate


class Collater(object):
    def __init__(self, transform: Optional[Callable] = None):
        self.transform = transform

    @staticmethod
    def _fix_angle_indices(batch: Sequence[Graph]) -> None:
        """loader.py:17-55: pre-subtract the node offset `collate` will add and add the edge offset of the level the index
        addresses (rows of angle_index{l}: level-l edges; angle_index{l}{l+1}: row level l, col level l+1)."""
        elem = batch[0]
        levels = [s for s in ("", "2", "3", "4") if hasattr(elem, f"angle_index{s}")]
        for s in levels:
            num_nodes, num_edges = elem.num_nodes, int(getattr(elem, f"edge_index{s}").size(1))
            lvl = 1 if s == "" else int(s)
            for graph in batch[1:]:
                shift = num_edges - num_nodes
                setattr(graph, f"angle_index{s}", getattr(graph, f"angle_index{s}") + shift)
                nxt, prv = f"angle_index{lvl}{lvl + 1}", f"angle_index{lvl - 1}{lvl}"
                if hasattr(elem, nxt):
                    t = getattr(graph, nxt).clone()
                    t[0] += shift
                    setattr(graph, nxt, t)
                if lvl > 1 and hasattr(elem, prv):
                    t = getattr(graph, prv).clone()
                    t[1] += shift
                    setattr(graph, prv, t)
                num_nodes += graph.num_nodes
                num_edges += int(getattr(graph, f"edge_index{s}").size(1))

    def collate(self, batch: List[Graph]):
        # (shallow copies: the reference shifts the angle indices of the dataset's own graphs in place, loader.py:23-55, which
        # compounds from epoch to epoch when the dataset holds its graphs in memory)
        batch = [Graph(**{k: v for k, v in g.__dict__.items()}) for g in batch]
        self._fix_angle_indices(batch)
        out = collate(list(batch))
        return out if self.transform is None else self.transform(out)

    def __call__(self, batch):
        return self.collate(batch)


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size: int = 1, shuffle: bool = False, transform: Optional[Callable] = None, **kwargs):
        kwargs.pop("collate_fn", None)
        super().__init__(dataset, batch_size, shuffle, collate_fn=Collater(transform), **kwargs)

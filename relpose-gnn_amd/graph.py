"""Graph containers for the hot path: duck-typed stand-ins for PyG ``Data`` / ``Batch``.

The reference passes a ``torch_geometric.data.Batch`` to ``model(data)`` and the model reads
only ``data.x``, ``data.edge_index``, ``data.edge_attr`` and ``data.batch``
(/root/reference/python/niantic/modules/posenet.py:1034,1044-1050); callers also use
``data.y`` and ``data.to(device)`` (/root/reference/python/niantic/testing/test.py:211,216).
PyG is not a dependency of this package, so the same attribute contract is provided here.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def fc_edge_index(n: int) -> torch.Tensor:
    """Directed fully-connected edge list [2, n(n-1)] int64 in the order the reference datasets
    store it (dataset_7Scenes_multi.py:377-385 and :418-422): upper-triangular pairs grouped by
    offset d = 1..n-1 (i -> i+d), followed by the same pairs reversed.  For n = 8 column 28 is
    the edge 1 -> 0, the one test.py:227-229 uses to recover the query pose."""
    i = torch.arange(n)
    src = torch.cat([i[: n - d] for d in range(1, n)])
    dst = torch.cat([i[d:] for d in range(1, n)])
    return torch.cat([torch.stack([src, dst]), torch.stack([dst, src])], dim=1).to(torch.int64)


class Data:
    """One graph: x [n, 3*H*W], edge_index [2, E], y [n, 6], edge_attr [E, 6] (or None)."""

    _FIELDS = ("x", "edge_index", "y", "edge_attr", "batch")

    def __init__(self, x=None, edge_index=None, y=None, edge_attr=None, batch=None, **extra):
        self.x, self.edge_index, self.y, self.edge_attr, self.batch = x, edge_index, y, edge_attr, batch
        for k, v in extra.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_graphs(self) -> int:
        gs = getattr(self, "graph_sizes", None)
        if gs is not None:
            return len(gs[0])
        return 1 if self.batch is None else int(self.batch.max().item()) + 1

    def __len__(self) -> int:              # test.py:207 uses len(data)
        return self.num_graphs

    def to(self, device, non_blocking: bool = False):
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
        return out


class Batch(Data):
    """Disjoint union of graphs: node tensors concatenated along dim 0, ``edge_index`` along
    dim 1 with node-id offsets, ``batch[i]`` = graph id of node i (PyG collation)."""

    @classmethod
    def from_data_list(cls, graphs: Sequence[Data]) -> "Batch":
        xs, eis, ys, eas, bs = [], [], [], [], []
        off = 0
        for g, d in enumerate(graphs):
            n = d.num_nodes
            xs.append(d.x)
            eis.append(d.edge_index + off)
            if d.y is not None:
                ys.append(d.y)
            if d.edge_attr is not None:
                eas.append(d.edge_attr)
            bs.append(torch.full((n,), g, dtype=torch.int64))
            off += n
        out = cls(x=torch.cat(xs, 0), edge_index=torch.cat(eis, 1),
                  y=torch.cat(ys, 0) if len(ys) == len(graphs) else None,
                  edge_attr=torch.cat(eas, 0) if len(eas) == len(graphs) else None,
                  batch=torch.cat(bs, 0))
        # host-side slice table (nodes / edges per graph): lets the HIP module cut the batch at graph boundaries and run
        # the parts concurrently on several streams without a device round trip
        out.graph_sizes = ([d.num_nodes for d in graphs], [int(d.edge_index.shape[1]) for d in graphs])
        return out


def fc_batch(x: torch.Tensor, nodes_per_graph: int, y: Optional[torch.Tensor] = None) -> Batch:
    """Wrap node images x [B*n, 3HW] as a batch of B fully-connected n-node graphs."""
    n_total = x.shape[0]
    assert n_total % nodes_per_graph == 0
    b = n_total // nodes_per_graph
    ei = fc_edge_index(nodes_per_graph)
    offs = (torch.arange(b, dtype=torch.int64) * nodes_per_graph).view(b, 1, 1)
    edge_index = (ei.unsqueeze(0) + offs).permute(1, 0, 2).reshape(2, -1)
    batch = torch.arange(b, dtype=torch.int64).repeat_interleave(nodes_per_graph)
    edge_attr = None
    if y is not None:
        edge_attr = y[edge_index[1]] - y[edge_index[0]]      # dataset_7Scenes_multi.py:425-429
    out = Batch(x=x, edge_index=edge_index.to(x.device), y=y, edge_attr=edge_attr, batch=batch.to(x.device))
    out.graph_sizes = ([nodes_per_graph] * b, [ei.shape[1]] * b)
    return out

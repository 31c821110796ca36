"""Minimal graph containers with the surface the reference's drivers use on
``torch_geometric.data.Data`` / ``Batch`` (reference graph_network.py:61,98,172;
data_utils.py:218-227; train.py:111-112,247).  Real PyG objects are accepted
wherever these are: the engine only reads attributes.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch


class Data:
    """Attribute bag: keyword construction, attribute get/set, ``.to(device)``.
    ``hasattr(data, 'globals')`` is False until someone assigns it (the reference
    probes exactly that at graph_network.py:62,99,168)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    # -- introspection ---------------------------------------------------------
    def keys(self) -> List[str]:
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __contains__(self, key: str) -> bool:
        return key in self.__dict__

    def __repr__(self) -> str:
        parts = []
        for k in self.keys():
            v = self.__dict__[k]
            parts.append(f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}")
        return f"{type(self).__name__}({', '.join(parts)})"

    @property
    def num_nodes(self) -> Optional[int]:
        x = self.__dict__.get("x")
        if torch.is_tensor(x):
            return x.shape[0]
        pos = self.__dict__.get("pos")
        return pos.shape[0] if torch.is_tensor(pos) else None

    @property
    def num_edges(self) -> int:
        ei = self.__dict__.get("edge_index")
        return int(ei.shape[1]) if torch.is_tensor(ei) else 0

    # -- movement --------------------------------------------------------------
    def to(self, device, non_blocking: bool = False) -> "Data":
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device, non_blocking=non_blocking)
        return self

    def cpu(self) -> "Data":
        return self.to("cpu")

    def cuda(self, device=None) -> "Data":
        return self.to("cuda" if device is None else device)


class Batch(Data):
    """Disjoint union of graphs (``Batch.from_data_list``, reference train.py:247):
    node-level tensors are concatenated, ``edge_index`` is offset per graph,
    ``batch[i]`` is the graph of node i, ``num_graphs`` the count."""

    @classmethod
    def from_data_list(cls, graphs: Iterable[Data]) -> "Batch":
        graphs = list(graphs)
        if not graphs:
            raise ValueError("from_data_list needs at least one graph")
        out = cls()
        offsets, n_tot = [], 0
        for g in graphs:
            offsets.append(n_tot)
            n_tot += g.x.shape[0]
        keys = [k for k in graphs[0].keys()]
        for k in keys:
            vals = [getattr(g, k, None) for g in graphs]
            if any(v is None for v in vals):
                setattr(out, k, None)
            elif k == "edge_index":
                out.edge_index = torch.cat([v + off for v, off in zip(vals, offsets)], dim=1)
            elif torch.is_tensor(vals[0]):
                setattr(out, k, torch.cat(vals, dim=0))
            else:
                setattr(out, k, vals)
        dev = graphs[0].x.device
        out.batch = torch.cat([torch.full((g.x.shape[0],), i, dtype=torch.long, device=dev)
                               for i, g in enumerate(graphs)])
        out.num_graphs = len(graphs)
        # engine hints survive when every member has the same fixed in-degree
        # (a member's hint counts only while it is bound to the member's own edge_index, see _graph_arrays)
        def bound(g):
            ei = getattr(g, "edge_index", None)
            return torch.is_tensor(ei) and getattr(g, "_cgnn_fixed_k_for", None) == (ei.data_ptr(), ei._version, tuple(ei.shape))
        ks = {getattr(g, "_cgnn_fixed_k", None) if bound(g) else None for g in graphs}
        if len(ks) == 1 and None not in ks:
            out._cgnn_fixed_k = ks.pop()
            ei = out.edge_index
            out._cgnn_fixed_k_for = (ei.data_ptr(), ei._version, tuple(ei.shape))
        orders = [getattr(g, "_cgnn_order", None) for g in graphs]
        if all(o is not None for o in orders):
            out._cgnn_order = torch.cat([o + off for o, off in zip(orders, offsets)])
        return out

    def to(self, device, non_blocking: bool = False) -> "Batch":
        super().to(device, non_blocking)
        return self

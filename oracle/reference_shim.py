"""Run the reference's own ``graph_network.py`` / ``data_utils.py`` in the build
container, behind stand-ins for the packages the image lacks.

TEST INFRASTRUCTURE, build container only.  ``/root/reference`` does not exist
on the GPU box; nothing reachable from ``pytest -m gpu``, ``smoke()`` or
``bench.py`` imports this module.  It is used by ``oracle/make_golden.py`` to
produce the committed fixtures under ``tests/golden/`` and by the container-only
test ``tests/test_oracle_vs_reference.py`` (skipped when the checkout is absent).

The reference files are imported from where they lie (never copied, no
bytecode written).  They need three third-party names the image does not have
(SURVEY.md F5): ``torch_geometric.data.Data``, ``torch_geometric.nn.MessagePassing``
and ``torch_cluster.knn`` (``knn_graph`` and ``torch_scatter`` are imported by
the reference but never called).  The stand-ins below encode OUR reading of those
packages' published behaviour; every other line that runs is the reference's.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import torch

REFERENCE_DIR = os.environ.get("CGNN_REFERENCE_DIR", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "graph_network.py"))


class Data:
    """Attribute bag with the pieces of ``torch_geometric.data.Data`` the
    reference touches: keyword construction, attribute get/set, ``.to``;
    ``hasattr(data, 'globals')`` is False unless someone sets it."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


class MessagePassing(torch.nn.Module):
    """torch-geometric 2.6.1 behaviour used at graph_network.py:67-92:
    ``flow='source_to_target'`` (j = edge_index[0], i = edge_index[1]), default
    ``message(x_j) = x_j``, ``aggr='add'`` as zeros.scatter_add_, identity update.
    Extra keyword arguments to ``propagate`` (the reference passes ``edge_attr=``)
    are collected and, since ``message`` does not name them, ignored."""

    def __init__(self, aggr: str = "add"):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError(aggr)
        self.aggr = aggr

    def message(self, x_j):
        return x_j

    def propagate(self, edge_index, x=None, **kwargs):
        msg = self.message(x.index_select(0, edge_index[0]))
        out = x.new_zeros((x.shape[0], msg.shape[1]))
        return out.scatter_add_(0, edge_index[1].view(-1, 1).expand_as(msg), msg)


def _knn(x, y, k, *a, **kw):
    """torch-cluster 1.6.3 ``knn`` on CPU (see oracle/cpu_ref.py)."""
    from oracle import cpu_ref
    return cpu_ref.knn_extended(x, y, k)


def _install_standins() -> None:
    if "torch_geometric" in sys.modules and not getattr(sys.modules["torch_geometric"], "_cgnn_standin", False):
        return  # a real PyG is present: use it
    tg = types.ModuleType("torch_geometric")
    tg._cgnn_standin = True
    tgd = types.ModuleType("torch_geometric.data")
    tgn = types.ModuleType("torch_geometric.nn")
    tgd.Data = Data
    tgn.MessagePassing = MessagePassing
    tgn.knn_graph = None          # imported at data_utils.py:3, never called
    tg.data, tg.nn = tgd, tgn
    tc = types.ModuleType("torch_cluster")
    tc.knn = _knn
    ts = types.ModuleType("torch_scatter")   # imported at data_utils.py:6, never called
    sys.modules.update({"torch_geometric": tg, "torch_geometric.data": tgd, "torch_geometric.nn": tgn,
                        "torch_cluster": tc, "torch_scatter": ts})


def load():
    """Returns ``(graph_network, data_utils)`` = the reference's modules."""
    if not available():
        raise FileNotFoundError(f"reference checkout not found at {REFERENCE_DIR}")
    _install_standins()
    sys.dont_write_bytecode = True
    saved = {n: sys.modules.pop(n, None) for n in ("graph_network", "data_utils")}
    sys.path.insert(0, REFERENCE_DIR)
    try:
        gn = importlib.import_module("graph_network")
        du = importlib.import_module("data_utils")
    finally:
        sys.path.remove(REFERENCE_DIR)
        for n in ("graph_network", "data_utils"):
            sys.modules.pop(n, None)
            if saved[n] is not None:
                sys.modules[n] = saved[n]
    return gn, du

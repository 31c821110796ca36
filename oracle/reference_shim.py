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

The reference's drivers (``one_step_test.py``, ``render_rollout.py``,
``validation.py``) additionally import ``h5py`` (absent) for their snapshot I/O.
:func:`load_drivers` imports them behind :class:`H5File`, an in-memory stand-in
for ``h5py.File`` (a read-only mapping of dataset name -> numpy array registered
with :func:`register_h5`); it carries no arithmetic, so everything those drivers
compute -- window slicing, un-normalisation, integration, MSEs, the momentum
term, the autoregressive loop -- is the reference's own code.
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


class Batch(Data):
    """``torch_geometric.data.Batch.from_data_list`` as the reference uses it (train.py:247,
    validation.py:56): node-level tensors concatenated, ``edge_index`` offset per graph, ``batch`` = graph id
    of every node, ``num_graphs``."""

    @classmethod
    def from_data_list(cls, graphs):
        graphs = list(graphs)
        out = cls()
        offs, n = [], 0
        for g in graphs:
            offs.append(n)
            n += g.x.shape[0]
        for k in graphs[0].__dict__:
            vals = [getattr(g, k) for g in graphs]
            if k == "edge_index":
                out.edge_index = torch.cat([v + o for v, o in zip(vals, offs)], dim=1)
            elif all(torch.is_tensor(v) for v in vals):
                setattr(out, k, torch.cat(vals, dim=0))
            else:
                setattr(out, k, vals)
        out.batch = torch.cat([torch.full((g.x.shape[0],), i, dtype=torch.long) for i, g in enumerate(graphs)])
        out.num_graphs = len(graphs)
        return out


_H5_FILES = {}


def register_h5(path: str, datasets: dict) -> None:
    """Make ``h5py.File(path, 'r')`` (inside the reference's drivers) open this mapping of name -> numpy array."""
    _H5_FILES[path] = dict(datasets)


class H5File:
    """Read-only stand-in for ``h5py.File``: context manager + ``f[name]`` returning a numpy array (which has the
    ``.shape`` and slicing the reference uses at one_step_test.py:38-59)."""

    def __init__(self, path, mode="r", *a, **kw):
        if mode != "r" or path not in _H5_FILES:
            raise OSError(f"H5File stand-in: {path!r} is not registered (register_h5) or mode {mode!r} is not 'r'")
        self._d = _H5_FILES[path]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def __getitem__(self, name):
        return self._d[name]

    def __contains__(self, name):
        return name in self._d

    def keys(self):
        return self._d.keys()


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
    tgd.Batch = Batch
    tgn.MessagePassing = MessagePassing
    tgn.knn_graph = None          # imported at data_utils.py:3, never called
    tg.data, tg.nn = tgd, tgn
    tc = types.ModuleType("torch_cluster")
    tc.knn = _knn
    ts = types.ModuleType("torch_scatter")   # imported at data_utils.py:6, never called
    sys.modules.update({"torch_geometric": tg, "torch_geometric.data": tgd, "torch_geometric.nn": tgn,
                        "torch_cluster": tc, "torch_scatter": ts})
    if "h5py" not in sys.modules:
        h5 = types.ModuleType("h5py")
        h5._cgnn_standin = True
        h5.File = H5File
        sys.modules["h5py"] = h5


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


def load_drivers():
    """Returns ``(one_step_test, validation, render_rollout, generate_metadata)`` = the reference's driver modules, importing the
    reference's own ``graph_network`` / ``data_utils`` underneath (behind the stand-ins above)."""
    if not available():
        raise FileNotFoundError(f"reference checkout not found at {REFERENCE_DIR}")
    _install_standins()
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")     # render_rollout.py imports matplotlib.pyplot at module level
    names = ("graph_network", "data_utils", "one_step_test", "validation", "render_rollout", "generate_metadata")
    saved = {n: sys.modules.pop(n, None) for n in names}
    sys.path.insert(0, REFERENCE_DIR)
    try:
        mods = [importlib.import_module(n) for n in names]
    finally:
        sys.path.remove(REFERENCE_DIR)
        for n in names:
            sys.modules.pop(n, None)
            if saved[n] is not None:
                sys.modules[n] = saved[n]
    return mods[2], mods[3], mods[4], mods[5]

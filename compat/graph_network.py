"""``import graph_network`` resolves to the MI355X engine.

Put this directory on ``PYTHONPATH`` (ahead of the reference checkout) and the reference's drivers run unchanged:
``from graph_network import EncodeProcessDecode`` (train.py:16, one_step_test.py:9, render_rollout.py:10) gets the
engine's drop-in classes (same names, constructor signatures and ``state_dict`` keys as reference
graph_network.py:15-187).
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from cosmology_gnn_simulation_amd.graph_network import (  # noqa: E402,F401
    EncodeProcessDecode, GraphIndependent, InteractionNetwork, build_mlp)

__all__ = ["build_mlp", "GraphIndependent", "InteractionNetwork", "EncodeProcessDecode"]

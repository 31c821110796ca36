"""``import data_utils`` resolves to the MI355X engine (see compat/graph_network.py): ``from data_utils import
preprocess`` (train.py:15, validation.py:3, one_step_test.py:10, render_rollout.py:11) builds the periodic k-NN
graph on the GPU with the reference's signature and return attributes (reference data_utils.py:72-228)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from cosmology_gnn_simulation_amd.data_utils import (  # noqa: E402,F401
    extend_positions_torch, generate_position_noise, generate_temperature_noise, knn_graph_periodic, preprocess)

__all__ = ["extend_positions_torch", "generate_position_noise", "generate_temperature_noise", "preprocess"]

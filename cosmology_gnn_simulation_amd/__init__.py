"""MI355X-native Interaction-Network message-passing engine.

Drop-in for the hot path of mattpan-peregrinus/Cosmology_GNN_Simulation:
``graph_network`` (EncodeProcessDecode / InteractionNetwork / GraphIndependent /
build_mlp) and ``data_utils`` (preprocess, periodic k-NN graph build), running on
hand-written gfx950 HIP kernels behind the C ABI declared in ``include/cgnn.h``.
There is no CPU fallback: compute entry points raise if ``libcgnn_hip.so`` or a
HIP device is missing.
"""
__version__ = "0.1.0"

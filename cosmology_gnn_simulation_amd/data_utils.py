"""Drop-in for the reference's ``data_utils`` module: window -> graph.

``preprocess`` keeps the reference signature and returns a ``Data`` with the same
attributes (reference data_utils.py:72-228) but builds the periodic k-NN graph and
its edge features on the GPU (``cgnn_knn_periodic``) instead of running
``torch_cluster.knn`` over a 27x ghost-extended copy on one CPU thread
(:148-164), and the node features in one kernel.  Only the training targets
(:166-214, used by ``train.py`` alone) stay torch element-wise expressions.

Faithfulness notes
* random-walk noise is drawn on the CPU with the reference's exact call sequence
  (``randn_like`` on ``[N, W-1, 3]`` then ``[N, W-1, 1]``, even when
  ``noise_std == 0``, :47,:63), so seeded runs see the same noise and the same RNG
  stream afterwards;
* node features (wrap, velocities, normalisation: :91-145) come from one HIP kernel
  (``cgnn_window_features``) with the reference's float32 operation order;
* edge displacements use the un-shifted sender (not minimum-image), as :151-164 do;
* the graph is receiver-sorted with ``num_neighbors`` edges per receiver, the
  receiver itself first (distance 0).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from ._lib import CgnnError
from .graph import Data

__all__ = ["extend_positions_torch", "generate_position_noise", "generate_temperature_noise", "preprocess",
           "knn_graph_periodic"]


def _default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise CgnnError("no HIP device visible: cosmology_gnn_simulation_amd builds graphs on the GPU only")
    return torch.device("cuda", torch.cuda.current_device())


def extend_positions_torch(positions: torch.Tensor, box_size):
    """The 27-image extension of reference data_utils.py:9-33, kept for API
    compatibility (the engine itself never materialises it).  Returns
    ``(extended [N*3^d, d], mapping [N*3^d])``."""
    n, d = positions.shape
    if isinstance(box_size, list):
        box_size = float(box_size[0])
    axis = torch.tensor([-box_size, 0.0, box_size], device=positions.device, dtype=torch.float32)
    shifts = torch.cartesian_prod(*([axis] * d)).reshape(-1, d)
    extended = (positions.unsqueeze(0) + shifts.unsqueeze(1)).reshape(-1, d)
    mapping = torch.arange(n, device=positions.device).repeat(shifts.shape[0])
    return extended, mapping


def _wrap_displacement(d: torch.Tensor, box_size: float) -> torch.Tensor:
    half = box_size / 2
    d = torch.where(d < -1 * half, d + box_size, d)
    return torch.where(d > half, d - box_size, d)


def generate_position_noise(position_seq: torch.Tensor, noise_std: float, box_size: float, dt: float) -> torch.Tensor:
    """Random-walk position noise for ``[N, W, 3]`` (reference data_utils.py:36-54)."""
    p = position_seq.float()
    vel = _wrap_displacement(p[:, 1:] - p[:, :-1], box_size) / dt
    steps = vel.size(1)
    walk = (torch.randn_like(vel, dtype=torch.float32) * (noise_std / (steps ** 0.5))).cumsum(dim=1)
    pos_noise = walk.cumsum(dim=1) * dt
    return torch.cat((torch.zeros_like(pos_noise)[:, 0:1], pos_noise), dim=1)


def generate_temperature_noise(temperature_seq: torch.Tensor, noise_std: float, temp_rate_std, dt: float):
    """Random-walk temperature noise for ``[N, W, 1]`` (reference data_utils.py:57-70)."""
    t = temperature_seq.float()
    rate = (t[:, 1:] - t[:, :-1]) / dt
    steps = rate.size(1)
    walk = (torch.randn_like(rate, dtype=torch.float32) * (noise_std * temp_rate_std / (steps ** 0.5))).cumsum(dim=1)
    t_noise = walk.cumsum(dim=1) * dt
    return torch.cat((torch.zeros_like(t_noise)[:, 0:1], t_noise), dim=1)


def knn_graph_periodic(pos: torch.Tensor, box_size: float, k: int, want_order: bool = False):
    """Periodic k-NN on the device.  Returns ``(edge_index int64 [2, N*k],
    edge_attr [N*k, 4], senders int32, order|None)``."""
    senders, edge_attr, order = ops.knn_periodic(pos, box_size, k, None, True, want_order)
    n = pos.shape[0]
    receivers = torch.arange(n, device=pos.device, dtype=torch.int64).repeat_interleave(k)
    edge_index = torch.stack([senders.to(torch.int64), receivers], dim=0)
    return edge_index, edge_attr, senders, order


def _meta(metadata: dict, key: str, device) -> torch.Tensor:
    return torch.tensor(metadata[key], dtype=torch.float32, device=device)


def _draw_reference_noise(pos_seq: torch.Tensor, tmp_seq: torch.Tensor, noise_std: float, temp_rate_std, dt: float,
                          box_size: float):
    """The reference's two RNG draws (data_utils.py:47,:63) on the CPU generator.  ``randn_like`` follows the
    memory layout of its argument, so the draw order depends on the strides of the (permuted) windows; they are
    reproduced on zero-filled host tensors with the inputs' shapes and strides (the noise does not depend on the
    values), which spares device-resident windows a copy to the host."""
    def host_like(t):
        return t if not t.is_cuda else torch.empty_strided(t.shape, t.stride(), dtype=torch.float32).zero_()
    pos_noise = generate_position_noise(host_like(pos_seq), noise_std, box_size, dt)
    tmp_noise = generate_temperature_noise(host_like(tmp_seq), noise_std, temp_rate_std, dt)
    return pos_noise, tmp_noise


def preprocess(position_seq, temperature_seq, metadata, target_position=None, target_temperature=None,
               noise_std=0.0, num_neighbors=16, dt=None, box_size=None, device: Optional[torch.device] = None,
               reference_rng: bool = True, check_bounds: bool = True):
    """Window ``[W, N, 3]`` / ``[W, N, 1]`` -> graph (reference data_utils.py:72-228).

    ``device`` (extension) selects the GPU; by default the inputs' device if they
    are already on one, else the current HIP device.  All returned tensors live
    there, so the caller's ``graph.to(device)`` is a no-op.  ``reference_rng=False``
    (extension, only honoured when ``noise_std == 0``) skips the two CPU random draws
    the reference makes even for zero noise; results are identical, only the global
    RNG stream is left untouched (used by the on-device rollout).  ``check_bounds=False`` (extension) drops the
    reference's sender-index assertion (:158-159), which costs one device-to-host synchronisation per call."""
    dt = float(dt)
    box_size = float(box_size)
    if device is None:
        device = position_seq.device if position_seq.is_cuda else _default_device()
    device = torch.device(device)

    pos_seq = position_seq.float().permute(1, 0, 2)                       # [N, W, 3]
    tmp_seq = temperature_seq.float()
    if tmp_seq.shape[0] == pos_seq.shape[1] and tmp_seq.shape[1] == pos_seq.shape[0]:
        tmp_seq = tmp_seq.permute(1, 0, 2)                                # [N, W, 1]
    if target_position is not None:
        target_position = target_position.float()

    # --- noise: drawn where the reference draws it (CPU RNG stream), then moved ---
    pos_noise_cpu = tmp_noise_cpu = None
    if reference_rng or noise_std != 0.0:
        trs_cpu = torch.tensor(metadata["temp_rate_std"], dtype=torch.float32)
        pos_noise_cpu, tmp_noise_cpu = _draw_reference_noise(pos_seq, tmp_seq, noise_std, trs_cpu, dt, box_size)

    pos_seq = pos_seq.to(device)                                          # [N, W, 3] view of the [W, N, 3] window
    tmp_seq = tmp_seq.to(device)
    pos_noise = tmp_noise = None
    if noise_std != 0.0:
        pos_noise, tmp_noise = pos_noise_cpu.to(device), tmp_noise_cpu.to(device)

    # --- node features and the wrapped last frame in one kernel (cgnn_window_features) ---
    node_features, recent_position = ops.window_features(pos_seq.permute(1, 0, 2), tmp_seq.permute(1, 0, 2), metadata, dt,
                                                         box_size, pos_noise, tmp_noise)

    velocity_seq = recent_temperature = None
    if target_position is not None or target_temperature is not None:
        # training targets (reference :166-214): the same element-wise expressions, on the device
        wp = torch.remainder(pos_seq + pos_noise, box_size) if noise_std != 0.0 else torch.remainder(pos_seq, box_size)
        velocity_seq = _wrap_displacement(wp[:, 1:] - wp[:, :-1], box_size) / dt
        recent_temperature = (tmp_seq + tmp_noise if noise_std != 0.0 else tmp_seq)[:, -1]

    if target_temperature is not None:
        target_temperature = target_temperature.float()
        if target_temperature.dim() == 3:
            target_temperature = target_temperature.permute(1, 0, 2).squeeze(1)
        elif target_temperature.dim() == 2 and target_temperature.shape[1] != 1:
            target_temperature = target_temperature.reshape(-1, 1)
        if target_temperature.shape != recent_temperature.shape and \
                target_temperature.numel() == recent_temperature.numel():
            target_temperature = target_temperature.reshape(recent_temperature.shape)

    # --- periodic k-NN graph + edge features on the device ---
    edge_index, edge_attr, senders, order = knn_graph_periodic(recent_position, box_size, int(num_neighbors),
                                                                want_order=True)
    n = recent_position.shape[0]
    if check_bounds:    # reference :158-159 (a host round trip: the on-device rollout turns it off)
        assert int(senders.max()) < n, f"Max sender index {int(senders.max())} >= {n}"

    acceleration = None
    temp_rate = None
    if target_position is not None:
        tp = target_position
        if tp.dim() == 3:
            tp = tp.permute(1, 0, 2).squeeze(1)
        elif tp.dim() == 2 and tp.shape[0] != n:
            tp = tp.reshape(-1, 3)
        if noise_std != 0.0:
            tp += pos_noise_cpu[:, -1].to(tp.device)   # in place on the caller's tensor, like the reference (:182)
        tp = tp.to(device)
        next_velocity = _wrap_displacement(tp - recent_position, box_size) / dt
        acceleration = (next_velocity - velocity_seq[:, -1]) / dt
        acceleration = (acceleration - _meta(metadata, "acc_mean", device)) / _meta(metadata, "acc_std", device)
    if target_temperature is not None:
        tt = target_temperature
        if tt.dim() == 3:
            tt = tt.squeeze(1)
        if noise_std != 0.0:
            tt += tmp_noise_cpu[:, -1].to(tt.device)   # in place, like the reference (:206)
        tt = tt.to(device)
        temp_rate = (tt - recent_temperature) / dt
        temp_rate = (temp_rate - _meta(metadata, "temp_rate_mean", device)) / _meta(metadata, "temp_rate_std", device)

    graph = Data(
        x=node_features.float(),
        edge_index=edge_index,
        edge_attr=edge_attr,
        y_acc=acceleration.float() if acceleration is not None else None,
        y_temp_rate=temp_rate.float() if temp_rate is not None else None,
        pos=recent_position,
        dt=torch.full((1,), dt, dtype=torch.float32, device=device),            # (a fill kernel: torch.tensor([dt],
        box_size=torch.full((1,), box_size, dtype=torch.float32, device=device),  # device=...) would synchronise)
    )
    graph._cgnn_fixed_k = int(num_neighbors)
    graph._cgnn_fixed_k_for = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape))
    graph._cgnn_order = order          # spatial (cell-sorted) particle order: a locality hint for the engine
    return graph

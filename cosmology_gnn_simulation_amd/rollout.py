"""Autoregressive rollout kept on the device: the counterpart of the reference's
``render_rollout.rollout`` (render_rollout.py:26-90).

Per step the reference rebuilds the k-NN graph on the CPU, moves it to the GPU, runs
the model, copies the predictions back, integrates on the CPU and grows the
trajectory with ``torch.cat`` (O(T^2) copies, :84-85).  Here the trajectory is a
pre-allocated device buffer and graph build -> forward -> integrate -> wrap never
leave the GPU: after the first step (weight packing) a step contains no host
synchronisation at all (tests run it under ``torch.cuda.set_sync_debug_mode("error")``);
the number of neighbours is a parameter (hard-coded 16 at :49).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .data_utils import preprocess
from .one_step import integrate_one_step, integration_constants


def rollout(model, data: Dict[str, torch.Tensor], metadata: dict, noise_std: float, dt: float, box_size: float,
            window_size: int = 6, num_neighbors: int = 16, num_steps: Optional[int] = None,
            device: Optional[torch.device] = None, reference_rng: bool = False) -> Dict[str, torch.Tensor]:
    """Same arguments and return value as the reference (``Coordinates [T, N, 3]``,
    ``InternalEnergy [T, N, 1]``, the first ``window_size`` frames copied from ``data``).  ``noise_std`` is
    accepted for signature compatibility only: the reference builds every rollout graph with ``noise_std=0.0``
    whatever the argument says (render_rollout.py:44-52, "Build a graph with no noise for rollout")."""
    del noise_std
    if device is None:
        device = next(model.parameters()).device
    device = torch.device(device)
    model.eval()
    coords = data["Coordinates"]
    energy = data["InternalEnergy"]
    if energy.dim() == 2:
        energy = energy.unsqueeze(-1)
    total_time = coords.size(0) if num_steps is None else window_size + num_steps
    n = coords.size(1)
    meta = dict(metadata)
    meta["dt"], meta["box_size"] = dt, box_size
    pos_traj = torch.empty((total_time, n, 3), dtype=torch.float32, device=device)
    tmp_traj = torch.empty((total_time, n, 1), dtype=torch.float32, device=device)
    pos_traj[:window_size] = coords[:window_size].to(device).float()
    tmp_traj[:window_size] = energy[:window_size].to(device).float()
    consts = integration_constants(meta, device)
    with torch.no_grad():
        for t in range(window_size, total_time):
            win_p = pos_traj[t - window_size:t]                      # [W, N, 3] views, no copies
            win_t = tmp_traj[t - window_size:t]
            graph = preprocess(position_seq=win_p, temperature_seq=win_t, metadata=meta, noise_std=0.0,
                               num_neighbors=num_neighbors, box_size=box_size, dt=dt, device=device,
                               reference_rng=reference_rng, check_bounds=False)
            pred = model(graph)
            new_p, new_t = integrate_one_step(pred["acceleration"], pred["temp_rate"], win_p, win_t, meta, consts)
            pos_traj[t] = new_p
            tmp_traj[t] = new_t
    return {"Coordinates": pos_traj, "InternalEnergy": tmp_traj}


def calculate_errors(rollout_data: Dict[str, torch.Tensor], ground_truth: Dict[str, torch.Tensor]) -> dict:
    """Per-frame MSE of positions and temperatures (reference render_rollout.py:92-120)."""
    pc, tc = rollout_data["Coordinates"], ground_truth["Coordinates"].to(rollout_data["Coordinates"].device)
    pt = rollout_data["InternalEnergy"].squeeze()
    tt = ground_truth["InternalEnergy"].to(pt.device).squeeze()
    frames = min(len(pc), len(tc))
    pos = [torch.mean((pc[t] - tc[t]) ** 2).item() for t in range(frames)]
    frames_t = min(len(pt), len(tt))
    tmp = [torch.mean((pt[t] - tt[t]) ** 2).item() for t in range(frames_t)]
    return {"position_errors": pos, "temperature_errors": tmp,
            "mean_position_error": sum(pos) / len(pos) if pos else None,
            "mean_temperature_error": sum(tmp) / len(tmp) if tmp else None}

"""One-step evaluation harness: the counterpart of the reference's
``one_step_test.validate_one_step`` (one_step_test.py:26-124).

Same arithmetic (window -> ``preprocess`` -> forward -> un-normalise ->
semi-implicit Euler -> periodic wrap -> MSE) and the same result dictionary, but
the snapshot is an ``.npz`` / dict with the HDF5 dataset names (h5py is not in the
image) and everything after the window slice stays on the GPU.
"""
from __future__ import annotations

import argparse
import json
from typing import Dict, Optional, Sequence, Union

import numpy as np
import torch

from . import synthetic
from .data_utils import preprocess
from .graph_network import EncodeProcessDecode


def load_model(model_path: str, args) -> EncodeProcessDecode:
    """reference one_step_test.py:12-24 (state_dict is loaded before the first forward)."""
    model = EncodeProcessDecode(latent_size=args.latent_size, mlp_hidden_size=args.mlp_hidden_size,
                                mlp_num_hidden_layers=args.mlp_num_hidden_layers,
                                num_message_passing_steps=args.num_message_passing_steps,
                                output_size=args.output_size)
    model.load_state_dict(torch.load(model_path, map_location=args.device))
    model = model.to(args.device)
    model.eval()
    return model


def integration_constants(metadata: dict, device) -> dict:
    """The four un-normalisation statistics as device tensors, made ONCE: building them inside the step would cost a
    blocking host-to-device copy (= a stream synchronisation) per statistic and step."""
    out = {}
    for key in ("acc_std", "acc_mean", "temp_rate_std", "temp_rate_mean"):
        host = torch.tensor(metadata[key], dtype=torch.float32)
        if torch.device(device).type == "cuda":     # pinned + non_blocking: no stream synchronisation
            host = host.pin_memory()
        out[key] = host.to(device, non_blocking=True)
    return out


def integrate_one_step(acc_pred: torch.Tensor, temp_rate_pred: torch.Tensor, coords_seq: torch.Tensor,
                       temp_seq: torch.Tensor, metadata: dict, consts: dict = None):
    """Un-normalise the predictions and advance one step (reference one_step_test.py:84-105).  ``consts``:
    :func:`integration_constants` of the same metadata (made here when omitted)."""
    dev = acc_pred.device
    dt, box = metadata["dt"], metadata["box_size"]
    if consts is None:
        consts = integration_constants(metadata, dev)
    acc = acc_pred * consts["acc_std"] + consts["acc_mean"]
    rate = temp_rate_pred * consts["temp_rate_std"] + consts["temp_rate_mean"]
    recent_p = coords_seq[-1].to(dev)
    recent_v = (recent_p - coords_seq[-2].to(dev)) / dt
    new_v = recent_v + acc * dt
    new_p = torch.remainder(recent_p + new_v * dt, box)
    new_t = temp_seq[-1].to(dev) + rate * dt
    return new_p, new_t


def validate_one_step(model, data: Union[str, Dict[str, torch.Tensor]], metadata: dict, window_size: int, device,
                      num_neighbors: int = 16, num_timesteps: int = 10, noise_std: float = 0.0,
                      start_indices: Optional[Sequence[int]] = None) -> dict:
    """reference one_step_test.py:26-124.  ``data`` is a snapshot dict or an ``.npz``
    path with ``Coordinates [T, N, 3]`` and ``InternalEnergy [T, N(,1)]``."""
    model.eval()
    snap = synthetic.load_snapshot(data) if isinstance(data, str) else data
    coords_all, energy_all = snap["Coordinates"], snap["InternalEnergy"]
    total_frames = coords_all.shape[0]
    max_start_idx = total_frames - window_size - 1
    if start_indices is None:
        if max_start_idx < num_timesteps:
            num_timesteps = max_start_idx
        start_indices = sorted(np.random.choice(max_start_idx, size=num_timesteps, replace=False))
    dt, box_size = metadata["dt"], metadata["box_size"]
    position_errors, temperature_errors, tested = [], [], []
    for start_idx in start_indices:
        coords_seq = coords_all[start_idx:start_idx + window_size].float()
        next_coords = coords_all[start_idx + window_size].float()
        temp_seq = energy_all[start_idx:start_idx + window_size].float()
        next_temp = energy_all[start_idx + window_size].float()
        if temp_seq.dim() == 2:
            temp_seq = temp_seq.unsqueeze(-1)
        if next_temp.dim() == 1:
            next_temp = next_temp.unsqueeze(-1)
        graph = preprocess(position_seq=coords_seq, temperature_seq=temp_seq, metadata=metadata,
                           noise_std=noise_std, num_neighbors=num_neighbors, box_size=box_size, dt=dt,
                           device=device)
        with torch.no_grad():
            pred = model(graph)
        new_p, new_t = integrate_one_step(pred["acceleration"], pred["temp_rate"], coords_seq, temp_seq, metadata)
        position_errors.append(torch.mean((new_p - next_coords.to(new_p.device)) ** 2).item())
        temperature_errors.append(torch.mean((new_t - next_temp.to(new_t.device)) ** 2).item())
        tested.append(int(start_idx) + window_size)
    return {
        "position_error": float(np.mean(position_errors)) if position_errors else float("nan"),
        "temperature_error": float(np.mean(temperature_errors)) if temperature_errors else float("nan"),
        "position_errors": position_errors,
        "temperature_errors": temperature_errors,
        "tested_timesteps": tested,
    }


def main(argv=None) -> None:
    """Same flags as the reference CLI (one_step_test.py:126-140); ``--test_data`` is an ``.npz``."""
    p = argparse.ArgumentParser(description="Validate one-step predictions (MI355X engine)")
    p.add_argument("--model_path", type=str, required=True)
    p.add_argument("--test_data", type=str, required=True)
    p.add_argument("--metadata_path", type=str, required=True)
    p.add_argument("--window_size", type=int, default=5)
    p.add_argument("--num_neighbors", type=int, default=16)
    p.add_argument("--num_timesteps", type=int, default=10)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--latent_size", type=int, default=128)
    p.add_argument("--mlp_hidden_size", type=int, default=128)
    p.add_argument("--mlp_num_hidden_layers", type=int, default=2)
    p.add_argument("--num_message_passing_steps", type=int, default=10)
    p.add_argument("--output_size", type=int, default=3)
    args = p.parse_args(argv)
    with open(args.metadata_path) as f:
        metadata = json.load(f)
    model = load_model(args.model_path, args)
    res = validate_one_step(model, args.test_data, metadata, args.window_size, args.device, args.num_neighbors,
                            args.num_timesteps)
    print(f"Number of timesteps tested: {len(res['position_errors'])}")
    print(f"Average position MSE: {res['position_error']:.6e}")
    print(f"Average temperature MSE: {res['temperature_error']:.6e}")


if __name__ == "__main__":
    main()

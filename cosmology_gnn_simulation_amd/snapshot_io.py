"""Snapshot / metadata I/O at the edges of the hot path.

The reference reads HDF5 files with datasets ``Coordinates [T, N, 3]``, ``InternalEnergy [T, N(, 1)]``,
``Velocities``, ``HydroAcceleration`` and scalars ``BoxSize``, ``TimeStep`` (README.md:31,
one_step_test.py:54-59) and derives the normalisation statistics with generate_metadata.py:6-48.  h5py is not
part of this image, so ``.npz`` files with the same dataset names are the native container; ``.hdf5`` / ``.h5``
are read when h5py happens to be importable.  This is offline, host-side code: nothing here is on the timed path.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import numpy as np
import torch

FIELDS = ("Coordinates", "InternalEnergy", "Velocities", "HydroAcceleration", "BoxSize", "TimeStep")


def read_snapshot(path: str) -> Dict[str, torch.Tensor]:
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        z = np.load(path)
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    if ext in (".hdf5", ".h5"):
        try:
            import h5py
        except ImportError as exc:  # pragma: no cover - h5py is absent from the build image
            raise ImportError("reading HDF5 snapshots needs h5py; convert to .npz with the same dataset names") from exc
        with h5py.File(path, "r") as f:
            return {k: torch.from_numpy(np.asarray(f[k][...])) for k in f.keys()}
    raise ValueError(f"unknown snapshot format {ext!r} (use .npz, .hdf5 or .h5)")


def write_snapshot(path: str, snap: Dict[str, torch.Tensor]) -> None:
    np.savez(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in snap.items()})


def generate_metadata(snap: Dict[str, torch.Tensor], output_path: Optional[str] = None) -> dict:
    """The ten statistics of reference generate_metadata.py:15-43 (same keys, same list-vs-scalar shapes).
    ``Velocities`` / ``HydroAcceleration`` default to finite differences of ``Coordinates`` when absent
    (rollout_conversion.py:56-92 builds them that way)."""
    def arr(name):
        return np.asarray(snap[name].cpu() if torch.is_tensor(snap[name]) else snap[name], dtype=np.float64)
    coords, energy = arr("Coordinates"), arr("InternalEnergy")
    box, dt = float(arr("BoxSize")), float(arr("TimeStep"))
    if "Velocities" in snap:
        vel = arr("Velocities")
    else:
        d = coords[1:] - coords[:-1]
        d = np.where(d < -box / 2, d + box, d)
        d = np.where(d > box / 2, d - box, d)
        vel = d / dt
    acc = arr("HydroAcceleration") if "HydroAcceleration" in snap else (vel[1:] - vel[:-1]) / dt
    rate = (energy[1:] - energy[:-1]) / dt
    meta = {
        "temp_mean": np.mean(energy, axis=(0, 1)).tolist(), "temp_std": np.std(energy, axis=(0, 1)).tolist(),
        "temp_rate_mean": np.mean(rate, axis=(0, 1)).tolist(), "temp_rate_std": np.std(rate, axis=(0, 1)).tolist(),
        "vel_mean": float(np.mean(np.mean(vel, axis=(0, 1)))), "vel_std": float(np.mean(np.std(vel, axis=(0, 1)))),
        "acc_mean": float(np.mean(np.mean(acc, axis=(0, 1)))), "acc_std": float(np.mean(np.std(acc, axis=(0, 1)))),
        "box_size": box, "dt": dt,
    }
    if output_path:
        with open(output_path, "w") as f:
            json.dump(meta, f, indent=4)
    return meta

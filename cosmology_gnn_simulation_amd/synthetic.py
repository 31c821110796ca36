"""Synthetic particle boxes, normalisation metadata and deterministic weights.

There is no network and no dataset on the build or GPU machines, so every test,
the smoke run and ``bench.py`` use inputs generated here.
The snapshot container mirrors the HDF5 dataset names the reference reads
(``Coordinates``, ``InternalEnergy``, ``BoxSize``, ``TimeStep``;
reference one_step_test.py:54-59, README.md:31) but is a ``.npz``: h5py is not in
the image.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict

import numpy as np
import torch

METADATA_KEYS = ("vel_mean", "vel_std", "acc_mean", "acc_std", "temp_mean", "temp_std",
                 "temp_rate_mean", "temp_rate_std", "dt", "box_size")  # reference generate_metadata.py:32-43


def make_metadata(box_size: float = 1.0, dt: float = 0.01) -> dict:
    return dict(vel_mean=0.0, vel_std=1.0, acc_mean=0.0, acc_std=1.0, temp_mean=1.0, temp_std=0.5,
                temp_rate_mean=0.0, temp_rate_std=2.0, dt=dt, box_size=box_size)


def make_snapshot(num_particles: int, window: int = 5, box_size: float = 1.0, dt: float = 0.01,
                  seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Uniform random box: ``W+1`` frames of positions ``[W+1, N, 3]`` and internal
    energy ``[W+1, N, 1]`` (the last frame is the one-step target)."""
    g = torch.Generator().manual_seed(seed)
    p0 = torch.rand(num_particles, 3, generator=g, dtype=torch.float32) * box_size
    v = torch.randn(num_particles, 3, generator=g, dtype=torch.float32) * 0.05
    t = torch.arange(window + 1, dtype=torch.float32).view(-1, 1, 1)
    coords = torch.remainder(p0.unsqueeze(0) + v.unsqueeze(0) * (dt * t), box_size)
    # remainder can round up to exactly box_size for tiny negatives; fold those back.
    coords = torch.where(coords >= box_size, coords - box_size, coords)
    energy = 1.0 + 0.1 * torch.randn(window + 1, num_particles, 1, generator=g, dtype=torch.float32).cumsum(dim=0)
    return dict(Coordinates=coords, InternalEnergy=energy,
                BoxSize=torch.tensor(box_size), TimeStep=torch.tensor(dt))


class LazySnapshot:
    """The box of :func:`make_snapshot` (same seed -> the same numbers, bit for bit) without materialising the
    ``[W+1, N, 3]`` trajectories: the random draws are kept (they have to be made in full to keep the generator's
    order), single frames and the window of a SUBSET of particles are built on demand.  For the multi-GPU bench
    (``dist.build_synthetic_shard``): every rank needs one global frame of positions for the ownership and the
    neighbour search, but the feature window of its own particles only -- the host work and the upload per rank then
    stay near N + N / world instead of 6 N as the job grows."""

    def __init__(self, num_particles: int, window: int = 5, box_size: float = 1.0, dt: float = 0.01, seed: int = 1234):
        g = torch.Generator().manual_seed(seed)
        self.window, self.box_size, self.dt = window, box_size, dt
        self.p0 = torch.rand(num_particles, 3, generator=g, dtype=torch.float32) * box_size
        self.v = torch.randn(num_particles, 3, generator=g, dtype=torch.float32) * 0.05
        self.noise = torch.randn(window + 1, num_particles, 1, generator=g, dtype=torch.float32)

    def _coords(self, p0: torch.Tensor, v: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        coords = torch.remainder(p0.unsqueeze(0) + v.unsqueeze(0) * (self.dt * t), self.box_size)
        return torch.where(coords >= self.box_size, coords - self.box_size, coords)

    def frame(self, index: int) -> torch.Tensor:
        """``Coordinates[index]`` of all particles, ``[N, 3]``."""
        t = torch.tensor([float(index)], dtype=torch.float32).view(-1, 1, 1)
        return self._coords(self.p0, self.v, t)[0]

    def window_of(self, ids: torch.Tensor):
        """``(Coordinates[:, ids], InternalEnergy[:, ids])``, all ``W + 1`` frames."""
        ids = ids.cpu().long()
        t = torch.arange(self.window + 1, dtype=torch.float32).view(-1, 1, 1)
        coords = self._coords(self.p0[ids], self.v[ids], t)
        energy = 1.0 + 0.1 * self.noise[:, ids].cumsum(dim=0)
        return coords, energy


def save_snapshot(path: str, snap: Dict[str, torch.Tensor]) -> None:
    np.savez(path, **{k: v.numpy() for k, v in snap.items()})


def load_snapshot(path: str) -> Dict[str, torch.Tensor]:
    z = np.load(path)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def state_dict_shapes(latent_size: int, mlp_hidden_size: int, mlp_num_hidden_layers: int,
                      num_message_passing_steps: int, output_size: int,
                      node_in: int = 17, edge_in: int = 4) -> "OrderedDict[str, tuple]":
    """Key -> shape of the reference model's ``state_dict`` (reference
    graph_network.py:24-32,133-152), in module order."""
    D, H, nh = latent_size, mlp_hidden_size, mlp_num_hidden_layers
    out: "OrderedDict[str, tuple]" = OrderedDict()

    def add_mlp(prefix: str, fan_in: int, fan_out: int):
        i_prev = fan_in
        for i in range(nh):
            out[f"{prefix}.{2 * i}.weight"] = (H, i_prev)
            out[f"{prefix}.{2 * i}.bias"] = (H,)
            i_prev = H
        out[f"{prefix}.{2 * nh}.weight"] = (fan_out, i_prev)
        out[f"{prefix}.{2 * nh}.bias"] = (fan_out,)

    def add_mlp_ln(prefix: str, fan_in: int):
        add_mlp(f"{prefix}.0", fan_in, D)
        out[f"{prefix}.1.weight"] = (D,)
        out[f"{prefix}.1.bias"] = (D,)

    add_mlp_ln("encoder.node_model", node_in)
    add_mlp_ln("encoder.edge_model", edge_in)
    for i in range(num_message_passing_steps):
        # ModuleList entries register node_model before edge_model (InteractionNetwork.__init__)
        add_mlp_ln(f"processor.{i}.node_model", 2 * D)
        add_mlp_ln(f"processor.{i}.edge_model", 3 * D)
    add_mlp("decoder_acc", D, output_size)
    add_mlp("decoder_temp_rate", D, 1)
    return out


def make_state_dict(latent_size: int = 128, mlp_hidden_size: int = 128, mlp_num_hidden_layers: int = 2,
                    num_message_passing_steps: int = 10, output_size: int = 3,
                    node_in: int = 17, edge_in: int = 4, seed: int = 7) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic weights: keys filled in sorted order from ``Generator(seed)``;
    Linear weight/bias ``U(-1/sqrt(fan_in), 1/sqrt(fan_in))`` (torch's default
    bound), LayerNorm weight ``1 + 0.1 N(0,1)``, bias ``0.1 N(0,1)`` so the affine
    part is exercised."""
    shapes = state_dict_shapes(latent_size, mlp_hidden_size, mlp_num_hidden_layers,
                               num_message_passing_steps, output_size, node_in, edge_in)
    g = torch.Generator().manual_seed(seed)
    vals = {}
    for key in sorted(shapes):
        shp = shapes[key]
        parts = key.split(".")
        is_ln = parts[-2] == "1" and parts[-3] in ("node_model", "edge_model")
        if is_ln:
            n = torch.randn(shp, generator=g, dtype=torch.float32)
            vals[key] = 1.0 + 0.1 * n if parts[-1] == "weight" else 0.1 * n
        else:
            wkey = key[: -len(parts[-1])] + "weight"
            bound = 1.0 / math.sqrt(shapes[wkey][1])
            vals[key] = (torch.rand(shp, generator=g, dtype=torch.float32) * 2.0 - 1.0) * bound
    return OrderedDict((k, vals[k]) for k in shapes)

"""Loss terms of the reference's training loop that sit on the hot path."""
from __future__ import annotations

import torch

from . import ops


def momentum_conservation_loss(accelerations: torch.Tensor, batch_graph, dt: float, momentum_weight: float):
    """``w / B * sum_g || sum_{n in g} acc[n] * dt ||^2`` (reference train.py:107-118,
    duplicated at validation.py:5-16).  The per-graph column sums run in one
    segmented float64 reduction on the device instead of a Python loop over
    boolean masks.  Returns a 0-d float32 tensor on the accelerations' device."""
    num_graphs = int(getattr(batch_graph, "num_graphs", 1) or 1)
    batch = getattr(batch_graph, "batch", None)
    sums = ops.segment_colsum(accelerations.detach(), batch, num_graphs)          # [B, 3] float64
    total = torch.sum((sums * float(dt)) ** 2)
    return (momentum_weight * total / num_graphs).to(torch.float32)

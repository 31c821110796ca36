"""Loss terms of the reference's training loop that sit on the hot path."""
from __future__ import annotations

import torch

from . import ops


class _SegmentColsum(torch.autograd.Function):
    """``sums[g] = sum_{n in graph g} acc[n]`` in float64 (``cgnn_segment_colsum``); the backward broadcasts each
    graph's gradient row back to its particles."""

    @staticmethod
    def forward(ctx, acc, batch, num_graphs):
        ctx.batch, ctx.n = batch, acc.shape[0]
        return ops.segment_colsum(acc, batch, num_graphs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_sums):
        d = d_sums.to(torch.float32)
        if ctx.batch is None:
            return d[0].expand(ctx.n, -1).contiguous(), None, None
        return d[ctx.batch.long()], None, None


def momentum_conservation_loss(accelerations: torch.Tensor, batch_graph, dt: float, momentum_weight: float):
    """``w / B * sum_g || sum_{n in g} acc[n] * dt ||^2`` (reference train.py:107-118,
    duplicated at validation.py:5-16).  The per-graph column sums run in one
    segmented float64 reduction on the device instead of a Python loop over
    boolean masks.  Returns a 0-d float32 tensor on the accelerations' device; differentiable with respect to
    ``accelerations``."""
    num_graphs = int(getattr(batch_graph, "num_graphs", 1) or 1)
    batch = getattr(batch_graph, "batch", None)
    sums = _SegmentColsum.apply(accelerations, batch, num_graphs)                 # [B, 3] float64
    total = torch.sum((sums * float(dt)) ** 2)
    return (momentum_weight * total / num_graphs).to(torch.float32)

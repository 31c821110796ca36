"""CPU restatement of the bf16 edge stream's arithmetic (BASELINE cfg3-5: "bf16 MLP, f32 accumulation").

TEST INFRASTRUCTURE ONLY, like ``oracle/cpu_ref.py``: nothing under ``cosmology_gnn_simulation_amd/`` may import this
module; importers are ``tests/`` only.  It is the yardstick for the one-launch edge-stream kernels
(``cgnn_edge_stream_run``, ``cgnn_edge_stream_run_w8``), whose f32-accurate counterpart is ``cpu_ref``.

What it restates (citations into the read-only reference checkout):

* ``graph_network.py:57``       edge encoder  ``e0 = LN(MLP(edge_attr))``
* ``graph_network.py:89-90``    edge update   ``e' = LN(MLP(cat[x[src], x[dst], e]))`` with the first Linear split by the
  ``cat`` order into ``Ws x[src] + Wd x[dst] + We e`` (the engine evaluates the first two per NODE: ``Ps``, ``Pd`` tables)
* ``graph_network.py:182``      edge residual ``e += e'``

and WHERE the kernels round (everything else is f32, as in ``cpu_ref``): every MFMA operand -- weights and the activations
that enter a Linear -- is rounded to bf16 (round to nearest even), products are exact, sums are f32 (here: float64, the
midpoint of every summation order); the per-node tables are rounded to their storage type (bf16, or fp16 for the
two-waves-per-SIMD kernel); biases, LayerNorm (eps 1e-5, biased variance) and the residual stream stay f32.  The model
feeds the two-waves-per-SIMD kernel folded LayerNorms (:func:`fold_state_dict`): the emulation of THAT path is this same
arithmetic on the folded parameters (the bf16 roundings of the residual stream then fall on ``e_r - B_r``).

Pinning: :func:`emulate_from_node_latents` runs this arithmetic on the node latents of ``cpu_ref`` itself, and
``tests/test_oracle_bf16_stream.py`` (CPU, every run) holds it within the stated bf16 bound (3e-2 relative L2) of
``cpu_ref.encode_process_decode``'s f32 edge latents on the reference-generated fixture graphs.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn.functional as F

def bf(t):
    return t.bfloat16().float()


def dot_bf16(a, w):
    """bf16 operands, wide accumulation (the MFMA's f32 accumulation order is not reproduced; f64 is the midpoint)."""
    return (bf(a).double() @ bf(w).double().t()).float()



def round_to(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Values as a table of ``dtype`` holds them (bf16 / fp16 storage of the per-node halves)."""
    return t.to(dtype).float()


def s32_position(H: int, dtype: torch.dtype, device=None) -> torch.Tensor:
    """Where feature f = 32t + 8g + 4h + c of a row sits in the engine's gather tables (include/cgnn.h, cgnn_ptable):
    CGNN_P_BF16_S32 (bfloat16): h*(H/2) + u with u = (4t + g)*4 + c; CGNN_P_F16_S32 (float16, H = 128): the two halves
    interleaved in 64-byte segments, (u/32)*64 + h*32 + u%32."""
    f = torch.arange(H, device=device)
    t, g, h, c = f // 32, (f % 32) // 8, (f % 8) // 4, f % 4
    u = (4 * t + g) * 4 + c
    if dtype == torch.float16:
        return (u // 32) * 64 + h * 32 + u % 32
    return h * (H // 2) + u


def s32_table_to_logical(table: torch.Tensor) -> torch.Tensor:
    """A CGNN_P_BF16_S32 (bfloat16) or CGNN_P_F16_S32 (float16) table -> float32 values in feature order (last dimension)."""
    return table[..., s32_position(table.shape[-1], table.dtype, table.device)].float()


def _mlp_tail(h0, lins, ln, centred: bool = False):
    """h0 = pre-activation of layer 0 (bias included); lins = [(w, b)] of layers 1..; ln = (gamma, beta).
    ``centred``: the output Linear was centred (:func:`fold_state_dict`) and the kernel normalises WITHOUT a mean
    (``y * rsqrt(E[y^2] + eps)``: what is left of the mean is the bf16 rounding of the centred weights, ~1e-4 sigma)."""
    h = bf(torch.relu(h0))
    for w, b in lins[:-1]:
        h = bf(torch.relu(dot_bf16(h, w) + b))
    w, b = lins[-1]
    out = dot_bf16(h, w) + b
    if centred:
        return out * torch.rsqrt((out * out).mean(dim=1, keepdim=True) + 1e-5) * ln[0] + ln[1]
    return F.layer_norm(out, (out.shape[1],), ln[0], ln[1], 1e-5)


def fold_state_dict(sd: dict, latent: int, nh: int, rounds: int) -> dict:
    """The reference's parameters with the edge stream's LayerNorms folded (include/cgnn.h, CGNN_STREAM_FOLDED) -- an
    equivalent model: the same final edge latents ``e_L`` in exact arithmetic, the intermediate ``e_r`` shifted by
    ``B_r = beta_0 + ... + beta_{r-1}`` (every later consumer of ``e_r`` is the next round's first Linear, :89, whose bias
    takes ``We_r B_r``).  Output Linears of the edge encoder and of every round's edge model are centred over their output
    features (``LN(y) == LN(y - mean(y))``, :42).  Only edge-model keys change; the node stream never reads ``e``."""
    D = latent
    out = dict(sd)

    def centre(prefix):
        w = sd[f"{prefix}.0.{2 * nh}.weight"].double()
        b = sd[f"{prefix}.0.{2 * nh}.bias"].double()
        out[f"{prefix}.0.{2 * nh}.weight"] = (w - w.mean(dim=0, keepdim=True)).float()
        out[f"{prefix}.0.{2 * nh}.bias"] = (b - b.mean()).float()

    centre("encoder.edge_model")
    B = torch.zeros(D, dtype=torch.float64)
    for r in range(rounds):
        pre = f"processor.{r}.edge_model"
        centre(pre)
        w0 = sd[f"{pre}.0.0.weight"].double()
        out[f"{pre}.0.0.bias"] = (sd[f"{pre}.0.0.bias"].double() + w0[:, 2 * D:3 * D] @ B.to(w0.device)).float()
        B = B + sd[f"{pre}.1.bias"].double().cpu()
        out[f"{pre}.1.bias"] = (B.float() if r + 1 == rounds else torch.zeros(D)).to(sd[f"{pre}.1.bias"].device)
    return out


def emulate_edge_stream_rows(sd: dict, rows: torch.Tensor, stream_inputs: dict, latent: int, nh: int, rounds: int,
                             with_encoder: bool = True, folded: bool = False) -> torch.Tensor:
    """The edge latents after ``rounds`` residual updates for the edge rows ``rows`` (engine numbering), from the
    reference's parameters ``sd`` (state_dict keys of graph_network.py:133-148) and the tables the kernel read.
    ``folded``: ``sd`` is a :func:`fold_state_dict` (the tables then carry its first-Linear biases) and every LayerNorm runs
    without a mean, as the CGNN_STREAM_FOLDED kernel's."""
    dev = rows.device
    D = latent
    W = lambda k: sd[k].to(dev)      # noqa: E731
    src, dst = stream_inputs["src"][rows].long(), stream_inputs["dst"][rows].long()
    if with_encoder:
        pre = "encoder.edge_model"
        lins = [(W(f"{pre}.0.{2 * i}.weight"), W(f"{pre}.0.{2 * i}.bias")) for i in range(nh + 1)]
        attr = stream_inputs["edge_attr"][rows]
        e = _mlp_tail(dot_bf16(attr, lins[0][0]) + lins[0][1], lins[1:], (W(f"{pre}.1.weight"), W(f"{pre}.1.bias")), folded)
    else:
        e = stream_inputs["e_in"][rows].clone()
    for r in range(rounds):
        pre = f"processor.{r}.edge_model"
        w0 = W(f"{pre}.0.0.weight")
        lins = [(W(f"{pre}.0.{2 * i}.weight"), W(f"{pre}.0.{2 * i}.bias")) for i in range(1, nh + 1)]
        ps = s32_table_to_logical(stream_inputs["ps_all"][r][src])
        pd = s32_table_to_logical(stream_inputs["pd_all"][r][dst])       # carries the layer-0 bias
        first = (ps + pd) + dot_bf16(e, w0[:, 2 * D:3 * D])
        e = e + _mlp_tail(first, lins, (W(f"{pre}.1.weight"), W(f"{pre}.1.bias")), folded)
    return e


def emulate_from_node_latents(sd: dict, xs: Sequence[torch.Tensor], edge_index: torch.Tensor, edge_attr: torch.Tensor,
                              latent: int, nh: int, table_dtype: torch.dtype = torch.bfloat16,
                              folded: bool = False) -> torch.Tensor:
    """The edge latents after ``len(xs)`` rounds for ALL edges, from the node latents ``xs[r]`` each round starts with
    (``cpu_ref``'s own: reference-faithful aggregation never feeds the edge stream back): the per-node halves
    ``Ps = Ws x``, ``Pd = Wd x + b`` with bf16 operands, rounded to ``table_dtype`` as the engine stores them.
    ``folded``: as in :func:`emulate_edge_stream_rows`."""
    D = latent
    src, dst = edge_index[0].long(), edge_index[1].long()
    pre = "encoder.edge_model"
    lins = [(sd[f"{pre}.0.{2 * i}.weight"], sd[f"{pre}.0.{2 * i}.bias"]) for i in range(nh + 1)]
    e = _mlp_tail(dot_bf16(edge_attr, lins[0][0]) + lins[0][1], lins[1:], (sd[f"{pre}.1.weight"], sd[f"{pre}.1.bias"]), folded)
    for r, x in enumerate(xs):
        pre = f"processor.{r}.edge_model"
        w0, b0 = sd[f"{pre}.0.0.weight"], sd[f"{pre}.0.0.bias"]
        lins = [(sd[f"{pre}.0.{2 * i}.weight"], sd[f"{pre}.0.{2 * i}.bias"]) for i in range(1, nh + 1)]
        ps = round_to(dot_bf16(x, w0[:, 0:D]), table_dtype)                   # cat order [x[src] | x[dst] | e] (:89)
        pd = round_to(dot_bf16(x, w0[:, D:2 * D]) + b0, table_dtype)          # the layer-0 bias lives in Pd
        first = (ps[src] + pd[dst]) + dot_bf16(e, w0[:, 2 * D:3 * D])
        e = e + _mlp_tail(first, lins, (sd[f"{pre}.1.weight"], sd[f"{pre}.1.bias"]), folded)
    return e

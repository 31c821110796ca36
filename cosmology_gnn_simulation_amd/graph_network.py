"""Drop-in for the reference's ``graph_network`` module, running on gfx950 HIP kernels.

Same public names, constructor signatures, ``state_dict`` keys and return types as
reference graph_network.py (``build_mlp`` :15, ``GraphIndependent`` :39,
``InteractionNetwork`` :67, ``EncodeProcessDecode`` :108), so ``train.py`` /
``one_step_test.py`` style drivers only change their import.  The torch modules
here are parameter containers; every forward runs the fused kernels of
``libcgnn_hip.so`` (``ops.py``).  Tensors must be on a HIP device: there is no CPU
path and no silent fallback.

Engine knobs (attributes, not constructor arguments, so the reference signature
is untouched):

``message_source``  ``"x_j"`` (default) reproduces the reference exactly: it never
    overrides ``MessagePassing.message``, so PyG aggregates the *sender node
    latents* and the updated edge latents never reach the nodes.
    ``"edge"`` aggregates the updated edge latents (the Interaction Network the
    reference's docstrings describe).
``edge_precision``  ``"fp32"`` (exact f32 MFMA, default), ``"bf16"`` (bf16 MFMA operands, f32
    accumulation, f32 LayerNorm / residual: BASELINE cfg3-5), or ``"fp16x2"`` (f32 accuracy from two fp16
    terms per operand on the fp16 matrix cores, latent = hidden = 128: BASELINE cfg2; ``"fp32x3"`` is accepted
    as a synonym -- there is no three-bf16-term edge kernel -- and other shapes take exact f32).
``node_precision``  ``"fp32"`` (default), ``"bf16"``, or one of the two f32 emulations on the matrix cores that
    hold the 1e-5 gate: ``"fp16x2"`` (two fp16 terms, three products; |activation| < 65504, else the row
    comes back NaN) and ``"fp32x3"`` (three bf16 terms, six products, f32 range).  DESIGN.md section 5.
``fuse_rounds`` / ``edge_stream_kernel``  x_j mode: all rounds of the edge stream in one launch
    (``"tile32w"`` two waves per SIMD, ``"tile32"`` one wave per SIMD, ``"tile16"`` first generation).
``train_precision`` / ``train_edge_stream``  arithmetic of the differentiable forward + backward
    (``"fp32"`` / ``"fp32x3"``) and whether a training step also runs the (gradient-free) edge stream.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import CgnnError, require_device
from .graph import Data

__all__ = ["build_mlp", "GraphIndependent", "InteractionNetwork", "EncodeProcessDecode"]


# ----------------------------------------------------------------------------
# module structure (parameter containers)
# ----------------------------------------------------------------------------

def build_mlp(hidden_size: int, num_hidden_layers: int, output_size: int) -> nn.Module:
    """``num_hidden_layers`` x (Linear, ReLU) then Linear(output_size); the first
    Linear is lazy so its fan-in is taken from the first input
    (reference graph_network.py:15-32; Linears sit at even Sequential indices)."""
    mods: List[nn.Module] = [nn.LazyLinear(hidden_size), nn.ReLU(inplace=True)]
    for _ in range(num_hidden_layers - 1):
        mods += [nn.Linear(hidden_size, hidden_size), nn.ReLU(inplace=True)]
    if num_hidden_layers < 1:
        mods = []
    mods.append(nn.Linear(hidden_size, output_size))
    return nn.Sequential(*mods)


def _split_mlp(module: nn.Module) -> Tuple[List[nn.Module], Optional[nn.LayerNorm]]:
    """``Sequential(mlp, LayerNorm)`` or a bare ``build_mlp`` -> (linears, layer_norm)."""
    ln = None
    body = module
    if isinstance(module, nn.Sequential) and len(module) == 2 and isinstance(module[1], nn.LayerNorm):
        body, ln = module[0], module[1]
    if not isinstance(body, nn.Sequential):
        raise TypeError("the fused kernels need models built by build_mlp (optionally followed by LayerNorm); "
                        f"got {type(body).__name__}")
    linears = []
    for i, m in enumerate(body):
        if i % 2 == 0:
            if not isinstance(m, (nn.Linear, nn.LazyLinear)):
                raise TypeError(f"expected Linear at position {i} of the MLP, got {type(m).__name__}")
            linears.append(m)
        elif not isinstance(m, nn.ReLU):
            raise TypeError(f"expected ReLU at position {i} of the MLP, got {type(m).__name__}")
    if len(linears) < 2 or len(body) != 2 * len(linears) - 1:
        raise TypeError("the fused kernels need at least one hidden layer and a Linear output layer")
    if ln is not None and (not ln.elementwise_affine or abs(ln.eps - 1e-5) > 0 or len(ln.normalized_shape) != 1):
        raise TypeError("LayerNorm must be 1-D, affine, eps=1e-5 (torch defaults, as in the reference)")
    return linears, ln


def _materialize(linear: nn.Module, in_features: int) -> None:
    """Give a LazyLinear its shape exactly as its first forward would (same RNG draw)."""
    if isinstance(linear, nn.LazyLinear) and linear.has_uninitialized_params():
        dev = linear.bias.device if linear.bias is not None else None
        with torch.no_grad():
            linear(torch.zeros((0, in_features), device=dev))


def _needs_grad(module: nn.Module) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


_TRAINING_MSG = ("only EncodeProcessDecode (message_source='x_j') has backward kernels; call this module under "
                 "torch.no_grad() / requires_grad_(False), or train through EncodeProcessDecode")


def _params_key(module: nn.Module, *extra) -> tuple:
    return tuple((p.data_ptr(), p._version) for p in module.parameters()) + tuple(extra)


def _wb(lin: nn.Module):
    return lin.weight, lin.bias


def _pack_mlp(module: nn.Module, precision, first_layer_cols=None) -> ops.PackedMLP:
    linears, ln = _split_mlp(module)
    return ops.PackedMLP([_wb(l) for l in linears], None if ln is None else (ln.weight, ln.bias), precision,
                         first_layer_cols)


def _centred_output(lin: nn.Module):
    """(weight, bias) of an MLP's output Linear with the mean over its OUTPUT features removed: the LayerNorm that follows
    (reference graph_network.py:38-42 ``build_mlp`` + ``LayerNorm``) sees ``y - mean(y)``, and ``LN(y) == LN(y - mean(y))``."""
    w = lin.weight.detach().double()
    w = (w - w.mean(dim=0, keepdim=True)).float()
    b = lin.bias
    if b is not None:
        b = b.detach().double()
        b = (b - b.mean()).float()
    return w, b


def _pack_mlp_centred(module: nn.Module, precision) -> ops.PackedMLP:
    """``_pack_mlp`` of an MLP + LayerNorm with the output Linear centred (the same function of its input)."""
    linears, ln = _split_mlp(module)
    wb = [_wb(l) for l in linears[:-1]] + [_centred_output(linears[-1])]
    return ops.PackedMLP(wb, (ln.weight, ln.bias), precision, None)


def fold_edge_stream(edge_models: Sequence[nn.Module], latent: int):
    """The rounds' edge models as ``CGNN_STREAM_FOLDED`` (include/cgnn.h) wants them: per round
    ``(linears [(weight, bias)], (gamma, beta))`` -- the same function ``e_L`` of the inputs as the residual updates of
    reference graph_network.py:89-90, :182, restated so that the one-launch kernel's LayerNorm needs neither a mean nor a shift:

    * every output Linear is centred (``_centred_output``);
    * with ``B_r = beta_0 + ... + beta_{r-1}`` the kernel carries ``e_r - B_r``: round r's first-Linear bias (which lives in
      its Pd table) becomes ``b1_r + We_r B_r`` (``We_r``: the edge-latent block of the first Linear, ``cat`` order :89),
      the LayerNorm shift of every round but the last becomes zero, the last round's becomes ``B_L`` (all of them at once).
    """
    D = latent
    out = []
    B = None
    for r, mod in enumerate(edge_models):
        linears, ln = _split_mlp(mod)
        _materialize(linears[0], 3 * D)
        w1, b1 = _wb(linears[0])
        beta = ln.bias.detach().double()
        if B is None:
            B = torch.zeros_like(beta)
        b1f = (b1.detach().double() if b1 is not None else torch.zeros(w1.shape[0], dtype=torch.float64, device=w1.device)) \
            + w1.detach().double()[:, 2 * D:3 * D] @ B
        last = r + 1 == len(edge_models)
        B = B + beta
        beta_img = B.float() if last else torch.zeros_like(ln.bias)
        wb = [(w1, b1f.float())] + [_wb(l) for l in linears[1:-1]] + [_centred_output(linears[-1])]
        out.append((wb, (ln.weight, beta_img)))
    return out


# ----------------------------------------------------------------------------
# graph arrays the kernels want (int32, fixed in-degree detection)
# ----------------------------------------------------------------------------

def _graph_arrays(data, num_nodes: int):
    """-> (src int32 [E], dst int32 [E], fixed_k).  ``fixed_k`` > 0 when the edge
    list is receiver-sorted with exactly k edges per receiver, which is what
    ``data_utils.preprocess`` emits.  Cached on the data object."""
    ei = data.edge_index
    require_device(ei, "edge_index")
    key = (ei.data_ptr(), ei._version, tuple(ei.shape))
    cached = getattr(data, "_cgnn_graph", None)
    if cached is not None and cached[0] == key:
        return cached[1:]
    src = ei[0].to(torch.int32).contiguous()
    dst = ei[1].to(torch.int32).contiguous()
    ne = ei.shape[1]
    fixed_k = 0
    hint = getattr(data, "_cgnn_fixed_k", None)
    # the hint is only trusted for the edge_index it was computed for (preprocess binds it): a caller that replaces or
    # reorders the edges of a preprocessed graph (same size, other order) gets the one-off layout check instead
    if hint is not None and num_nodes * int(hint) == ne and getattr(data, "_cgnn_fixed_k_for", None) == key:
        fixed_k = int(hint)
    elif num_nodes > 0 and ne > 0 and ne % num_nodes == 0:
        k = ne // num_nodes
        want = torch.arange(num_nodes, device=ei.device, dtype=torch.int32).repeat_interleave(k)
        if bool(torch.equal(dst, want)):
            fixed_k = k
    try:
        data._cgnn_graph = (key, src, dst, fixed_k)
        if fixed_k:
            data._cgnn_fixed_k, data._cgnn_fixed_k_for = fixed_k, key
    except Exception:  # foreign Data types may refuse private attributes
        pass
    return src, dst, fixed_k


def _locality_plan(data, n: int, k: int, src: torch.Tensor):
    """Renumber the particles in the spatial (cell-sorted) order the k-NN build produced, so that a receiver's
    senders sit in nearby rows and the per-edge gathers hit L2 instead of HBM.  Returns ``(order, inverse,
    src_sorted, dst_sorted)`` or ``None`` when the graph carries no order hint.  Index bookkeeping only (torch
    indexing, once per graph, cached on the data object); the per-forward row permutations run in HIP."""
    order = getattr(data, "_cgnn_order", None)
    if order is None or k <= 0 or order.numel() != n:
        return None
    cached = getattr(data, "_cgnn_plan", None)
    if cached is not None and cached[0] is src:
        return cached[1]
    order = order.to(torch.int32)
    ol = order.long()
    inv = torch.empty(n, dtype=torch.int32, device=order.device)
    inv[ol] = torch.arange(n, dtype=torch.int32, device=order.device)
    src_sorted = inv[src.view(n, k)[ol].reshape(-1).long()].contiguous()
    dst_sorted = torch.arange(n, dtype=torch.int32, device=order.device).repeat_interleave(k)
    plan = (order, inv, src_sorted, dst_sorted)
    try:
        data._cgnn_plan = (src, plan)
    except Exception:
        pass
    return plan


# ----------------------------------------------------------------------------
# modules
# ----------------------------------------------------------------------------

class GraphIndependent(nn.Module):
    """Independent node / edge encoders (reference graph_network.py:39-64)."""

    def __init__(self, node_model: nn.Module, edge_model: nn.Module):
        super().__init__()
        self.node_model = node_model
        self.edge_model = edge_model
        self.node_precision = "fp32"
        self.edge_precision = "fp32"
        self._packed = None

    def _pack(self, node_in: int, edge_in: Optional[int]):
        _materialize(_split_mlp(self.node_model)[0][0], node_in)
        if edge_in is not None:
            _materialize(_split_mlp(self.edge_model)[0][0], edge_in)
        key = _params_key(self, self.node_precision, self.edge_precision, edge_in is None)
        if self._packed is None or self._packed[0] != key:
            pn = _pack_mlp(self.node_model, self.node_precision)
            pe = _pack_mlp(self.edge_model, self.edge_precision) if edge_in is not None else None
            self._packed = (key, pn, pe)
        return self._packed[1], self._packed[2]

    def forward(self, data) -> Data:
        if _needs_grad(self):
            raise NotImplementedError(_TRAINING_MSG)
        x = data.x
        require_device(x, "data.x")
        ea = getattr(data, "edge_attr", None)
        with torch.no_grad():
            pn, pe = self._pack(x.shape[1], None if ea is None else ea.shape[1])
            new_x = ops.mlp_rows(pn, x)
            new_e = ops.mlp_rows(pe, ea) if ea is not None else None
        out = Data(x=new_x, edge_index=data.edge_index, edge_attr=new_e)
        if hasattr(data, "globals"):
            out.globals = data.globals
        for hint in ("_cgnn_fixed_k", "_cgnn_fixed_k_for", "_cgnn_graph"):
            if hasattr(data, hint):
                setattr(out, hint, getattr(data, hint))
        return out


class _PackedProcessor:
    """Packed weights of one InteractionNetwork round."""

    def __init__(self, net: "InteractionNetwork", latent: int, edge_precision, node_precision,
                 keep_32_row_edges: bool = False, folded_edge=None):
        """``folded_edge``: this round's entry of ``fold_edge_stream`` -- the edge model (and with it the Pd bias) is then
        packed in that form: for the one-launch edge stream only (``keep_32_row_edges``), a folded round is not the
        round-by-round kernels' round."""
        e_lin, e_ln = _split_mlp(net.edge_model)
        n_lin, n_ln = _split_mlp(net.node_model)
        if e_ln is None or n_ln is None:
            raise TypeError("InteractionNetwork models must end in LayerNorm (build_mlp_with_layer_norm)")
        D = latent
        _materialize(e_lin[0], 3 * D)
        _materialize(n_lin[0], 2 * D)
        w1e, b1e = _wb(e_lin[0])
        w1n, b1n = _wb(n_lin[0])
        if folded_edge is not None:
            if not keep_32_row_edges:
                raise CgnnError("a folded edge model is the one-launch edge stream's (keep_32_row_edges)")
            b1e = folded_edge[0][0][1]
        if w1e.shape[1] != 3 * D or w1n.shape[1] != 2 * D:
            raise CgnnError(f"first-layer fan-in {w1e.shape[1]}/{w1n.shape[1]} does not match latent size {D}")
        # cat([x[src], x[dest], edge_attr]) -> [Ws | Wd | We]      (reference graph_network.py:89)
        if str(edge_precision).lower() in ("fp16x2", "f16x2", "fp32x3", "f32x3"):
            # f32 accuracy on the matrix cores: the two-fp16-term edge kernel (latent = hidden = 128, <= 3 hidden layers,
            # f32 Ps / Pd tables); other shapes take the exact-f32 kernels.  "fp32x3" has no three-bf16-term EDGE kernel: it
            # keeps f32's exponent range only through the exact-f32 kernels, so it is routed there (never to fp16 terms,
            # whose range ends at 65504)
            e_lins = _split_mlp(net.edge_model)[0]
            two_terms = str(edge_precision).lower() in ("fp16x2", "f16x2")
            fits = two_terms and D == 128 and w1e.shape[0] == 128 and len(e_lins) - 1 <= 3 and \
                all(l.bias is not None for l in e_lins)
            proj_precision, edge_precision = ("fp16x2" if two_terms else "fp32"), ("fp16x2_n16" if fits else "fp32")
        else:
            proj_precision = edge_precision
        self.ws = ops.PackedLinear(w1e, None, proj_precision, 0, D)
        self.wd = ops.PackedLinear(w1e, b1e, proj_precision, D, D)
        if folded_edge is not None:
            self.edge = ops.PackedMLP(folded_edge[0], folded_edge[1], edge_precision, (2 * D, D))
        else:
            self.edge = _pack_mlp(net.edge_model, edge_precision, first_layer_cols=(2 * D, D))
        # keep_32_row_edges: the model runs its edge stream through cgnn_edge_stream_run, which takes the 32-row packing
        wide = D == 256 and self.edge.hidden == 256 and self.edge.num_hidden_layers <= 3 and \
            all(l.bias is not None for l in _split_mlp(net.edge_model)[0])
        if not keep_32_row_edges and self.edge.precision == _lib.BF16 and (wide or (
                D <= 128 and self.edge.hidden <= 128 and self.edge.lds_bytes() <= _lib.LDS_WEIGHT_BUDGET)):
            # the 16-edge-per-wave kernels (own packing + P-table format): weights resident in LDS up to 128, streamed
            # through an LDS ring at latent = hidden = 256
            self.edge = _pack_mlp(net.edge_model, "bf16_n16", first_layer_cols=(2 * D, D))
        self.p_format = ops.p_table_format(self.edge.precision)
        self.p_dtype = ops.p_table_dtype(self.edge.precision)
        # cat([x, aggregated]) -> [Wx | Wa]                         (reference graph_network.py:94)
        node_fmt = node_precision
        if ops._prec(node_precision) in _lib.F32_EMULATED and D <= 128 and w1n.shape[0] == D:
            # square layers <= 128: the 16-row, two-waves-per-SIMD node kernel, on two fp16 or three bf16 terms
            node_fmt = "fp16x2_n16" if ops._prec(node_precision) == _lib.F16X2 else "fp32x3_n16"
        self.wx = ops.PackedLinear(w1n, b1n, node_fmt, 0, D)
        self.wa = ops.PackedLinear(w1n, None, node_fmt, D, D)
        self.node = _pack_mlp(net.node_model, node_fmt, first_layer_cols=(0, D))
        # projection weights in the packing the PREVIOUS round's node kernel can fuse (it produces this round's
        # Ps / Pd): the N16 node kernel takes CGNN_BF16_N16, everything else the 32-row packing above
        self.ws_fused, self.wd_fused = self.ws, self.wd
        if self.node.precision in _lib.N16_NODE and self.ws.precision == _lib.BF16 and w1e.shape[0] == D:
            self.ws_fused = ops.PackedLinear(w1e, None, "bf16_n16", 0, D)
            self.wd_fused = ops.PackedLinear(w1e, b1e, "bf16_n16", D, D)


def _run_round(p: _PackedProcessor, x: torch.Tensor, e: torch.Tensor, src, dst, fixed_k: int, message_source: str,
               residual: bool, x_out=None, e_out=None, scratch=None, projected: bool = False,
               next_round: Optional[_PackedProcessor] = None):
    """One message-passing round.  Returns (x_new, e_new).  ``projected``: the Ps/Pd tables in ``scratch`` were
    already filled for this round (by the previous round's node kernel); ``next_round``: fill them for the next."""
    n = x.shape[0]
    ps = pd = agg = e_upd = None
    if scratch is not None:
        ps, pd, agg, e_upd = scratch
    if not projected:
        ps, pd = ops.project_nodes(p.ws, p.wd, x, ps, pd, p.p_format)
    if message_source not in ("x_j", "edge"):
        raise ValueError(f"message_source must be 'x_j' or 'edge', got {message_source!r}")
    if message_source == "edge" and p.edge.precision == _lib.BF16_N16 and fixed_k in (8, 16) and x.shape[1] <= 128:
        # the 16-edge kernel folds the aggregation of the edge updates in: one wave tile is exactly one (k=16) or
        # two (k=8) receivers, so the sum is a cross-lane reduction and the e_upd round trip (2 E D 4 bytes)
        # disappears.  (Folding the x_j gather in as well measured slower than the stand-alone kernel: +2.0 ms
        # against 0.7 ms at cfg3 -- its row gathers are better coalesced there.)
        if agg is None:
            agg = torch.empty((n, x.shape[1]), dtype=torch.float32, device=x.device)
        e_new = ops.edge_block(p.edge, ps, pd, src, dst, e, e_out, None, residual, agg_out=agg, x_gather=None,
                               seg_k=fixed_k)
    else:
        if message_source == "edge" and e_upd is None:
            e_upd = e.empty_like()
        e_new = ops.edge_block(p.edge, ps, pd, src, dst, e, e_out, e_upd if message_source == "edge" else None,
                               residual)
        if message_source == "x_j":
            agg = ops.aggregate(x, src, dst, n, fixed_k, src.numel(), agg,
                                plan=ops.AggregatePlan.of(src, n, fixed_k, x.shape[1]))
        else:
            agg = ops.aggregate(e_upd, None, dst, n, fixed_k, src.numel(), agg)
    nxt = None
    if next_round is not None and next_round.p_format == p.p_format and next_round.p_dtype == ps.dtype:
        fused_ok = p.node.precision in _lib.N16_NODE and next_round.ws_fused.precision == _lib.BF16_N16
        nxt = (next_round.ws_fused if fused_ok else next_round.ws, next_round.wd_fused if fused_ok else next_round.wd,
               ps, pd, next_round.p_format)
    x_new = ops.node_block(p.node, p.wx, p.wa, x, agg, x_out, residual, nxt)
    return x_new, e_new, nxt is not None


def stream_table_format(rounds, stream_kernel: Optional[str], stream_lag: int = 0) -> int:
    """``cgnn_ptable`` format of the Ps / Pd tables the node stream leaves for the one-launch edge stream.
    ``cgnn_edge_stream_run_w8`` (``"tile32w"``, lag 0) adds ``Ps[src] + Pd[dst]`` on the vector pipe when the tables are fp16
    (``CGNN_P_F16_S32``: the row's halves interleaved in 64-byte segments, the same projection arithmetic, the f32 sums rounded to 11 significand bits instead of
    bf16's 8): 16 fewer MFMAs per tile and round than the selector MFMAs that bf16 rows need."""
    fmt = rounds[0].p_format
    if stream_kernel == "tile32w" and stream_lag == 0 and fmt == _lib.P_BF16_S32:
        fmt = _lib.P_F16_S32
    return fmt


def _run_rounds_fused(rounds, x: torch.Tensor, e, src, dst, fixed_k: int, agg: Optional[torch.Tensor],
                      encoder=None, edge_attr: Optional[torch.Tensor] = None, image=None, keep: Optional[dict] = None,
                      stream_kernel: str = "tile32", stream_lag: int = 0):
    """All residual rounds under the reference's data flow (aggregation of sender NODE latents): the node
    stream does not read the edge stream, so it runs first and leaves every round's Ps / Pd tables behind (the node
    kernel's epilogue writes round i+1's); then one launch applies all edge updates while each edge tile stays in
    registers.  Same kernels' arithmetic as the round-by-round path: results are bit-identical."""
    n, L = x.shape[0], len(rounds)
    H = rounds[0].ws.out_dim
    fmt = stream_table_format(rounds, stream_kernel if image is not None else None, stream_lag)
    pdt = ops.p_format_dtype(fmt)
    ps_all = torch.empty((L, n, H), dtype=pdt, device=x.device)
    pd_all = torch.empty((L, n, H), dtype=pdt, device=x.device)
    ops.project_nodes(rounds[0].ws, rounds[0].wd, x, ps_all[0], pd_all[0], fmt)
    plan = ops.AggregatePlan.of(src, n, fixed_k, x.shape[1])       # built once per graph, cached on its sender list
    for i, p in enumerate(rounds):
        agg = ops.aggregate(x, src, dst, n, fixed_k, src.numel(), agg, plan=plan)
        nxt = None
        if i + 1 < L:
            q = rounds[i + 1]
            fused_ok = p.node.precision in _lib.N16_NODE and q.ws_fused.precision == _lib.BF16_N16
            nxt = (q.ws_fused if fused_ok else q.ws, q.wd_fused if fused_ok else q.wd, ps_all[i + 1], pd_all[i + 1], fmt)
        x = ops.node_block(p.node, p.wx, p.wa, x, agg, x, True, nxt)
    if keep is not None:      # tests: what the one-launch edge stream is about to consume (EncodeProcessDecode.keep_stream_inputs)
        keep.update(ps_all=ps_all, pd_all=pd_all, src=src, dst=dst, edge_attr=edge_attr, p_format=fmt,
                    e_in=None if e is None else e.to_rows(), folded=bool(image is not None and image.folded))
    # `encoder` (the packed edge encoder) given: the initial edge latents are computed inside the same launch and
    # never written to memory (e is None then)
    if image is not None:      # cgnn_edge_stream_run: the rounds (and the encoder, if it is part of the image) as one image
        e = ops.edge_stream_run(image, ps_all, pd_all, src, dst, e, e, edge_attr if image.enc_in else None,
                                kernel=stream_kernel, lag=stream_lag, fixed_k=fixed_k)
    else:
        e = ops.edge_stream([p.edge for p in rounds], ps_all, pd_all, src, dst, e, e, encoder, edge_attr)
    return x, e


class InteractionNetwork(nn.Module):
    """One round of message passing without residuals (reference
    graph_network.py:67-101): edge update from (sender, receiver, edge), sum
    aggregation at the receivers, node update from (node, aggregate)."""

    def __init__(self, node_model: nn.Module, edge_model: nn.Module, aggr: str = "add"):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError("only aggr='add' is built (the reference never uses another)")
        self.aggr = aggr
        self.node_model = node_model
        self.edge_model = edge_model
        self.message_source = "x_j"
        self.node_precision = "fp32"
        self.edge_precision = "fp32"
        self._packed = None

    def _pack(self, latent: int) -> _PackedProcessor:
        key = None
        lazy = any(isinstance(m, nn.LazyLinear) and m.has_uninitialized_params() for m in self.modules())
        if not lazy:
            key = _params_key(self, self.node_precision, self.edge_precision, latent)
            if self._packed is not None and self._packed[0] == key:
                return self._packed[1]
        packed = _PackedProcessor(self, latent, self.edge_precision, self.node_precision)
        self._packed = (_params_key(self, self.node_precision, self.edge_precision, latent), packed)
        return packed

    def forward(self, data) -> Data:
        x, edge_attr = data.x, getattr(data, "edge_attr", None)
        if edge_attr is None:
            raise ValueError("edge_attr must not be None in InteractionNetwork")
        if _needs_grad(self):
            raise NotImplementedError(_TRAINING_MSG)
        require_device(x, "data.x")
        with torch.no_grad():
            src, dst, fixed_k = _graph_arrays(data, x.shape[0])
            p = self._pack(x.shape[1])
            xf = x.float().contiguous()
            ef = ops.TiledRows.from_rows(edge_attr.float().contiguous())     # engine-internal edge layout
            new_x, new_e, _ = _run_round(p, xf, ef, src, dst, fixed_k, self.message_source, residual=False)
        out = Data(x=new_x, edge_index=data.edge_index, edge_attr=new_e.to_rows())
        if hasattr(data, "globals"):
            out.globals = data.globals
        for hint in ("_cgnn_fixed_k", "_cgnn_fixed_k_for", "_cgnn_graph"):
            if hasattr(data, hint):
                setattr(out, hint, getattr(data, hint))
        return out


class EncodeProcessDecode(nn.Module):
    """Encoder, ``num_message_passing_steps`` residual InteractionNetwork rounds, and
    the acceleration / temperature-rate decoders (reference graph_network.py:108-183).
    ``forward`` returns ``{'acceleration': [N, output_size], 'temp_rate': [N, 1]}``."""

    def __init__(self, latent_size: int, mlp_hidden_size: int, mlp_num_hidden_layers: int,
                 num_message_passing_steps: int, output_size: int):
        super().__init__()
        self._latent_size = latent_size
        self._mlp_hidden_size = mlp_hidden_size
        self._mlp_num_hidden_layers = mlp_num_hidden_layers
        self._num_message_passing_steps = num_message_passing_steps
        self._output_size = output_size

        def mlp_ln() -> nn.Module:
            return nn.Sequential(build_mlp(mlp_hidden_size, mlp_num_hidden_layers, latent_size),
                                 nn.LayerNorm(latent_size))

        self.encoder = GraphIndependent(node_model=mlp_ln(), edge_model=mlp_ln())
        self.processor = nn.ModuleList(
            [InteractionNetwork(edge_model=mlp_ln(), node_model=mlp_ln()) for _ in range(num_message_passing_steps)])
        self.decoder_acc = build_mlp(mlp_hidden_size, mlp_num_hidden_layers, output_size)
        self.decoder_temp_rate = build_mlp(mlp_hidden_size, mlp_num_hidden_layers, 1)

        self.message_source = "x_j"
        self.node_precision = "fp32"
        self.edge_precision = "fp32"
        self.train_precision = "fp32"  # arithmetic of the differentiable forward + backward: "fp32" or "fp32x3"
        self.locality_sort = True     # run in the k-NN build's spatial order when the graph carries it
        self.fuse_rounds = True       # x_j mode: all rounds of the edge stream in one launch
        # which one-launch kernel: "tile32" = cgnn_edge_stream_run (32-edge MFMA tiles, one wave per SIMD, two tiles per
        # wave), "tile32w" = cgnn_edge_stream_run_w8 (32-edge tiles, two waves per SIMD, one tile each; latent 128 only,
        # other shapes take "tile32"), "tile16" = cgnn_edge_stream (16-edge tiles; the first generation, kept for comparison)
        self.edge_stream_kernel = "tile32w"
        self.edge_stream_lag = 0      # "tile32w": second wave of a SIMD one layer behind the first (1) or in step (0)
        # tests only: forward_with_latents() also returns the one-launch edge stream's inputs (every round's Ps / Pd table,
        # the renumbered edge list and edge features) under "stream_inputs", so that sampled edge rows can be recomputed
        self.keep_stream_inputs = False
        self._packed = None
        self._train_packed = None

    # -- packing ---------------------------------------------------------------
    def _materialize_all(self, node_in: int, edge_in: int) -> None:
        """Shape every LazyLinear in the order the reference's first forward would
        (encoder node, encoder edge, per round edge then node, decoders), so a
        seeded random initialisation matches the reference's."""
        D = self._latent_size
        _materialize(_split_mlp(self.encoder.node_model)[0][0], node_in)
        _materialize(_split_mlp(self.encoder.edge_model)[0][0], edge_in)
        for net in self.processor:
            _materialize(_split_mlp(net.edge_model)[0][0], 3 * D)
            _materialize(_split_mlp(net.node_model)[0][0], 2 * D)
        _materialize(_split_mlp(self.decoder_acc)[0][0], D)
        _materialize(_split_mlp(self.decoder_temp_rate)[0][0], D)

    def _pack(self, node_in: int, edge_in: int):
        self._materialize_all(node_in, edge_in)
        key = _params_key(self, self.node_precision, self.edge_precision, self.fuse_rounds, self.message_source,
                          self.edge_stream_kernel)
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        D, H, nh, L = self._latent_size, self._mlp_hidden_size, self._mlp_num_hidden_layers, len(self.processor)
        # cgnn_edge_stream_run (all rounds of the edge stream in one launch, 32-edge tiles) under the reference's data flow
        tile32 = (self.fuse_rounds and self.message_source == "x_j" and self.edge_stream_kernel in ("tile32", "tile32w") and L > 0 and
                  ops._prec(self.edge_precision) == _lib.BF16 and ops.StreamImage.supported(D, H, nh, L))
        enc_edge = _pack_mlp(self.encoder.edge_model, self.edge_precision)
        enc_in_image = tile32 and ops.StreamImage.supported(D, H, nh, L, enc_edge.in_dim)
        enc_lins = _split_mlp(self.encoder.edge_model)[0]
        # latent = hidden = 256: the encoder's weights stream through the LDS ring of the 256-wide edge kernel
        # (edge_block_ring256.hip), which reads an edge's features with one aligned 16-byte load: exactly 4 of them
        wide_enc = D == 256 and enc_edge.hidden == 256 and enc_edge.in_dim == 4 and enc_edge.num_hidden_layers <= 3 and \
            all(l.bias is not None for l in enc_lins)
        if not tile32 and enc_edge.precision == _lib.BF16 and enc_edge.in_dim <= 32 and (wide_enc or (
                D <= 128 and enc_edge.hidden <= 128 and enc_edge.lds_bytes() <= _lib.LDS_WEIGHT_BUDGET)):
            enc_edge = _pack_mlp(self.encoder.edge_model, "bf16_n16")    # 16-edge-per-wave encoder, TILED32 output
        # the one-launch edge stream runs the rounds' edge models with their LayerNorms folded (CGNN_STREAM_FOLDED): the same
        # e_L, 128 fewer vector instructions per 32-edge tile and round
        folded = fold_edge_stream([net.edge_model for net in self.processor], D) if tile32 else None
        rounds = [_PackedProcessor(net, D, self.edge_precision, self.node_precision, keep_32_row_edges=tile32,
                                   folded_edge=folded[i] if folded is not None else None)
                  for i, net in enumerate(self.processor)]
        enc_image = _pack_mlp_centred(self.encoder.edge_model, self.edge_precision) if enc_in_image else None
        packed = dict(
            enc_node=_pack_mlp(self.encoder.node_model, self.node_precision),
            enc_edge=enc_edge,
            rounds=rounds,
            dec_acc=_pack_mlp(self.decoder_acc, self.node_precision),
            dec_tr=_pack_mlp(self.decoder_temp_rate, self.node_precision),
            image=ops.StreamImage([p.edge for p in rounds], enc_image, folded=True) if tile32 else None,
            image_w8=None,      # the same for cgnn_edge_stream_run_w8 (every bias one chunk early), built on first use
            image_parts=([p.edge for p in rounds], enc_image) if tile32 else None,
        )
        self._packed = (key, packed)
        return packed

    def _edge_stream_plan(self, P, fixed_k: int, num_edges: int, edge_attr: Optional[torch.Tensor]):
        """-> (image, kernel) for ``ops.edge_stream_run``: the two-waves-per-SIMD kernel (``cgnn_edge_stream_run_w8``, its own
        image, built on first use) where ``edge_stream_kernel == "tile32w"`` and the shape, the graph layout (receiver-sorted,
        fixed in-degree 8, 16, 32, ...) and the edge-feature layout qualify; otherwise ``cgnn_edge_stream_run``."""
        image = P["image"]
        if image is None or self.edge_stream_kernel != "tile32w":
            return image, "tile32"
        ok = ops.stream_w8_supported(image.latent, image.nh, fixed_k) and num_edges % max(fixed_k, 1) == 0
        if ok and image.enc_in:
            ok = image.enc_in <= 4 and edge_attr is not None and edge_attr.stride(0) % 4 == 0 and edge_attr.data_ptr() % 16 == 0
        if not ok:
            return image, "tile32"
        if P["image_w8"] is None:
            P["image_w8"] = ops.StreamImage(*P["image_parts"], kernel="tile32w", folded=True)
        return P["image_w8"], "tile32w"

    def _can_fuse_rounds(self, rounds, latent: int) -> bool:
        """One launch for the whole edge stream (``cgnn_edge_stream``): only under the reference's own data flow
        (``message_source="x_j"``: the node stream never reads the edge stream), with the 16-edge bf16 kernels."""
        if not (self.fuse_rounds and self.message_source == "x_j" and rounds):
            return False
        nh = rounds[0].edge.num_hidden_layers
        if len(rounds) > 32 or len(rounds) * (nh + 1) > 64 or latent not in (32, 64, 128):
            return False
        lds = 3 * 2 * latent * latent + len(rounds) * (nh + 2) * latent * 4 + 12 * 64     # ring + vectors + layer table
        return lds <= 160 * 1024 and all(
            p.edge.precision == _lib.BF16_N16 and p.p_format == _lib.P_BF16_S16 and p.edge.hidden == latent and
            p.edge.num_hidden_layers == nh for p in rounds)

    def _encoder_fits_stream(self, P) -> bool:
        """The edge encoder can run inside ``cgnn_edge_stream`` (16-edge bf16 packing, same depth and widths as the
        rounds' edge models, LDS room for its vectors)."""
        rounds, enc = P["rounds"], P["enc_edge"]
        if not rounds or enc.precision != _lib.BF16_N16:
            return False
        e0 = rounds[0].edge
        latent = e0.out_dim
        lds = 3 * 2 * latent * latent + (len(rounds) * (e0.num_hidden_layers + 2) + e0.num_hidden_layers + 3) * latent * 4 \
            + 12 * 64
        return (enc.num_hidden_layers == e0.num_hidden_layers and enc.hidden == e0.hidden and enc.out_dim == latent and
                enc.in_dim <= 32 and (len(rounds) + 1) * (e0.num_hidden_layers + 1) <= 64 and lds <= 160 * 1024)

    def invalidate_packed(self) -> None:
        """Forget the MFMA-packed copies of the weights.  They are rebuilt when a parameter tensor is replaced or
        modified in place THROUGH autograd-visible operations (``optimizer.step()``, ``load_state_dict``, ``p.copy_``):
        the cache is keyed on ``(data_ptr, _version)``.  Updates made through ``param.data`` (``p.data.mul_()``, EMA /
        SWA averaging, some initialisers) do not bump that version: call this afterwards."""
        self._packed = None
        self._train_packed = None
        self.encoder._packed = None
        for net in self.processor:
            net._packed = None

    # -- forward ---------------------------------------------------------------
    def forward(self, input_graph) -> dict:
        out = self._forward(input_graph, want_latents=False)
        return {"acceleration": out["acceleration"], "temp_rate": out["temp_rate"]}

    def forward_with_latents(self, input_graph) -> dict:
        """Same as :meth:`forward` plus ``x_latent`` / ``edge_latent`` after the last round."""
        return self._forward(input_graph, want_latents=True)

    def _train_packs(self):
        from .training import TrainPacks
        key = _params_key(self, "train", getattr(self, "train_precision", "fp32"))
        if self._train_packed is None or self._train_packed[0] != key:
            self._train_packed = (key, TrainPacks(self))
        return self._train_packed[1]

    def _forward_train(self, g) -> dict:
        """Differentiable forward (exact f32) for ``train.py``: see :mod:`.training`.  Edge-model parameters get no
        gradient, exactly as under the reference."""
        from . import training
        if self.message_source != "x_j":
            raise NotImplementedError("training is built for the reference-faithful message_source='x_j'; the "
                                      "'edge' extension has no backward kernels")
        x = g.x
        require_device(x, "input_graph.x")
        edge_attr = getattr(g, "edge_attr", None)
        if edge_attr is None:
            raise ValueError("edge_attr must not be None in InteractionNetwork")
        glob = getattr(g, "globals", None)
        if glob is not None:
            x = torch.cat([x, glob.unsqueeze(0).expand(x.shape[0], -1)], dim=-1)
        x = x.float().contiguous()
        n = x.shape[0]
        with torch.no_grad():
            src, dst, fixed_k = _graph_arrays(g, n)
            plan = _locality_plan(g, n, fixed_k, src) if (self.locality_sort and fixed_k > 0) else None
            self._materialize_all(x.shape[1], edge_attr.shape[1])
            packs = self._train_packs()
        if plan is not None:
            order, inv, src, dst = plan
            x = training.permute_rows(x, order, inv)
        cached = getattr(g, "_cgnn_by_sender", None)      # transposed adjacency, once per graph
        if cached is None or cached[0] is not src:
            cached = (src, ops.SenderCsr(src, dst, n))
            try:
                g._cgnn_by_sender = cached
            except Exception:
                pass
        packs.edge_stream_fn = None
        if getattr(self, "train_edge_stream", False):
            # like-for-like with the reference's step, which computes the edge stream although nothing reads it
            ea = edge_attr.detach().float().contiguous()
            if plan is not None:
                ea = ops.gather_rows(ea.view(n, -1), order).view(n * fixed_k, -1)
            node_in = x.shape[1]
            packs.edge_stream_fn = lambda xs: training.edge_stream_of(self, xs, src, dst, fixed_k, ea, node_in)
        acc, tr = training.forward_train(self, x, src, dst, fixed_k, packs, cached[1])
        packs.edge_stream_fn = None
        if plan is not None:
            acc = training.permute_rows(acc, inv, order)
            tr = training.permute_rows(tr, inv, order)
        return {"acceleration": acc, "temp_rate": tr}

    def _forward(self, g, want_latents: bool) -> dict:
        if torch.is_grad_enabled() and (_needs_grad(self) or g.x.requires_grad):
            if want_latents:
                raise NotImplementedError("forward_with_latents is inference-only: call it under torch.no_grad()")
            return self._forward_train(g)
        x = g.x
        require_device(x, "input_graph.x")
        edge_attr = getattr(g, "edge_attr", None)
        if edge_attr is None:
            raise ValueError("edge_attr must not be None in InteractionNetwork")
        with torch.no_grad():
            glob = getattr(g, "globals", None)
            if glob is not None:  # reference graph_network.py:168-173 (never set by its own drivers)
                x = torch.cat([x, glob.unsqueeze(0).expand(x.shape[0], -1)], dim=-1)
            x = x.float().contiguous()
            edge_attr = edge_attr.float().contiguous()
            n = x.shape[0]
            src, dst, fixed_k = _graph_arrays(g, n)
            plan = _locality_plan(g, n, fixed_k, src) if (self.locality_sort and fixed_k > 0) else None
            if plan is not None:
                order, inv, src, dst = plan
                x = ops.gather_rows(x, order)
                edge_attr = ops.gather_rows(edge_attr.view(n, -1), order).view(n * fixed_k, -1)
            P = self._pack(x.shape[1], edge_attr.shape[1])
            xl = ops.mlp_rows(P["enc_node"], x)
            image = P["image"]
            fuse = image is not None or self._can_fuse_rounds(P["rounds"], xl.shape[1])
            enc_in_stream = bool(image.enc_in) if image is not None else (fuse and self._encoder_fits_stream(P))
            # edge latents live in TILED32 layout; in the fused path the encoder runs inside cgnn_edge_stream
            el = None if enc_in_stream else ops.mlp_rows(P["enc_edge"], edge_attr, tiled=True)
            H = P["rounds"][0].ws.out_dim if P["rounds"] else 0
            scratch = None
            if P["rounds"]:
                dev = x.device
                pdt = P["rounds"][0].p_dtype
                ps = torch.empty((n, H), dtype=pdt, device=dev)
                pd = torch.empty((n, H), dtype=pdt, device=dev)
                agg = torch.empty((n, xl.shape[1]), dtype=torch.float32, device=dev)
                e_upd = el.empty_like() if (self.message_source == "edge" and el is not None) else None
                scratch = (ps, pd, agg, e_upd)
            projected = False
            rounds = P["rounds"]
            keep = {} if (want_latents and getattr(self, "keep_stream_inputs", False)) else None
            if fuse:
                image, stream_kernel = self._edge_stream_plan(P, fixed_k, src.numel(), edge_attr)
                xl, el = _run_rounds_fused(rounds, xl, el, src, dst, fixed_k, agg,
                                           P["enc_edge"] if enc_in_stream else None, edge_attr, image, keep,
                                           stream_kernel, int(getattr(self, "edge_stream_lag", 0)))
                rounds = []
            for i, p in enumerate(rounds):
                # residual streams updated in place (reference graph_network.py:181-182); the node kernel also
                # emits the next round's sender / receiver projections
                nxt = rounds[i + 1] if i + 1 < len(rounds) else None
                xl, el, projected = _run_round(p, xl, el, src, dst, fixed_k, self.message_source, residual=True,
                                               x_out=xl, e_out=el, scratch=scratch, projected=projected,
                                               next_round=nxt)
            out = {"acceleration": ops.mlp_rows(P["dec_acc"], xl), "temp_rate": ops.mlp_rows(P["dec_tr"], xl)}
            if want_latents:
                out["x_latent"], out["edge_latent"] = xl, el.to_rows()
                if keep:       # engine numbering (the locality order), next to the edge latents in that same numbering
                    keep["edge_latent_sorted"] = out["edge_latent"]
                    out["stream_inputs"] = keep
            if plan is not None:   # back to the caller's particle numbering
                out["acceleration"] = ops.gather_rows(out["acceleration"], inv)
                out["temp_rate"] = ops.gather_rows(out["temp_rate"], inv)
                if want_latents:
                    out["x_latent"] = ops.gather_rows(xl, inv)
                    out["edge_latent"] = ops.gather_rows(out["edge_latent"].view(n, -1), inv).view(n * fixed_k, -1)
        return out

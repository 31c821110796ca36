"""Training through the fused kernels: forward + backward of the node stream.

What ``combined_loss.backward()`` of reference train.py:263 actually differentiates: the reference never overrides
``MessagePassing.message`` (graph_network.py:92-101), so the aggregation sums the *sender node latents* and the edge
stream never reaches the outputs.  The autograd graph from the loss therefore contains only

    x0 --node encoder--> x_0;   agg_i = sum_{senders} x_i;   x_{i+1} = x_i + LN(MLP_i([x_i, agg_i]));
    acceleration = dec_acc(x_L),  temp_rate = dec_tr(x_L)

and every edge-model parameter keeps ``grad = None`` under the reference as well.  This module runs exactly that
graph in exact f32: the training forward SKIPS the (dead) edge stream -- every step time quoted for it carries that label;
``model.train_edge_stream = True`` runs the edge stream's forward as well (the reference computes it although nothing
reads it), for a like-for-like step time -- keeps ``x_i`` and ``agg_i`` per round, and the backward
recomputes the activations tile by tile (``cgnn_mlp_backward``), transposes the aggregation by gathering through
the sender-major adjacency (``cgnn_csr_build`` once per graph, ``cgnn_aggregate_csr``) and reduces the parameter
gradients with ``cgnn_weight_grad`` / ``cgnn_col_dot``.

``message_source="edge"`` (the engine's extension, not the reference's behaviour) has no backward yet.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import CgnnError


class _TrainMLP:
    """Forward and transposed packings (exact f32, or f32 emulated by three bf16 terms) of one ``build_mlp`` (+LayerNorm), plus its parameter list in
    ``module.parameters()`` order: (w, b) per Linear, then LayerNorm (weight, bias)."""

    def __init__(self, linears: Sequence[nn.Module], ln: Optional[nn.LayerNorm], split_at: Optional[int] = None,
                 precision: str = "fp32", latent_input: bool = False):
        if ops._prec(precision) not in (_lib.F32, _lib.F32X3):
            raise CgnnError(f"training arithmetic must be 'fp32' (exact) or 'fp32x3' (three bf16 terms), got {precision!r}")
        self.linears, self.ln = list(linears), ln
        self.split_at = split_at
        self.precision = precision
        w0 = linears[0].weight
        wb = [(l.weight, l.bias) for l in linears]
        lnp = None if ln is None else (ln.weight, ln.bias)
        self.in1 = int(split_at if split_at is not None else w0.shape[1])
        self.in2 = int(w0.shape[1] - self.in1)
        cols = (0, self.in1) if split_at is not None else None
        two_terms = ops._prec(precision) == _lib.F32X3 and latent_input       # see below
        fwd_prec = "fp16x2" if two_terms else precision
        self.fwd = ops.PackedMLP(wb, lnp, fwd_prec, first_layer_cols=cols)
        self.fwd2 = ops.PackedLinear(w0, None, fwd_prec, self.in1, self.in2) if self.in2 else None
        # Under the emulated arithmetic, MLPs fed by latents (processor rounds, decoders: values of O(1) behind a
        # LayerNorm) take the two-fp16-term kernels wherever no gradient passes through the operands:
        #   run / run2  the differentiable forward (it only hands f32-accurate latents to the backward): the 16-row
        #               ring kernel for square layers up to 128, else the 32-row packing;
        #   rec / rec2  the forward that cgnn_mlp_backward recomputes (it takes (fp16x2, fp32x3) pairs).
        # The encoder sees raw features and every gradient can be 1e-8: those stay on three bf16 terms (f32 range).
        self.run, self.run2 = self.fwd, self.fwd2
        self.rec, self.rec2 = self.fwd, self.fwd2
        if two_terms:
            square = self.in2 == self.in1 == int(w0.shape[0]) and self.in1 in (32, 64, 128) and ln is not None and \
                all(int(l.weight.shape[0]) == self.in1 == int(l.weight.shape[1]) for l in linears[1:])
            if square:
                self.run = ops.PackedMLP(wb, lnp, "fp16x2_n16", first_layer_cols=cols)
                self.run2 = ops.PackedLinear(w0, None, "fp16x2_n16", self.in1, self.in2)
        t = lambda w: w.detach().t().contiguous()  # noqa: E731
        tw = [(t(w0[:, :self.in1]), None)] + [(t(l.weight), None) for l in linears[1:]]
        self.bwd = ops.PackedMLP(tw, None, precision)
        self.bwd2 = ops.PackedLinear(t(w0[:, self.in1:]), None, precision) if self.in2 else None
        self.hidden = self.fwd.hidden
        self.out_dim = self.fwd.out_dim
        self.out_padded = (self.out_dim + 31) // 32 * 32
        self.nh = self.fwd.num_hidden_layers

    def params(self) -> List[torch.Tensor]:
        out: List[torch.Tensor] = []
        for l in self.linears:
            out += [l.weight, l.bias]
        if self.ln is not None:
            out += [self.ln.weight, self.ln.bias]
        return out

    def backward(self, u1: torch.Tensor, u2: Optional[torch.Tensor], dy: torch.Tensor, scratch: ops.BackwardScratch,
                 want_du1: bool, want_du2: bool = True):
        """-> (du1, du2, [parameter gradients in ``params()`` order])."""
        n = u1.shape[0]
        dy = dy.contiguous()
        du1, du2 = ops.mlp_backward(self.rec, self.rec2, self.bwd, self.bwd2, u1, u2, dy, scratch, want_du1, want_du2)
        grads: List[torch.Tensor] = []
        H = self.hidden
        for l, lin in enumerate(self.linears):
            last = l == self.nh
            g = scratch.g_o if last else scratch.g_a[l]
            ld_g = self.out_padded if last else H
            out_dim = lin.weight.shape[0]
            dw = torch.zeros_like(lin.weight, memory_format=torch.contiguous_format)
            db = torch.zeros(out_dim, dtype=torch.float32, device=dw.device)
            if l == 0:
                ops.weight_grad(g, ld_g, out_dim, u1, self.in1, n, dw, 0, db, self.precision)
                if u2 is not None:
                    ops.weight_grad(g, ld_g, out_dim, u2, self.in2, n, dw, self.in1, None, self.precision)
            else:
                ops.weight_grad(g, ld_g, out_dim, scratch.h[l - 1], H, n, dw, 0, db, self.precision)
            grads += [dw, db]
        if self.ln is not None:
            dgamma = torch.zeros(self.out_dim, dtype=torch.float32, device=dy.device)
            dbeta = torch.zeros_like(dgamma)
            ops.col_dot2(dy, dy.stride(0), scratch.zhat, self.out_padded, n, self.out_dim, dgamma, dbeta)
            grads += [dgamma, dbeta]
        return du1, du2, grads


class TrainPacks:
    """All node-stream MLPs of an ``EncodeProcessDecode`` packed for training."""

    def __init__(self, model):
        from .graph_network import _split_mlp
        D = model._latent_size
        self.latent = D
        prec = getattr(model, "train_precision", "fp32")
        self.precision = prec
        self.enc = _TrainMLP(*_split_mlp(model.encoder.node_model), precision=prec)
        self.rounds = [_TrainMLP(*_split_mlp(net.node_model), split_at=D, precision=prec, latent_input=True)
                       for net in model.processor]
        self.dec_acc = _TrainMLP(*_split_mlp(model.decoder_acc), precision=prec, latent_input=True)
        self.dec_tr = _TrainMLP(*_split_mlp(model.decoder_temp_rate), precision=prec, latent_input=True)
        self.all = [self.enc] + self.rounds + [self.dec_acc, self.dec_tr]
        self.hidden = self.enc.hidden
        # the (hidden, latent) pairs the kernels are compiled for (CGNN_FOR_EACH_PAIR): squares, and hidden 128 with latent
        # 64 or 256 -- the reference passes the two sizes independently (config.py:19-20, train.py:165-171)
        pair_ok = (self.hidden == D and D in (32, 64, 128, 256)) or (self.hidden == 128 and D in (64, 256))
        for m in self.all:
            if m.hidden != self.hidden or not pair_ok:
                raise CgnnError(f"training kernels are built for mlp_hidden_size == latent_size in (32, 64, 128, 256) and for "
                                f"mlp_hidden_size 128 with latent_size 64 or 256; got hidden {m.hidden}, latent {D}")
        if self.enc.in1 > 64 or (self.enc.in1 > 32 and (self.hidden != D or D < 64)):
            raise CgnnError(f"training kernels take at most 64 node input features (window_size <= 16), more than 32 only with "
                            f"mlp_hidden_size == latent_size >= 64 (got {self.enc.in1} features, hidden {self.hidden}, latent {D})")
        self.nh = self.enc.nh
        self.edge_stream_fn = None      # set per call by EncodeProcessDecode._forward_train when model.train_edge_stream

    def params(self) -> List[torch.Tensor]:
        return [p for m in self.all for p in m.params()]


class _NodeStream(torch.autograd.Function):
    """acceleration, temp_rate = f(x0; node-stream parameters).  Non-tensor context first, then ``x0`` and the
    parameters (so autograd routes one gradient to each)."""

    @staticmethod
    def forward(ctx, packs: TrainPacks, graph, x0: torch.Tensor, *params: torch.Tensor):
        src, dst, fixed_k, _ = graph
        n = x0.shape[0]
        xs = [ops.mlp_rows(packs.enc.fwd, x0)]      # raw features: keep the f32 exponent range (three bf16 terms)
        aggs = []                   # kept for the backward (N x D x 4 bytes per round; recomputing them cost 6 % of a step)
        plan = ops.AggregatePlan.of(src, n, fixed_k, xs[0].shape[1]) if fixed_k > 0 else None
        for r in packs.rounds:
            x = xs[-1]
            agg = ops.aggregate(x, src, dst, n, fixed_k, src.numel(), plan=plan)
            aggs.append(agg)
            xs.append(ops.node_block(r.run, r.run.layers[0], r.run2, x, agg, None, residual=True))
        acc = ops.mlp_rows(packs.dec_acc.run, xs[-1])
        tr = ops.mlp_rows(packs.dec_tr.run, xs[-1])
        if packs.edge_stream_fn is not None:      # model.train_edge_stream: the (dead) edge stream, for like-for-like step times
            packs.edge_stream_fn(xs)
        ctx.packs, ctx.graph, ctx.x0, ctx.xs, ctx.aggs = packs, graph, x0, xs, aggs
        return acc, tr

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_acc, d_tr):
        packs, (src, dst, fixed_k, by_sender), x0, xs = ctx.packs, ctx.graph, ctx.x0, ctx.xs
        n, D = x0.shape[0], packs.latent
        scratch = ops.BackwardScratch(n, packs.hidden, max(D, 32), packs.nh, x0.device)
        grads_of = {}
        xl = xs[-1]
        zero = lambda t, w: torch.zeros((n, w), dtype=torch.float32, device=x0.device) if t is None else t  # noqa: E731
        dx, _, grads_of[id(packs.dec_acc)] = packs.dec_acc.backward(xl, None, zero(d_acc, packs.dec_acc.out_dim),
                                                                    scratch, True)
        dx2, _, grads_of[id(packs.dec_tr)] = packs.dec_tr.backward(xl, None, zero(d_tr, packs.dec_tr.out_dim),
                                                                   scratch, True)
        dx = dx.add_(dx2)
        for i in range(len(packs.rounds) - 1, -1, -1):
            r, x = packs.rounds[i], xs[i]
            agg = ctx.aggs[i]
            ctx.aggs[i] = None
            du1, du2, grads_of[id(r)] = r.backward(x, agg, dx, scratch, True, True)
            # x_{i+1} = x_i + f(x_i, agg(x_i)):  dx_i = dx_{i+1} + du1 + A^T du2     (A^T: senders <- receivers)
            dx = ops.aggregate_csr(du2, by_sender, out=dx, add1=dx, add2=du1)      # one pass instead of three
        need_dx0 = ctx.needs_input_grad[2]
        dx0, _, grads_of[id(packs.enc)] = packs.enc.backward(x0, None, dx, scratch, need_dx0)
        flat = [g for m in packs.all for g in grads_of[id(m)]]
        ctx.xs = None
        return (None, None, dx0, *flat)


def edge_stream_of(model, xs: Sequence[torch.Tensor], src: torch.Tensor, dst: torch.Tensor, fixed_k: int,
                   edge_attr: torch.Tensor, node_in: int):
    """The edge stream the training forward otherwise skips (under the reference nothing reads it, but its
    forward is computed -- reference graph_network.py:89-90, :182): the inference kernels at the model's ``edge_precision``
    on the node latents ``xs[i]`` of every round.  No gradient passes through it.  Returns the final edge latents."""
    from . import graph_network as gn
    with torch.no_grad():
        P = model._pack(node_in, edge_attr.shape[1])
        rounds = P["rounds"]
        n = xs[0].shape[0]
        image = P["image"]
        if image is not None:
            H = rounds[0].ws.out_dim
            ps_all = torch.empty((len(rounds), n, H), dtype=torch.bfloat16, device=xs[0].device)
            pd_all = torch.empty_like(ps_all)
            for i, p in enumerate(rounds):
                ops.project_nodes(p.ws, p.wd, xs[i], ps_all[i], pd_all[i], p.p_format)
            img, kernel = model._edge_stream_plan(P, fixed_k, src.numel(), edge_attr)
            e0 = None if img.enc_in else ops.mlp_rows(P["enc_edge"], edge_attr, tiled=True)
            return ops.edge_stream_run(img, ps_all, pd_all, src, dst, e0, e0, edge_attr if img.enc_in else None,
                                       kernel=kernel, lag=int(getattr(model, "edge_stream_lag", 0)), fixed_k=fixed_k)
        e = ops.mlp_rows(P["enc_edge"], edge_attr, tiled=True)
        for i, p in enumerate(rounds):
            ps, pd = ops.project_nodes(p.ws, p.wd, xs[i], None, None, p.p_format)
            e = ops.edge_block(p.edge, ps, pd, src, dst, e, e, None, True)
        return e


def forward_train(model, x0: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, fixed_k: int, packs: TrainPacks,
                  by_sender: "ops.SenderCsr"):
    """Differentiable ``(acceleration, temp_rate)`` for node features ``x0`` (already float32, contiguous, on the
    device, in the kernels' particle order).  ``by_sender``: ``ops.SenderCsr(src, dst, n)``, the transposed
    adjacency the backward of the aggregation gathers through."""
    return _NodeStream.apply(packs, (src, dst, fixed_k, by_sender), x0, *packs.params())


class _Permute(torch.autograd.Function):
    """``out = rows[idx]`` for a permutation ``idx`` with inverse ``inv`` (HIP row gather both ways)."""

    @staticmethod
    def forward(ctx, rows, idx, inv):
        ctx.inv = inv
        return ops.gather_rows(rows.contiguous(), idx)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        return ops.gather_rows(d_out.contiguous(), ctx.inv), None, None


def permute_rows(rows: torch.Tensor, idx: torch.Tensor, inv: torch.Tensor) -> torch.Tensor:
    return _Permute.apply(rows, idx, inv)

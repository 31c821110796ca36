"""Developer tool: time each hot-path op alone at a given size (HIP events).  Not part of the product or tests.
    python scripts/time_ops.py [--particles 1000000] [--latent 128] [--edge-precision bf16]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmology_gnn_simulation_amd import _lib  # noqa: E402
if os.environ.get("CGNN_LIB_PATH"):      # developer A/B: time another build of the library (scripts/ab/)
    _lib.LIB_PATH = os.environ["CGNN_LIB_PATH"]
from cosmology_gnn_simulation_amd import data_utils, graph_network, ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1_000_000)
ap.add_argument("--neighbors", type=int, default=16)
ap.add_argument("--latent", type=int, default=128)
ap.add_argument("--edge-precision", default="bf16")
ap.add_argument("--node-precision", default="fp32")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--mp-steps", type=int, default=10, help="rounds for the edge_stream line")
ap.add_argument("--only", default=None, help="time just this op")
ap.add_argument("--stream-kernel", default="tile32", choices=["tile32", "tile16"])
a = ap.parse_args()
dev = "cuda"
n, k, d = a.particles, a.neighbors, a.latent
snap = synthetic.make_snapshot(n, seed=1236)
meta = synthetic.make_metadata()
g = data_utils.preprocess(snap["Coordinates"][:5], snap["InternalEnergy"][:5], meta, None, None, 0.0, k, 0.01, 1.0)
m = graph_network.EncodeProcessDecode(d, d, 2, 1, 3)
m.load_state_dict(synthetic.make_state_dict(d, d, 2, 1, 3))
m = m.to(dev).eval()
m.edge_precision, m.node_precision = a.edge_precision, a.node_precision
m.fuse_rounds = False          # the single-round ops below use the per-round packing
src, dst, fk = graph_network._graph_arrays(g, n)
order, inv, src, dst = graph_network._locality_plan(g, n, fk, src)
P = m._pack(17, 4)
p = P["rounds"][0]
x = torch.randn(n, d, device=dev)
e = ops.TiledRows.from_rows(torch.randn(n * k, d, device=dev))
ea = torch.randn(n * k, 4, device=dev)
xf = torch.randn(n, 17, device=dev)
ps, pd = ops.project_nodes(p.ws, p.wd, x, None, None, p.p_format)
agg = ops.aggregate(x, src, dst, n, fk)


def t(name, fn, nbytes=None, flops=None):
    if a.only and not name.startswith(a.only):      # --only takes a prefix
        return
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.iters * 1e3
    extra = ""
    if nbytes:
        extra += f"  {nbytes / ms / 1e6:8.1f} GB/s(alg)"
    if flops:
        extra += f"  {flops / ms / 1e9:8.1f} TFLOP/s(exec)"
    print(f"{name:16s} {ms:8.3f} ms{extra}", flush=True)


E = n * k
L = a.mp_steps
mL = graph_network.EncodeProcessDecode(d, d, 2, L, 3)
mL.load_state_dict(synthetic.make_state_dict(d, d, 2, L, 3))
mL = mL.to(dev).eval()
mL.edge_precision, mL.node_precision, mL.edge_stream_kernel = a.edge_precision, a.node_precision, a.stream_kernel
PL_ = mL._pack(17, 4)
roundsL = PL_["rounds"]
if a.edge_precision == "bf16" and (not a.only or a.only.startswith("edge_stream")):
    ps_all = torch.randn(L, n, d, device=dev).to(torch.bfloat16)
    pd_all = torch.randn(L, n, d, device=dev).to(torch.bfloat16)
    enc_flops = 2.0 * E * (32 * d + 2 * d * d)
    if PL_["image"] is not None:        # cgnn_edge_stream_run (32-edge tiles)
        img0 = ops.StreamImage([r.edge for r in roundsL], None)
        t("edge_stream", lambda: ops.edge_stream_run(img0, ps_all, pd_all, src, dst, e, e),
          2 * E * d * 4 + 2 * E * 4 + L * 2 * n * d * 2, L * 6.0 * E * d * d)
        if PL_["image"].enc_in:
            t("edge_stream+enc", lambda: ops.edge_stream_run(PL_["image"], ps_all, pd_all, src, dst, None, e, ea),
              E * d * 4 + E * 16 + 2 * E * 4 + L * 2 * n * d * 2, L * 6.0 * E * d * d + enc_flops)
    else:
        t("edge_stream", lambda: ops.edge_stream([r.edge for r in roundsL], ps_all, pd_all, src, dst, e, e),
          2 * E * d * 4 + 2 * E * 4 + L * 2 * n * d * 2, L * 6.0 * E * d * d)
        if mL._encoder_fits_stream(PL_):
            t("edge_stream+enc", lambda: ops.edge_stream([r.edge for r in roundsL], ps_all, pd_all, src, dst, None, e,
                                                         PL_["enc_edge"], ea),
              E * d * 4 + E * 16 + 2 * E * 4 + L * 2 * n * d * 2, L * 6.0 * E * d * d + enc_flops)
t("edge_block", lambda: ops.edge_block(p.edge, ps, pd, src, dst, e, e, None, True),
  2 * E * d * 4 + 2 * E * 4 + 2 * n * d * 4, 6.0 * E * d * d)
t("aggregate x_j", lambda: ops.aggregate(x, src, dst, n, fk, E, agg), E * d * 4 + E * 4 + n * d * 4)
if ops.AggregatePlan.supported(n, fk, d) and (not a.only or a.only.startswith("aggregate")):
    t("aggregate_plan", lambda: ops.AggregatePlan(src, n, fk))
    plan_ = ops.AggregatePlan(src, n, fk)
    rows_ = 64 if fk in (8, 16) else 32
    cnt_ = plan_.blob[: 4 * ((n + rows_ - 1) // rows_)].view(torch.int32).float()
    print(f"   (plan: {float(cnt_.clamp(min=0).mean()):.0f} distinct rows per block of {rows_} receivers x k = {rows_ * fk} references, "
          f"{int(((cnt_ < 0) | (cnt_ > 352)).sum())} of {cnt_.numel()} blocks gather directly)")
    t("aggregate planned", lambda: ops.aggregate(x, src, dst, n, fk, E, agg, plan=plan_), E * d * 4 + E * 4 + n * d * 4)
# general scatter-add (fixed_k = 0): receiver-sorted list (one atomic row per receiver) and a shuffled one (one per edge);
# GB/s = table rows read + atomic bytes added
if not a.only or a.only == "aggregate_atomic":
    perm = torch.randperm(E, device=dev)
    src_sh, dst_sh = src[perm].contiguous(), dst[perm].contiguous()
    _o = a.only
    a.only = None
    t("scatter sorted", lambda: ops.aggregate(x, src, dst, n, 0, E, agg), E * d * 4 + E * 8 + n * d * 4)
    t("scatter shuffled", lambda: ops.aggregate(x, src_sh, dst_sh, n, 0, E, agg), E * d * 4 + E * 8 + E * d * 4)
    print(f"   (shuffled: {E * d * 4 / 1e9:.2f} GB of float atomics per call)")
    a.only = _o
t("node_block", lambda: ops.node_block(p.node, p.wx, p.wa, x, agg, x, True), 3 * n * d * 4, 8.0 * n * d * d)
if len(roundsL) > 1 and roundsL[0].node.precision in _lib.N16_NODE:      # 16-row node kernels: projections of the next round fused in
    q_ = roundsL[1]
    # ... in the table format the forward's edge stream takes (fp16 rows for the two-waves-per-SIMD kernel)
    kern_ = mL._edge_stream_plan(PL_, fk, E, ea)[1] if PL_["image"] is not None else None
    fmt_ = graph_network.stream_table_format(roundsL, kern_, int(getattr(mL, "edge_stream_lag", 0)))
    psf_, pdf_ = ps.to(ops.p_format_dtype(fmt_)), pd.to(ops.p_format_dtype(fmt_))
    t("node_block+proj", lambda: ops.node_block(roundsL[0].node, roundsL[0].wx, roundsL[0].wa, x, agg, x, True,
                                                (q_.ws_fused, q_.wd_fused, psf_, pdf_, fmt_)),
      3 * n * d * 4 + 2 * n * d * 2, 8.0 * n * d * d)
t("project_nodes", lambda: ops.project_nodes(p.ws, p.wd, x, ps, pd, p.p_format), 3 * n * d * 4, 4.0 * n * d * d)
t("enc_edge", lambda: ops.mlp_rows(P["enc_edge"], ea, out=e), E * (16 + d * 4), 2.0 * E * (32 * d + 2 * d * d))
t("enc_node", lambda: ops.mlp_rows(P["enc_node"], xf, out=x), n * (68 + d * 4), 2.0 * n * (32 * d + 2 * d * d))
t("dec_acc", lambda: ops.mlp_rows(P["dec_acc"], x), n * (d * 4 + 12), 2.0 * n * (2 * d * d + 32 * d))
t("knn", lambda: ops.knn_periodic(g.pos, 1.0, k), n * 12 + E * 20)

"""Developer tool for profiling: run ONE op a few times at a given size (so that a rocprofv3 --pmc pass
attributes counters to that kernel only).  python scripts/run_one_op.py edge_block --particles 1000000"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmology_gnn_simulation_amd import _lib  # noqa: E402
if os.environ.get("CGNN_LIB_PATH"):      # developer A/B: time another build of the library (scripts/ab/)
    _lib.LIB_PATH = os.environ["CGNN_LIB_PATH"]
from cosmology_gnn_simulation_amd import data_utils, graph_network, ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("op", choices=["edge_block", "edge_stream", "aggregate", "aggregate_planned", "scatter_shuffled", "node_block", "node_block_proj", "project_nodes", "enc_edge", "knn"])
ap.add_argument("--particles", type=int, default=1_000_000)
ap.add_argument("--neighbors", type=int, default=16)
ap.add_argument("--latent", type=int, default=128)
ap.add_argument("--edge-precision", default="bf16")
ap.add_argument("--node-precision", default="fp32")
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--real-graph", action="store_true", help="build the periodic k-NN graph (slower under --pmc)")
a = ap.parse_args()
dev = "cuda"
n, k, d = a.particles, a.neighbors, a.latent
gen = torch.Generator(device=dev).manual_seed(0)
if a.op == "knn":
    pos = torch.rand(n, 3, device=dev, generator=gen)
m = graph_network.EncodeProcessDecode(d, d, 2, 1, 3)
m.load_state_dict(synthetic.make_state_dict(d, d, 2, 1, 3))
m = m.to(dev).eval()
m.edge_precision, m.node_precision = a.edge_precision, a.node_precision
# synthetic spatially-local graph (no k-NN build under the profiler): senders within +-4096 rows of the receiver
fk = k
if a.real_graph:
    pos = torch.rand(n, 3, device=dev, generator=gen)
    snd, _, order = ops.knn_periodic(pos, 1.0, k, want_edge_attr=False, want_order=True)
    g = graph_network.Data(edge_index=torch.stack([snd.long(), torch.arange(n, device=dev).repeat_interleave(k)]))
    g._cgnn_order, g._cgnn_fixed_k = order, k
    src0, dst0, _ = graph_network._graph_arrays(g, n)
    _, _, src, dst = graph_network._locality_plan(g, n, k, src0)
else:
    dst = torch.arange(n, device=dev, dtype=torch.int32).repeat_interleave(k)
    src = ((dst.long() + torch.randint(-4096, 4097, (n * k,), device=dev, generator=gen)) % n).to(torch.int32)
P = m._pack(17, 4)
p = P["rounds"][0]
x = torch.randn(n, d, device=dev, generator=gen)
e = ops.TiledRows.from_rows(torch.randn(n * k, d, device=dev, generator=gen))
ea = torch.randn(n * k, 4, device=dev, generator=gen)
ps, pd = ops.project_nodes(p.ws, p.wd, x, None, None, p.p_format)
agg = ops.aggregate(x, src, dst, n, fk)
if a.op == "edge_stream":
    L = 10
    mL = graph_network.EncodeProcessDecode(d, d, 2, L, 3)
    mL.load_state_dict(synthetic.make_state_dict(d, d, 2, L, 3))
    mL = mL.to(dev).eval()
    mL.edge_precision, mL.node_precision = a.edge_precision, a.node_precision
    mL.edge_stream_kernel = os.environ.get("CGNN_RUN_STREAM_KERNEL", "tile32w")
    PL = mL._pack(17, 4)
    roundsL = [r.edge for r in PL["rounds"]]
    imageL = PL["image"]
    kernL = "tile32"
    if imageL is not None:
        imageL, kernL = mL._edge_stream_plan(PL, fk, n * k, ea)
    encL = None if imageL is not None else (PL["enc_edge"] if mL._encoder_fits_stream(PL) else None)
    # the table format the model itself would hand this kernel (fp16 rows for the two-waves-per-SIMD kernel)
    pdtL = ops.p_format_dtype(graph_network.stream_table_format(PL["rounds"], kernL if imageL is not None else None, mL.edge_stream_lag))
    ps_all = torch.randn(L, n, d, device=dev, generator=gen).to(pdtL)
    pd_all = torch.randn(L, n, d, device=dev, generator=gen).to(pdtL)
if a.op == "node_block_proj":     # the node block as the fused forward runs it: next round's projections inside
    m2 = graph_network.EncodeProcessDecode(d, d, 2, 2, 3)
    m2.load_state_dict(synthetic.make_state_dict(d, d, 2, 2, 3))
    m2 = m2.to(dev).eval()
    m2.edge_precision, m2.node_precision = a.edge_precision, a.node_precision
    P2 = m2._pack(17, 4)
    r0, r1 = P2["rounds"]
    # ... in the table format of the forward's edge stream (graph_network.stream_table_format)
    kern2 = m2._edge_stream_plan(P2, fk, n * k, ea)[1] if P2["image"] is not None else None
    fmt2 = graph_network.stream_table_format(P2["rounds"], kern2, int(getattr(m2, "edge_stream_lag", 0)))
    ps2, pd2 = ops.project_nodes(r1.ws, r1.wd, x, None, None, fmt2)
if a.op == "aggregate_planned":
    plan_ = ops.AggregatePlan(src, n, fk)
if a.op == "scatter_shuffled":      # general edge list (fixed_k = 0), shuffled: one float atomic row per edge
    perm_ = torch.randperm(n * k, device=dev, generator=gen)
    src_sh, dst_sh = src[perm_].contiguous(), dst[perm_].contiguous()
fn = {
    "aggregate_planned": lambda: ops.aggregate(x, src, dst, n, fk, n * k, agg, plan=plan_),
    "scatter_shuffled": lambda: ops.aggregate(x, src_sh, dst_sh, n, 0, n * k, agg),
    "node_block_proj": lambda: ops.node_block(r0.node, r0.wx, r0.wa, x, agg, x, True,
                                              (r1.ws_fused, r1.wd_fused, ps2, pd2, fmt2)),
    "edge_stream": lambda: (ops.edge_stream_run(imageL, ps_all, pd_all, src, dst, None if imageL.enc_in else e, e,
                                                ea if imageL.enc_in else None, kernel=kernL, lag=mL.edge_stream_lag,
                                                fixed_k=fk) if imageL is not None else
                            ops.edge_stream(roundsL, ps_all, pd_all, src, dst, None if encL else e, e, encL,
                                            ea if encL else None)),
    "edge_block": lambda: ops.edge_block(p.edge, ps, pd, src, dst, e, e, None, True),
    "aggregate": lambda: ops.aggregate(x, src, dst, n, fk, n * k, agg),
    "node_block": lambda: ops.node_block(p.node, p.wx, p.wa, x, agg, x, True),
    "project_nodes": lambda: ops.project_nodes(p.ws, p.wd, x, ps, pd, p.p_format),
    "enc_edge": lambda: ops.mlp_rows(P["enc_edge"], ea, out=e),
    "knn": lambda: ops.knn_periodic(pos, 1.0, k),
}[a.op]
torch.cuda.synchronize()
for _ in range(a.iters):
    fn()
torch.cuda.synchronize()
print("done", a.op)

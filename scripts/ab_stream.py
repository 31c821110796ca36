"""Developer tool: the one-launch edge-stream kernels side by side in ONE process, interleaved rounds (same device, same
clock state): cgnn_edge_stream_run ("tile32") and cgnn_edge_stream_run_w8 ("tile32w", lag 0 / 1), with the edge encoder in
the launch, at a BASELINE shape.  Prints median / min per variant and the issued-flops fraction of the bf16 MFMA peak.
    python scripts/ab_stream.py [--particles 1000000] [--neighbors 16] [--mp-steps 10] [--rounds 5]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmology_gnn_simulation_amd import _lib  # noqa: E402
if os.environ.get("CGNN_LIB_PATH"):
    _lib.LIB_PATH = os.environ["CGNN_LIB_PATH"]
from cosmology_gnn_simulation_amd import data_utils, graph_network, ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1_000_000)
ap.add_argument("--neighbors", type=int, default=16)
ap.add_argument("--latent", type=int, default=128)
ap.add_argument("--mp-steps", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variants", default="tile32,tile32w:0,tile32w:1")
ap.add_argument("--no-encoder", action="store_true")
ap.add_argument("--bf16-tables", action="store_true", help="tile32w: bf16 P tables (selector MFMAs) instead of the model's fp16 ones")
ap.add_argument("--fixed-k", type=int, default=1, help="0: do not tell the kernel about the graph's fixed in-degree")
a = ap.parse_args()
dev = "cuda"
n, k, d, L = a.particles, a.neighbors, a.latent, a.mp_steps
snap = synthetic.make_snapshot(n, seed=1236)
meta = synthetic.make_metadata()
g = data_utils.preprocess(snap["Coordinates"][:5], snap["InternalEnergy"][:5], meta, None, None, 0.0, k, 0.01, 1.0)
src, dst, fk = graph_network._graph_arrays(g, n)
order, inv, src, dst = graph_network._locality_plan(g, n, fk, src)
m = graph_network.EncodeProcessDecode(d, d, 2, L, 3)
m.load_state_dict(synthetic.make_state_dict(d, d, 2, L, 3))
m = m.to(dev).eval()
m.edge_precision, m.node_precision = "bf16", "fp16x2"
P = m._pack(17, 4)
parts = ([r.edge for r in P["rounds"]], None if a.no_encoder else P["enc_edge"])
images = {k: ops.StreamImage(*parts, kernel=k) for k in ("tile32", "tile32w")}
E = n * k
ps_all = torch.randn(L, n, d, device=dev).to(torch.bfloat16)
pd_all = torch.randn(L, n, d, device=dev).to(torch.bfloat16)
# what the model feeds the two-waves-per-SIMD kernel (lag 0): the same tables in fp16 (CGNN_P_F16_S32)
ps16, pd16 = ps_all.to(torch.float16), pd_all.to(torch.float16)
ea = torch.randn(E, 4, device=dev)
e = ops.TiledRows.from_rows(torch.randn(E, d, device=dev))
flops = L * 6.0 * E * d * d + (0 if a.no_encoder else 2.0 * E * (32 * d + 2 * d * d))
variants = []
for v in a.variants.split(","):
    kern, _, lag = v.partition(":")      # "tile32w:0f": timing of the CGNN_STREAM_FOLDED kernel (on an image that is not folded)
    variants.append((v, kern, (int(lag.rstrip("f") or 0), lag.endswith("f"))))


def run(kern, lag_fold):
    lag, fold = lag_fold
    ps_, pd_ = (ps16, pd16) if (kern == "tile32w" and (lag == 2 or (lag == 0 and not a.bf16_tables))) else (ps_all, pd_all)
    if a.no_encoder:
        ops.edge_stream_run(images[kern], ps_, pd_, src, dst, e, e, None, kernel=kern, lag=lag, fixed_k=fk if a.fixed_k else 0, folded=fold)
    else:
        ops.edge_stream_run(images[kern], ps_, pd_, src, dst, None, e, ea, kernel=kern, lag=lag, fixed_k=fk if a.fixed_k else 0, folded=fold)


times = {v[0]: [] for v in variants}
for name, kern, lag in variants:
    run(kern, lag)
torch.cuda.synchronize()
for _ in range(a.rounds):
    for name, kern, lag in variants:
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        run(kern, lag)
        t1.record()
        torch.cuda.synchronize()
        times[name].append(t0.elapsed_time(t1))
for name, ts in times.items():
    med = statistics.median(ts)
    print(f"{name:12s} median {med:8.3f} ms  min {min(ts):8.3f} ms  {flops / med / 1e9:7.1f} TFLOP/s issued = "
          f"{flops / med / 1e9 / 2500:.3f} of the bf16 peak   ({E * L / med / 1e6:.2f} G edge-updates/s)", flush=True)

"""Developer tool: time one training step (forward + backward + Adam) through the HIP node-stream kernels and
print the per-op split (HIP events on the launch stream).  Not part of the product or tests.
    python scripts/time_train.py [--particles 1000000] [--latent 128] [--mp-steps 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmology_gnn_simulation_amd import data_utils, graph_network, losses, ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1_000_000)
ap.add_argument("--neighbors", type=int, default=16)
ap.add_argument("--latent", type=int, default=128)
ap.add_argument("--mp-steps", type=int, default=10)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--train-precision", default="fp32", choices=["fp32", "fp32x3"])
ap.add_argument("--hidden", type=int, default=None, help="mlp_hidden_size (default: the latent size)")
ap.add_argument("--with-edge-stream", action="store_true",
                help="also run the edge stream's forward (model.train_edge_stream): under the reference nothing reads it "
                     "(SURVEY F1) but its step computes it -- the like-for-like step time")
a = ap.parse_args()
dev = "cuda"
n, k, d, L = a.particles, a.neighbors, a.latent, a.mp_steps
snap = synthetic.make_snapshot(n, seed=1236)
meta = synthetic.make_metadata()
c, e = snap["Coordinates"], snap["InternalEnergy"]
g = data_utils.preprocess(c[:5], e[:5], meta, c[5], e[5], 0.0, k, 0.01, 1.0)
hd = a.hidden or d
m = graph_network.EncodeProcessDecode(d, hd, 2, L, 3)
m.load_state_dict(synthetic.make_state_dict(d, hd, 2, L, 3))
m = m.to(dev).train()
m.train_precision = a.train_precision
m.train_edge_stream = a.with_edge_stream
if a.with_edge_stream:
    m.edge_precision, m.node_precision = "bf16", "fp16x2"      # bench.py's edge stream
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
mse = torch.nn.functional.mse_loss


def step():
    pred = m(g)
    loss = (mse(pred["acceleration"], g.y_acc) + mse(pred["temp_rate"], g.y_temp_rate)
            + losses.momentum_conservation_loss(pred["acceleration"], g, 0.01, 0.1))
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.iters * 1e3
print(f"training step ({'edge stream forward included' if a.with_edge_stream else 'edge stream skipped (F1)'}): {ms:.2f} ms  "
      f"({n} particles, k={k}, latent {d}, hidden {hd}, {L} rounds; "
      f"{n * k * L / ms / 1e6:.3f} G edge-updates/s)", flush=True)
with ops.OpTimer() as tm:
    step()
for name, (calls, total) in sorted(tm.summary().items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:16s} {calls:4d} calls {total:9.3f} ms", flush=True)
print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")

"""Developer tool: on-device autoregressive rollout (rollout.py; reference render_rollout.py:26-90) timed at a BASELINE
size: steps per second, and the share of a step spent in the k-NN graph build.
    python scripts/time_rollout.py [--particles 262144] [--steps 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmology_gnn_simulation_amd import graph_network, ops, rollout, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=262_144)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--neighbors", type=int, default=16)
ap.add_argument("--latent", type=int, default=128)
ap.add_argument("--mp-steps", type=int, default=10)
ap.add_argument("--window", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda")
n, Wn = a.particles, a.window
snap = synthetic.make_snapshot(n, window=Wn + 1, seed=1235)
meta = synthetic.make_metadata()
m = graph_network.EncodeProcessDecode(a.latent, a.latent, 2, a.mp_steps, 3)
m.load_state_dict(synthetic.make_state_dict(a.latent, a.latent, 2, a.mp_steps, 3, node_in=3 * (Wn - 1) + Wn))
m = m.to(dev).eval()
m.edge_precision, m.node_precision = "bf16", "fp32x3"
data = {"Coordinates": snap["Coordinates"].to(dev), "InternalEnergy": snap["InternalEnergy"].to(dev)}
rollout.rollout(m, data, meta, 0.0, meta["dt"], meta["box_size"], window_size=Wn, num_neighbors=a.neighbors, num_steps=2)
torch.cuda.synchronize()
with ops.OpTimer() as tm:
    t0 = time.perf_counter()
    out = rollout.rollout(m, data, meta, 0.0, meta["dt"], meta["box_size"], window_size=Wn, num_neighbors=a.neighbors,
                          num_steps=a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
summ = tm.summary()
knn = summ["knn_periodic"][1] / a.steps
print(f"rollout: {n} particles, k={a.neighbors}, latent {a.latent}, {a.mp_steps} rounds: {dt / a.steps * 1e3:.2f} ms/step "
      f"= {a.steps / dt:.1f} steps/s; k-NN graph build {knn:.2f} ms/step ({knn / (dt / a.steps * 1e3) * 100:.0f} %); "
      f"edge updates/s incl. graph build {n * a.neighbors * a.mp_steps * a.steps / dt / 1e9:.2f} G")
for name, (c, ms) in sorted(summ.items()):
    print(f"   {name:16s} {c / a.steps:5.1f} calls/step {ms / a.steps:8.3f} ms/step")

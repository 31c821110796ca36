"""Developer experiment: do the node kernel and the aggregation kernel overlap when launched on two HIP streams?
    python scripts/stream_overlap_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmology_gnn_simulation_amd import graph_network, ops, synthetic  # noqa: E402

dev = "cuda"
n, k, d, L = 1_000_000, 16, 128, 2
gen = torch.Generator(device=dev).manual_seed(0)
m = graph_network.EncodeProcessDecode(d, d, 2, L, 3)
m.load_state_dict(synthetic.make_state_dict(d, d, 2, L, 3))
m = m.to(dev).eval()
m.edge_precision, m.node_precision = "bf16", "fp32x3"
P = m._pack(17, 4)
p, q = P["rounds"][0], P["rounds"][1]
dst = torch.arange(n, device=dev, dtype=torch.int32).repeat_interleave(k)
src = ((dst.long() + torch.randint(-4096, 4097, (n * k,), device=dev, generator=gen)) % n).to(torch.int32)
x = torch.randn(n, d, device=dev, generator=gen)
x2 = torch.empty_like(x)
agg = ops.aggregate(x, src, dst, n, k)
agg2 = torch.empty_like(agg)
ps = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
pd = torch.empty_like(ps)
nxt = (q.ws_fused, q.wd_fused, ps, pd, q.p_format)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def node():
    ops.node_block(p.node, p.wx, p.wa, x, agg, x2, True, nxt)


def aggr():
    ops.aggregate(x, src, dst, n, k, n * k, agg2)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def both():
    with torch.cuda.stream(s1):
        node()
    with torch.cuda.stream(s2):
        aggr()


def both_rev():
    with torch.cuda.stream(s2):
        aggr()
    with torch.cuda.stream(s1):
        node()


print(f"node alone       {timeit(node):.3f} ms")
print(f"aggregate alone  {timeit(aggr):.3f} ms")
print(f"sequential       {timeit(lambda: (node(), aggr())):.3f} ms")
print(f"two streams      {timeit(both):.3f} ms")
print(f"two streams (agg first) {timeit(both_rev):.3f} ms")

"""GPU: the LDS-ring kernels (node_block_f2.hip, edge_block_f2.hip, edge_block_ring256.hip) keep several LDS-DMA chunks,
row prefetches and stores in flight behind hand-counted `s_waitcnt vmcnt(N)` waits.  A wrong count does not fail every
time: it reads a chunk or a row that has not landed yet, now and then, depending on memory load.  These tests run
each kernel many times on the same inputs, at sizes with one and with many steps per workgroup and while another stream
keeps the memory system busy, and require bit-identical results every time (and agreement with the independent
kernels of the same arithmetic)."""
import numpy as np
import pytest
import torch

from cosmology_gnn_simulation_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mlp(gen, fin, d, nh):
    dims = [fin] + [d] * nh + [d]
    lin = []
    for i in range(nh + 1):
        bound = 1.0 / np.sqrt(dims[i])
        lin.append((((torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * bound).to(DEV),
                    ((torch.rand(dims[i + 1], generator=gen) * 2 - 1) * bound).to(DEV)))
    ln = ((1 + 0.1 * torch.randn(d, generator=gen)).to(DEV), (0.1 * torch.randn(d, generator=gen)).to(DEV))
    return lin, ln


class _Noise:
    """A second stream streaming a large buffer back and forth: changes the timing of every memory operation."""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.a = torch.empty(64 << 20, device=DEV)
        self.b = torch.empty_like(self.a)

    def burst(self, n=4):
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                self.b.copy_(self.a)
                self.a.copy_(self.b)


@pytest.mark.parametrize("n", [128 * 256, 128 * 256 * 3 + 64, 128 * 7 + 5, 1_000_000])
def test_node_ring_kernel_repeats_bit_identically(n):
    d, nh = 128, 2
    gen = torch.Generator().manual_seed(n % 1000)
    lin, ln = _mlp(gen, 2 * d, d, nh)
    w1, b1 = lin[0]
    x = torch.randn(n, d, generator=gen).to(DEV)
    agg = (torch.randn(n, d, generator=gen) * 4).to(DEV)
    w1e = ((torch.rand(d, 3 * d, generator=gen) * 2 - 1) / np.sqrt(3 * d)).to(DEV)
    b1e = torch.zeros(d, device=DEV)
    ws16, wd16 = ops.PackedLinear(w1e, None, "bf16_n16", 0, d), ops.PackedLinear(w1e, b1e, "bf16_n16", d, d)
    wx, wa = ops.PackedLinear(w1, b1, "fp16x2_n16", 0, d), ops.PackedLinear(w1, None, "fp16x2_n16", d, d)
    mlp = ops.PackedMLP([(w1[:, :d].contiguous(), None)] + lin[1:], ln, "fp16x2_n16")
    noise = _Noise()

    def run():
        ps = torch.empty(n, d, dtype=torch.bfloat16, device=DEV)
        pd = torch.empty_like(ps)
        out = ops.node_block(mlp, wx, wa, x, agg, None, True, (ws16, wd16, ps, pd, 1))
        return out, ps, pd

    first = run()
    torch.cuda.synchronize()
    for it in range(12 if n < 500_000 else 6):
        if it % 2:
            noise.burst()
        got = run()
        torch.cuda.synchronize()
        for a, b in zip(first, got):
            assert torch.equal(a, b), f"run {it} differs"
    # the rows of whole 128-row steps (ring kernel) and the same rows pushed through the remainder kernel alone
    m = min(n, 96)
    small = ops.node_block(mlp, wx, wa, x[:m].contiguous(), agg[:m].contiguous(), None, True)
    assert torch.equal(small, first[0][:m])


@pytest.mark.parametrize("fmt,d,k", [("fp16x2_n16", 128, 16), ("bf16_n16", 256, 32)])
@pytest.mark.parametrize("n", [128 * 16, 70_001, 300_000])
def test_edge_ring_kernels_repeat_bit_identically(fmt, d, k, n):
    if d == 256 and n == 300_000:
        n = 150_000
    gen = torch.Generator().manual_seed(n % 997 + d)
    E = n * k
    lin, ln = _mlp(gen, 3 * d, d, 2)
    w1, b1 = lin[0]
    x = torch.randn(n, d, generator=gen).to(DEV)
    e = torch.randn(E, d, generator=gen).to(DEV)
    src = torch.randint(0, n, (E,), generator=gen).int().to(DEV)
    dst = torch.arange(n).repeat_interleave(k).int().to(DEV)
    mlp = ops.PackedMLP(lin, ln, fmt, first_layer_cols=(2 * d, d))
    pprec = "fp32" if fmt == "fp16x2_n16" else "bf16"
    ws, wd = ops.PackedLinear(w1, None, pprec, 0, d), ops.PackedLinear(w1, b1, pprec, d, d)
    ps, pd = ops.project_nodes(ws, wd, x, None, None, ops.p_table_format(mlp.precision))
    et = ops.TiledRows.from_rows(e)
    noise = _Noise()
    first = ops.edge_block(mlp, ps, pd, src, dst, et, None, None, True).buf.clone()
    torch.cuda.synchronize()
    for it in range(10):
        if it % 2:
            noise.burst()
        got = ops.edge_block(mlp, ps, pd, src, dst, et, None, None, True).buf
        torch.cuda.synchronize()
        assert torch.equal(got, first), f"run {it} differs"


@pytest.mark.parametrize("n,nh", [(50, 2), (128, 2), (1000, 1), (4099, 3), (70_003, 2)])
def test_edge_encoder_at_latent_256_through_the_ring(n, nh):
    """Reference graph_network.py:57 at latent = hidden = 256: cgnn_mlp_rows with CGNN_BF16_N16 weights streams them through
    the LDS ring of the 256-wide edge kernel (edge_block_ring256.hip).  Against a torch emulation of its arithmetic (bf16
    operands, wide accumulation, f32 LayerNorm), and against the 32-row kernel on the same bf16 weights."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(256 + n + nh)
    d = 256
    dims = [4] + [d] * nh + [d]
    lin = []
    for i in range(nh + 1):
        bound = 1.0 / dims[i] ** 0.5
        lin.append((((torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * bound).to(DEV),
                    ((torch.rand(dims[i + 1], generator=gen) * 2 - 1) * bound).to(DEV)))
    ln = ((1 + 0.1 * torch.randn(d, generator=gen)).to(DEV), (0.1 * torch.randn(d, generator=gen)).to(DEV))
    attr = (torch.randn(n, 4, generator=gen) * 0.5).to(DEV)
    bf = lambda t: t.bfloat16().float()                                        # noqa: E731
    dot = lambda a, w: (bf(a).double() @ bf(w).double().t()).float()           # noqa: E731
    h = dot(attr, lin[0][0]) + lin[0][1]
    for w, b in lin[1:]:
        h = dot(torch.relu(h), w) + b
    want = F.layer_norm(h, (d,), ln[0], ln[1], 1e-5)
    ring = ops.mlp_rows(ops.PackedMLP(lin, ln, "bf16_n16"), attr, tiled=True).to_rows()
    rows32 = ops.mlp_rows(ops.PackedMLP(lin, ln, "bf16"), attr, tiled=True).to_rows()
    torch.cuda.synchronize()
    assert ring.shape == want.shape and bool(torch.isfinite(ring).all())
    scale = float(want.abs().max())
    for got in (ring, rows32):
        assert float((got - want).abs().max()) <= 1e-2 * scale
        assert float((got - want).norm() / want.norm()) <= 1e-3
    # every row, not just the norm: one wrong 16-edge tile must not hide among 70,000 edges
    per_row = (ring - want).norm(dim=1) / want.norm(dim=1)
    assert float(per_row.max()) <= 1e-2, int(per_row.argmax())
    # a second run returns the same bits (ring protocol: no read of a chunk that has not landed)
    again = ops.mlp_rows(ops.PackedMLP(lin, ln, "bf16_n16"), attr, tiled=True).to_rows()
    assert torch.equal(ring, again)

"""GPU: cgnn_node_block in its two f32-emulating forms -- three bf16 terms (CGNN_F32X3_N16) and two fp16 terms
(CGNN_F16X2_N16) -- against a float64 evaluation of the reference's node update (graph_network.py:94-96: node MLP on
cat([x, aggregated]) with LayerNorm, :182 residual) and against the exact-f32 kernel.  Both must sit at f32 rounding
level (the model-level 1e-5 gate is tested in test_gpu_parity.py); the fused projection epilogue carries the bf16
table's own rounding."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cosmology_gnn_simulation_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand_node_mlp(gen, d, nh, w_scale=1.0):
    dims = [2 * d] + [d] * nh + [d]
    lin = []
    for i in range(nh + 1):
        bound = w_scale / np.sqrt(dims[i])
        w = (torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * bound
        b = (torch.rand(dims[i + 1], generator=gen) * 2 - 1) * bound
        lin.append((w.to(DEV), b.to(DEV)))
    ln = ((1 + 0.1 * torch.randn(d, generator=gen)).to(DEV), (0.1 * torch.randn(d, generator=gen)).to(DEV))
    return lin, ln


def _f64(lin, ln, x, agg, residual=True):
    h = torch.cat([x, agg], dim=1).double()
    for i, (w, b) in enumerate(lin):
        h = h @ w.double().t() + b.double()
        if i < len(lin) - 1:
            h = torch.relu(h)
    h = F.layer_norm(h, (h.shape[1],), ln[0].double(), ln[1].double(), 1e-5)
    return h + x.double() if residual else h


def _run(fmt, lin, ln, x, agg, residual=True):
    d = x.shape[1]
    w1, b1 = lin[0]
    wx = ops.PackedLinear(w1, b1, fmt, 0, d)
    wa = ops.PackedLinear(w1, None, fmt, d, d)
    mlp = ops.PackedMLP([(w1[:, :d].contiguous(), None)] + lin[1:], ln, fmt)
    out = ops.node_block(mlp, wx, wa, x, agg, None, residual)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("fmt", ["fp32x3_n16", "fp16x2_n16"])
@pytest.mark.parametrize("n,d,nh", [(1000, 128, 2), (17, 128, 2), (4099, 64, 1), (300, 32, 3), (70000, 128, 2),
                                    (500, 128, 4), (129, 128, 1), (128, 128, 3)])      # 4 hidden layers: the two-slot kernel
def test_node_block_f32_emulations_sit_at_f32_rounding_level(fmt, n, d, nh):
    gen = torch.Generator().manual_seed(n + d)
    lin, ln = _rand_node_mlp(gen, d, nh)
    x = torch.randn(n, d, generator=gen).to(DEV) * 2
    agg = torch.randn(n, d, generator=gen).to(DEV) * 8            # a sum of 16 latents
    want = _f64(lin, ln, x, agg)
    got = _run(fmt, lin, ln, x, agg).double()
    exact = _run("fp32", lin, ln, x, agg).double()
    scale = float(want.abs().max())
    err, err_exact = float((got - want).abs().max()) / scale, float((exact - want).abs().max()) / scale
    assert err <= 2e-6, (err, err_exact)
    assert err <= 4 * err_exact + 2e-7, (err, err_exact)           # no worse than the true-f32 kernel's own rounding noise
    assert float((got - want).norm() / want.norm()) <= 3e-7


@pytest.mark.parametrize("fmt", ["fp32x3_n16", "fp16x2_n16"])
def test_node_block_emulations_hold_over_value_scales(fmt):
    """Weights of 1e-3 .. 10 and activations of 1e-4 .. 1e3 in one matrix: the two-term fp16 split scales its residual
    by 2^11, so small values keep their bits (unscaled they would fall into fp16's subnormals)."""
    n, d, nh = 2048, 128, 2
    gen = torch.Generator().manual_seed(3)
    lin, ln = _rand_node_mlp(gen, d, nh)
    col = torch.logspace(-3, 1, d).to(DEV)
    lin[0] = (lin[0][0] * torch.cat([col, col])[None, :], lin[0][1])       # per-input-column weight scale
    row = torch.logspace(-4, 3, n)[:, None].to(DEV)
    x = torch.randn(n, d, generator=gen).to(DEV) * row
    agg = torch.randn(n, d, generator=gen).to(DEV) * row
    want = _f64(lin, ln, x, agg, residual=False)
    got = _run(fmt, lin, ln, x, agg, residual=False).double()
    exact = _run("fp32", lin, ln, x, agg, residual=False).double()
    err = (got - want).abs().amax(dim=1) / want.abs().amax(dim=1)          # per row: every scale counts
    err_exact = (exact - want).abs().amax(dim=1) / want.abs().amax(dim=1)
    assert float(err.max()) <= 5e-6, (float(err.max()), float(err_exact.max()))
    assert float(err.max()) <= 4 * float(err_exact.max()) + 2e-7


def test_fp16x2_overflow_is_loud():
    """|activation| >= 65520 does not fit fp16: the row comes back non-finite, not as a wrong finite number; the other
    rows are untouched."""
    n, d = 64, 128
    gen = torch.Generator().manual_seed(4)
    lin, ln = _rand_node_mlp(gen, d, 2)
    x = torch.randn(n, d, generator=gen).to(DEV)
    agg = torch.randn(n, d, generator=gen).to(DEV)
    x[5, 7] = 7.0e4
    got = _run("fp16x2_n16", lin, ln, x, agg)
    assert not torch.isfinite(got[5]).any()
    keep = torch.ones(n, dtype=torch.bool, device=DEV)
    keep[5] = False
    want = _f64(lin, ln, x, agg)
    assert torch.isfinite(got[keep]).all()
    assert float((got[keep].double() - want[keep]).abs().max()) <= 2e-6 * float(want[keep].abs().max())
    ok = _run("fp32x3_n16", lin, ln, x, agg)                                 # the bf16 form has the f32 range
    assert torch.isfinite(ok).all()


@pytest.mark.parametrize("fmt", ["fp32x3_n16", "fp16x2_n16"])
@pytest.mark.parametrize("p_format", [_lib.P_BF16_S32, _lib.P_BF16_S16, _lib.P_F16_S32])
def test_node_block_fused_projection_epilogue(fmt, p_format):
    """next round's Ps / Pd written by the node kernel == cgnn_project_nodes on its output (same bf16 MFMA products)."""
    n, d = 3000, 128
    gen = torch.Generator().manual_seed(5)
    lin, ln = _rand_node_mlp(gen, d, 2)
    x = torch.randn(n, d, generator=gen).to(DEV)
    agg = torch.randn(n, d, generator=gen).to(DEV) * 4
    w1e = ((torch.rand(d, 3 * d, generator=gen) * 2 - 1) / np.sqrt(3 * d)).to(DEV)
    b1e = ((torch.rand(d, generator=gen) * 2 - 1) / np.sqrt(3 * d)).to(DEV)
    ws16, wd16 = ops.PackedLinear(w1e, None, "bf16_n16", 0, d), ops.PackedLinear(w1e, b1e, "bf16_n16", d, d)
    w1, b1 = lin[0]
    wx, wa = ops.PackedLinear(w1, b1, fmt, 0, d), ops.PackedLinear(w1, None, fmt, d, d)
    mlp = ops.PackedMLP([(w1[:, :d].contiguous(), None)] + lin[1:], ln, fmt)
    ps = torch.empty(n, d, dtype=ops.p_format_dtype(p_format), device=DEV)
    pd = torch.empty_like(ps)
    out = ops.node_block(mlp, wx, wa, x, agg, None, True, (ws16, wd16, ps, pd, p_format))
    plain = ops.node_block(mlp, wx, wa, x, agg, None, True)
    assert torch.equal(out, plain)
    ws, wd = ops.PackedLinear(w1e, None, "bf16", 0, d), ops.PackedLinear(w1e, b1e, "bf16", d, d)
    ps2, pd2 = ops.project_nodes(ws, wd, out, None, None, p_format)
    torch.cuda.synchronize()
    for a, b in ((ps, ps2), (pd, pd2)):
        diff = (a.float() - b.float()).abs()
        assert float(diff.max()) <= 2.0 ** -7 * float(b.float().abs().max())      # at most a bf16 (fp16) rounding flip
        assert float((diff > 0).float().mean()) < 0.02
    if p_format == _lib.P_F16_S32:
        # the fp16 tables hold the SAME projections as the bf16 ones, rounded to 11 bits instead of 8 and laid out with the
        # row's halves interleaved in 64-byte segments (include/cgnn.h): oracle.bf16_stream.s32_position is that order
        from oracle.bf16_stream import s32_table_to_logical
        pb, db = ops.project_nodes(ws, wd, out, None, None, _lib.P_BF16_S32)
        for f16, b16 in ((ps, pb), (pd, db)):
            lf, lb = s32_table_to_logical(f16), s32_table_to_logical(b16)
            assert float((lf - lb).abs().max()) <= 2.0 ** -8 * float(lb.abs().max())      # half a bf16 unit in the last place
            assert float((lf - lb).abs().mean()) > 0      # (and not the same numbers: fp16 keeps three more bits)


@pytest.mark.parametrize("n,d,h,nh", [(1000, 256, 256, 2), (333, 64, 128, 2), (70, 256, 128, 1), (2000, 128, 128, 2),
                                      (5000, 256, 256, 2), (2100, 256, 256, 3), (4133, 256, 256, 1)])
def test_node_block_fp16x2_32_row_packing(n, d, h, nh):
    """CGNN_F16X2 (the two-fp16-term arithmetic in the 32-row packing, any supported latent / hidden pair): f32 rounding
    level against float64, like the three-bf16-term form it replaces at half the matrix work."""
    gen = torch.Generator().manual_seed(n + d + h)
    dims = [2 * d] + [h] * nh + [d]
    lin = []
    for i in range(nh + 1):
        bound = 1.0 / np.sqrt(dims[i])
        lin.append((((torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * bound).to(DEV),
                    ((torch.rand(dims[i + 1], generator=gen) * 2 - 1) * bound).to(DEV)))
    ln = ((1 + 0.1 * torch.randn(d, generator=gen)).to(DEV), (0.1 * torch.randn(d, generator=gen)).to(DEV))
    x = torch.randn(n, d, generator=gen).to(DEV) * 2
    agg = torch.randn(n, d, generator=gen).to(DEV) * 8
    want = _f64(lin, ln, x, agg)
    got = _run("fp16x2", lin, ln, x, agg).double()
    exact = _run("fp32", lin, ln, x, agg).double()
    three = _run("fp32x3", lin, ln, x, agg).double()
    scale = float(want.abs().max())
    err, err_exact, err3 = (float((t - want).abs().max()) / scale for t in (got, exact, three))
    assert err <= 2e-6, (err, err_exact, err3)
    assert err <= 4 * max(err_exact, err3) + 2e-7, (err, err_exact, err3)


@pytest.mark.parametrize("d,h", [(128, 128), (256, 256), (64, 128)])
def test_project_nodes_fp16x2_writes_f32_tables(d, h):
    """cgnn_project_nodes with CGNN_F16X2 weights: CGNN_P_F32 tables at f32 rounding level (the exact-f32 kernel's
    tables are the comparison; the edge kernels that gather f32 rows take either)."""
    n = 1500
    gen = torch.Generator().manual_seed(d + h)
    w = ((torch.rand(h, 3 * d, generator=gen) * 2 - 1) / np.sqrt(3 * d)).to(DEV)
    b = ((torch.rand(h, generator=gen) * 2 - 1) / np.sqrt(3 * d)).to(DEV)
    x = (torch.randn(n, d, generator=gen) * 3).to(DEV)
    tabs = {}
    for prec in ("fp32", "fp16x2"):
        ws, wd = ops.PackedLinear(w, None, prec, 0, d), ops.PackedLinear(w, b, prec, d, d)
        tabs[prec] = ops.project_nodes(ws, wd, x)
    torch.cuda.synchronize()
    want_s = x.double() @ w[:, :d].double().t()
    want_d = x.double() @ w[:, d:2 * d].double().t() + b.double()
    for (got, want) in ((tabs["fp16x2"][0], want_s), (tabs["fp16x2"][1], want_d)):
        assert got.dtype == torch.float32
        assert float((got.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    with pytest.raises(_lib.CgnnError):
        ops.project_nodes(ops.PackedLinear(w, None, "fp16x2", 0, d), ops.PackedLinear(w, b, "fp16x2", d, d), x,
                          p_format=_lib.P_BF16_S32)


@pytest.mark.parametrize("nh", [1, 3, 4])
def test_mlp_rows_fp16x2_depths_and_loud_overflow(nh):
    """The 32-row two-fp16-term packing through cgnn_mlp_rows at other depths, and its range contract: an input of
    7e4 (beyond fp16) turns its row non-finite instead of producing a wrong finite number; other rows are untouched."""
    n, d = 500, 128
    gen = torch.Generator().manual_seed(nh)
    dims = [d] + [d] * nh + [3]
    lin = []
    for i in range(nh + 1):
        bound = 1.0 / np.sqrt(dims[i])
        lin.append((((torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * bound).to(DEV),
                    ((torch.rand(dims[i + 1], generator=gen) * 2 - 1) * bound).to(DEV)))
    x = torch.randn(n, d, generator=gen).to(DEV) * 3
    h = x.double()
    for i, (w, b) in enumerate(lin):
        h = h @ w.double().t() + b.double()
        if i < nh:
            h = torch.relu(h)
    got = ops.mlp_rows(ops.PackedMLP(lin, None, "fp16x2"), x)
    assert float((got.double() - h).abs().max()) <= 2e-6 * float(h.abs().max())
    x[7, 5] = 7.0e4
    bad = ops.mlp_rows(ops.PackedMLP(lin, None, "fp16x2"), x)
    assert not torch.isfinite(bad[7]).any()
    keep = torch.ones(n, dtype=torch.bool, device=DEV)
    keep[7] = False
    assert torch.equal(bad[keep], got[keep])
    ok = ops.mlp_rows(ops.PackedMLP(lin, None, "fp32x3"), x)          # three bf16 terms have the f32 range
    assert torch.isfinite(ok).all()


@pytest.mark.parametrize("fin,d,out,ln", [(17, 128, 128, True), (128, 128, 3, False), (4, 64, 64, True), (64, 64, 1, False)])
def test_mlp_rows_fp16x2_with_weights_resident_in_lds(fin, d, out, ln):
    """From 4096 rows on cgnn_mlp_rows keeps two-fp16-term weights of up to 128-wide layers in LDS (encoder and decoder
    shapes): same results as below that size (weights read through L2) bit for bit, f32 rounding level against float64."""
    n, nh = 6000, 2
    gen = torch.Generator().manual_seed(fin + out)
    dims = [fin] + [d] * nh + [out]
    lin = []
    for i in range(nh + 1):
        bound = 1.0 / np.sqrt(dims[i])
        lin.append((((torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * bound).to(DEV),
                    ((torch.rand(dims[i + 1], generator=gen) * 2 - 1) * bound).to(DEV)))
    lnp = ((1 + 0.1 * torch.randn(out, generator=gen)).to(DEV), (0.1 * torch.randn(out, generator=gen)).to(DEV)) if ln else None
    x = torch.randn(n, fin, generator=gen).to(DEV) * 2
    h = x.double()
    for i, (w, b) in enumerate(lin):
        h = h @ w.double().t() + b.double()
        if i < nh:
            h = torch.relu(h)
    if ln:
        h = F.layer_norm(h, (out,), lnp[0].double(), lnp[1].double(), 1e-5)
    mlp = ops.PackedMLP(lin, lnp, "fp16x2")
    got = ops.mlp_rows(mlp, x)
    assert float((got.double() - h).abs().max()) <= 2e-6 * float(h.abs().max())
    small = torch.cat([ops.mlp_rows(mlp, x[i:i + 2000].contiguous()) for i in range(0, n, 2000)])     # < 4096 rows: weights from L2
    assert torch.equal(got, small)


@pytest.mark.parametrize("prec,d,fmt", [("bf16", 128, None), ("bf16", 64, None), ("fp16x2", 128, None), ("fp16x2", 32, None),
                                         ("bf16", 128, _lib.P_BF16_S16)])
def test_project_nodes_with_weights_resident_in_lds(prec, d, fmt):
    """From 4096 rows on cgnn_project_nodes copies both matrices into LDS once per workgroup (bf16 / two fp16 terms up to
    128 x 128): the tables equal those of the global-weight path (calls below that size) bit for bit."""
    n = 9000
    gen = torch.Generator().manual_seed(d)
    w = ((torch.rand(d, 3 * d, generator=gen) * 2 - 1) / np.sqrt(3 * d)).to(DEV)
    b = ((torch.rand(d, generator=gen) * 2 - 1) / np.sqrt(3 * d)).to(DEV)
    x = (torch.randn(n, d, generator=gen) * 3).to(DEV)
    ws, wd = ops.PackedLinear(w, None, prec, 0, d), ops.PackedLinear(w, b, prec, d, d)
    ps, pd = ops.project_nodes(ws, wd, x, p_format=fmt)
    parts = [ops.project_nodes(ws, wd, x[i:i + 3000].contiguous(), p_format=fmt) for i in range(0, n, 3000)]
    assert torch.equal(ps, torch.cat([p[0] for p in parts])) and torch.equal(pd, torch.cat([p[1] for p in parts]))
    only_d = ops.project_nodes(None, wd, x, p_format=fmt)[1]        # one table alone (ghost rows take this form)
    assert torch.equal(only_d, pd)

"""GPU: cgnn_edge_stream_run (all edge rounds in one launch, 32-edge MFMA tiles) against a torch emulation of its
arithmetic -- bf16 operands, f32 accumulation, f32 LayerNorm and residual (reference graph_network.py:57,:89-90,:182 with
the first Linear split into Ps[src] + Pd[dst] + We e).  The emulation rounds to bf16 exactly where the kernel does, so
the two agree to accumulation-order noise plus the odd bf16 rounding flip: far below what a wrong bias, LayerNorm
vector, fragment or tile would cost."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cosmology_gnn_simulation_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(t):
    return t.bfloat16().float()


def _dot(a, w):      # bf16 operands, wide accumulation
    return (bf(a).double() @ bf(w).double().t()).float()


def _s32_table(p, dtype=torch.bfloat16):
    """logical [R, N, H] values -> table in CGNN_P_BF16_S32 order (include/cgnn.h): feature f = 32t+8g+4h+c sits at
    h*(H/2) + (4t+g)*4 + c; ``dtype`` float16 = CGNN_P_F16_S32 (halves interleaved in 64-byte segments)."""
    from oracle.bf16_stream import s32_position
    pos = s32_position(p.shape[-1], dtype, p.device)
    out = torch.empty_like(p)
    out[..., pos] = p
    return out.to(dtype).contiguous()


def _rand_mlp(gen, fin, d, nh):
    dims = [fin] + [d] * nh + [d]
    lin = []
    for i in range(nh + 1):
        bound = 1.0 / np.sqrt(dims[i])
        w = (torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * bound
        b = (torch.rand(dims[i + 1], generator=gen) * 2 - 1) * bound
        lin.append((w.to(DEV), b.to(DEV)))
    ln = ((1 + 0.1 * torch.randn(d, generator=gen)).to(DEV), (0.1 * torch.randn(d, generator=gen)).to(DEV))
    return lin, ln


def _emulate_mlp(lin, ln, first, ratios=None, no_mean=False):
    """first = pre-activation of layer 0 (bias included).  ``ratios``: a list that receives |mean| / std of every
    LayerNorm input row (then LayerNorm itself runs centred, in float64: the yardstick for the kernels' one-pass variance).
    ``no_mean``: y * rsqrt(E[y^2] + eps), the CGNN_STREAM_FOLDED kernel's LayerNorm of a centred output Linear."""
    h = bf(torch.relu(first))
    for w, b in lin[1:-1]:
        h = bf(torch.relu(_dot(h, w) + b))
    w, b = lin[-1]
    out = _dot(h, w) + b
    if no_mean:
        return out * torch.rsqrt((out * out).mean(1, keepdim=True) + 1e-5) * ln[0] + ln[1]
    if ratios is not None:
        ratios.append(out.mean(1).abs() / out.std(1, unbiased=False))
        return F.layer_norm(out.double(), (out.shape[1],), ln[0].double(), ln[1].double(), 1e-5).float()
    return F.layer_norm(out, (out.shape[1],), ln[0], ln[1], 1e-5)


def _problem(seed, n, k, d, nh, rounds, with_encoder, ragged=0):
    gen = torch.Generator().manual_seed(seed)
    ne = n * k - ragged
    src = torch.randint(0, n, (ne,), generator=gen, dtype=torch.int32).to(DEV)
    dst = torch.randint(0, n, (ne,), generator=gen, dtype=torch.int32).to(DEV)
    ps = bf(torch.randn(rounds, n, d, generator=gen)).to(DEV)
    pd = bf(torch.randn(rounds, n, d, generator=gen)).to(DEV)
    mlps = [_rand_mlp(gen, d, d, nh) for _ in range(rounds)]
    enc = _rand_mlp(gen, 4, d, nh) if with_encoder else None
    attr = (torch.randn(ne, 4, generator=gen) * 0.3).to(DEV)
    e0 = torch.randn(ne, d, generator=gen).to(DEV)
    return src, dst, ps, pd, mlps, enc, attr, e0


def _emulate(src, dst, ps, pd, mlps, enc, attr, e0, ratios=None, no_mean=False):
    s, t = src.long(), dst.long()
    if enc is not None:
        lin, ln = enc
        e = _emulate_mlp(lin, ln, _dot(attr, lin[0][0]) + lin[0][1], ratios, no_mean)
    else:
        e = e0.clone()
    for r, (lin, ln) in enumerate(mlps):
        first = (ps[r][s] + pd[r][t]) + _dot(e, lin[0][0])           # the round's layer-0 bias lives in Pd
        e = e + _emulate_mlp(lin, ln, first, ratios, no_mean)
    return e


# (kernel, lag[, "f16"[, "fold"]]): see ops.edge_stream_run; "f16" = float16 (CGNN_P_F16_S32) tables; "fold" = the problem
# with folded LayerNorms and the CGNN_STREAM_FOLDED flag (include/cgnn.h): what the model feeds tile32w
KERNELS = [("tile32", 0), ("tile32w", 0), ("tile32w", 1), ("tile32w", 0, "f16"), ("tile32w", 0, "f16", "fold")]


def _kid(kk):
    return f"{kk[0]}-lag{kk[1]}" + ("-f16" if len(kk) > 2 else "") + ("-fold" if len(kk) > 3 else "")


def _centre(lin):
    w, b = lin[-1]
    w, b = w.double(), b.double()
    return lin[:-1] + [((w - w.mean(0, keepdim=True)).float(), (b - b.mean()).float())]


def _fold(prob):
    """The problem with its LayerNorms folded (include/cgnn.h, CGNN_STREAM_FOLDED; restated here on the test's own tensors):
    output Linears centred, B_r = beta_0 + .. + beta_{r-1} carried into round r's Pd values (We_r B_r: the bias the caller's
    projection would add), no shift in any round but the last, which takes B_L.  The same e_L in exact arithmetic."""
    src, dst, ps, pd, mlps, enc, attr, e0 = prob
    d = ps.shape[-1]
    B = torch.zeros(d, dtype=torch.float64, device=ps.device)
    pd2, out = pd.clone(), []
    for r, (lin, ln) in enumerate(mlps):
        pd2[r] = pd[r] + (lin[0][0].double() @ B).float()
        B = B + ln[1].double()
        out.append((_centre(lin), (ln[0], B.float() if r + 1 == len(mlps) else torch.zeros_like(ln[1]))))
    return (src, dst, ps, pd2, out, None if enc is None else (_centre(enc[0]), enc[1]), attr, e0)


def _as_fed(prob, kernel):
    """What _run hands the kernel, and what the emulation therefore starts from: the folded problem for a "fold" kernel."""
    return _fold(prob) if len(kernel) > 3 else prob


def _p_as_the_kernel_sees_it(prob, kernel):
    """The emulation's Ps / Pd values: the problem's (bf16-representable) numbers, through float16 when the kernel gets
    float16 tables (identical but for values below fp16's normal range)."""
    if len(kernel) < 3:
        return prob
    src, dst, ps, pd, mlps, enc, attr, e0 = prob
    return (src, dst, ps.half().float(), pd.half().float(), mlps, enc, attr, e0)


def _kernel_applies(kernel, d, nh, k, ragged):
    """The two-waves-per-SIMD kernel takes the receiver-sorted fixed-k edge lists data_utils.preprocess emits."""
    return kernel == "tile32" or (ragged == 0 and ops.stream_w8_supported(d, nh, k))


def _fixed_k(prob, n, k):
    """The same problem on a receiver-sorted edge list of fixed in-degree k (dst[e] == e // k)."""
    src, dst, ps, pd, mlps, enc, attr, e0 = prob
    return (src, torch.arange(n, dtype=torch.int32, device=DEV).repeat_interleave(k), ps, pd, mlps, enc, attr, e0)


def _run(src, dst, ps, pd, mlps, enc, attr, e0, kernel=("tile32", 0), fixed_k=0):
    if len(kernel) > 3:
        src, dst, ps, pd, mlps, enc, attr, e0 = _fold((src, dst, ps, pd, mlps, enc, attr, e0))
    packed = [ops.PackedMLP([(lin[0][0], None)] + lin[1:], ln, "bf16") for lin, ln in mlps]
    penc = ops.PackedMLP(enc[0], enc[1], "bf16") if enc is not None else None
    image = ops.StreamImage(packed, penc, kernel=kernel[0], folded=len(kernel) > 3)
    e_in = None if enc is not None else ops.TiledRows.from_rows(e0)
    pdt = torch.float16 if len(kernel) > 2 else torch.bfloat16
    out = ops.edge_stream_run(image, _s32_table(ps, pdt), _s32_table(pd, pdt), src, dst, e_in, None,
                              attr if enc is not None else None, kernel=kernel[0], lag=kernel[1], fixed_k=fixed_k)
    torch.cuda.synchronize()
    return out.to_rows()


CASES = [
    # n, k, latent, nh, rounds, encoder, ragged (edges dropped from n*k: partial last tile)
    (7, 5, 128, 2, 3, False, 0),            # 35 edges: one pair, second tile missing
    (8, 8, 128, 2, 2, True, 0),             # exactly one pair
    (100, 16, 128, 2, 10, True, 0),
    (100, 16, 128, 2, 10, False, 3),
    (1000, 16, 128, 1, 4, True, 0),         # one hidden layer
    (600, 8, 128, 3, 2, False, 5),          # three hidden layers
    (3000, 16, 64, 2, 5, True, 1),
    (700, 8, 64, 1, 12, False, 0),
    (900, 16, 32, 2, 3, True, 7),
    (333, 8, 32, 3, 6, False, 0),
    (1000, 32, 128, 2, 2, False, 0),        # one receiver per tile
    (150, 64, 128, 2, 2, True, 0),          # two tiles per receiver
    (14000, 16, 128, 2, 2, True, 0),        # 224,000 edges: several pairs per wave, uneven iteration counts
    (9000, 16, 128, 2, 3, False, 0),
    (9000, 16, 128, 2, 3, False, 11),
    (20000, 8, 64, 2, 2, True, 9),
]


@pytest.mark.parametrize("kernel", KERNELS, ids=_kid)
@pytest.mark.parametrize("n,k,d,nh,rounds,with_enc,ragged", CASES)
def test_edge_stream_run_matches_bf16_emulation(n, k, d, nh, rounds, with_enc, ragged, kernel):
    if not _kernel_applies(kernel[0], d, nh, k, ragged):
        pytest.skip("the two-waves-per-SIMD kernel is built for latent 128 and fixed in-degrees 8, 16, 32, ...")
    prob = _problem(1000 + n + d + rounds, n, k, d, nh, rounds, with_enc, ragged)
    fixed_k = 0
    if kernel[0] == "tile32w":
        prob, fixed_k = _fixed_k(prob, n, k), k
    got = _run(*prob, kernel=kernel, fixed_k=fixed_k)
    want = _emulate(*_p_as_the_kernel_sees_it(_as_fed(prob, kernel), kernel), no_mean=len(kernel) > 3)
    assert got.shape == want.shape
    scale = float(want.abs().max())
    err = (got - want).abs()
    assert float(err.max()) <= 1e-2 * scale, (float(err.max()), scale, int(err.argmax()) // d)
    assert float((got - want).norm() / want.norm()) <= 1e-3
    # every tile was written: no row left at its initial value / garbage
    assert torch.isfinite(got).all()
    if len(kernel) > 3:      # the folded problem is the same model: bf16-rounding distance from the plain problem's emulation
        plain = _emulate(*_p_as_the_kernel_sees_it(prob, kernel))
        assert float((want - plain).norm() / plain.norm()) <= 1e-2


@pytest.mark.parametrize("kernel", KERNELS, ids=_kid)
@pytest.mark.parametrize("offset,min_ratio,row_gate,l2_gate", [(10.0, 25.0, 1e-2, 5e-3), (40.0, 100.0, 6e-2, 2e-2)])
def test_edge_stream_layernorm_rows_with_a_large_mean(kernel, offset, min_ratio, row_gate, l2_gate):
    """The stream kernels take LayerNorm's variance in one pass, E[x^2] - mean^2 in f32 (edge_stream32w.hip, ln_stats_finish):
    relative error about 1e-7 (1 + mean^2 / var).  Here every output layer (and the encoder's) carries a constant bias far
    above the spread of W h, so that each LayerNorm input row has |mean| of about 36 standard deviations (offset 10) --
    where the kernels must hold the same 1e-2 x scale row gate as everywhere else -- and about 145 (offset 40), beyond the
    ~100 sigma the kernel header promises: there the error must stay bounded (it grows like (mean / sigma)^2: a few 1e-2 of
    the scale), never garbage.  The yardstick is the emulation with a centred float64 LayerNorm.  The relative-L2 gates are
    wider than the 1e-3 of the other cases: a 1e-4 perturbation of one round's LayerNorm flips bf16 roundings of the
    next rounds' operands, and the flips, not the perturbation, set the L2 distance (a float32 torch model of the
    one-pass form against the float64 one: 2.7e-3 at 36 sigma, 7.8e-3 at 145)."""
    n, k, d, nh, rounds = 3000, 16, 128, 2, 4
    src, dst, ps, pd, mlps, enc, attr, e0 = _problem(4242, n, k, d, nh, rounds, True)
    shift = lambda mlp: (mlp[0][:-1] + [(mlp[0][-1][0], mlp[0][-1][1] + offset)], mlp[1])      # noqa: E731
    prob = (src, dst, ps, pd, [shift(m) for m in mlps], shift(enc), attr, e0)
    fixed_k = 0
    if kernel[0] == "tile32w":
        prob, fixed_k = _fixed_k(prob, n, k), k
    ratios = []
    # (a "fold" kernel never sees these means: centring the output Linear removes the constant with them.  Its yardstick
    # is the same plain emulation; what it differs by is the bf16 rounding of e_r - B_r instead of e_r)
    want = _emulate(*_p_as_the_kernel_sees_it(prob, kernel), ratios=ratios)
    med = float(torch.cat(ratios).median())
    assert med >= min_ratio, med
    got = _run(*prob, kernel=kernel, fixed_k=fixed_k)
    scale = float(want.abs().max())
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= row_gate * scale, (float((got - want).abs().max()), scale, med)
    assert float((got - want).norm() / want.norm()) <= l2_gate, med


@pytest.mark.parametrize("kernel", KERNELS, ids=_kid)
def test_edge_stream_run_is_deterministic_and_in_place(kernel):
    prob = _fixed_k(_problem(5, 5000, 16, 128, 2, 4, False), 5000, 16)
    fk = 16 if kernel[0] == "tile32w" else 0
    a = _run(*prob, kernel=kernel, fixed_k=fk)
    b = _run(*prob, kernel=kernel, fixed_k=fk)
    assert torch.equal(a, b)
    src, dst, ps, pd, mlps, enc, attr, e0 = prob
    src, dst, ps, pd, mlps, enc, attr, e0 = _as_fed(prob, kernel)
    packed = [ops.PackedMLP([(lin[0][0], None)] + lin[1:], ln, "bf16") for lin, ln in mlps]
    image = ops.StreamImage(packed, None, kernel=kernel[0], folded=len(kernel) > 3)
    e = ops.TiledRows.from_rows(e0)
    pdt = torch.float16 if len(kernel) > 2 else torch.bfloat16
    ops.edge_stream_run(image, _s32_table(ps, pdt), _s32_table(pd, pdt), src, dst, e, e, kernel=kernel[0], lag=kernel[1], fixed_k=fk)   # e_out aliases e_in
    assert torch.equal(e.to_rows(), a)


@pytest.mark.parametrize("k", [8, 16, 32, 64])
def test_two_waves_per_simd_kernel_on_every_supported_in_degree(k):
    """k = 8: four receivers per 32-edge tile ... k = 64: two tiles per receiver (sender rows through the LDS staging area,
    receiver rows broadcast); against the emulation and against the one-wave-per-SIMD kernel (same bf16 P values, same
    weights: rounding-level agreement)."""
    n = 20000 // k
    for enc in (True, False):
        prob = _fixed_k(_problem(70 + k, n, k, 128, 2, 3, enc), n, k)
        got = _run(*prob, kernel=("tile32w", 1), fixed_k=k)
        want = _emulate(*prob)
        assert float((got - want).abs().max()) <= 1e-2 * float(want.abs().max())
        assert float((got - want).norm() / want.norm()) <= 1e-3
        other = _run(*prob, kernel=("tile32", 0))
        assert float((got - other).norm() / other.norm()) <= 1e-3
        # float16 tables (Ps[src] + Pd[dst] on the vector pipe): the same sums of the same numbers, f32 either way
        f16 = _run(*prob, kernel=("tile32w", 0, "f16"), fixed_k=k)
        assert float((f16 - _emulate(*_p_as_the_kernel_sees_it(prob, ("tile32w", 0, "f16")))).abs().max()) <= 1e-2 * float(want.abs().max())
        assert float((f16 - got).norm() / got.norm()) <= 1e-3
        # ... and with folded LayerNorms (the same model restated: bf16-rounding distance from the plain kernels)
        fk = ("tile32w", 0, "f16", "fold")
        fold = _run(*prob, kernel=fk, fixed_k=k)
        assert float((fold - _emulate(*_p_as_the_kernel_sees_it(_fold(prob), fk), no_mean=True)).abs().max()) <= 1e-2 * float(want.abs().max())
        assert float((fold - got).norm() / got.norm()) <= 1e-2
    with pytest.raises(ops.CgnnError):       # not a supported in-degree: the caller must take cgnn_edge_stream_run
        _run(*_fixed_k(_problem(3, 100, 12, 128, 2, 2, False), 100, 12), kernel=("tile32w", 1), fixed_k=12)
    assert not ops.stream_w8_supported(128, 2, 0) and not ops.stream_w8_supported(64, 2, 16)


def test_two_waves_per_simd_kernel_does_not_depend_on_the_lag():
    """lag only shifts WHEN the second wave of a SIMD runs a layer: same arithmetic per tile, same bits."""
    for seed, enc in ((7, True), (8, False)):
        prob = _fixed_k(_problem(seed, 9000, 16, 128, 2, 3, enc), 9000, 16)
        assert torch.equal(_run(*prob, kernel=("tile32w", 0), fixed_k=16), _run(*prob, kernel=("tile32w", 1), fixed_k=16))


def test_edge_stream_run_one_round_at_a_time_equals_all_rounds():
    """L launches of one round each (the latents cross memory between them) == one launch of L rounds: same code path,
    so the same bits."""
    prob = _problem(6, 3000, 16, 128, 2, 5, False, 2)
    src, dst, ps, pd, mlps, enc, attr, e0 = prob
    all_at_once = _run(*prob)
    tps, tpd = _s32_table(ps), _s32_table(pd)
    e = ops.TiledRows.from_rows(e0)
    for r, (lin, ln) in enumerate(mlps):
        image = ops.StreamImage([ops.PackedMLP([(lin[0][0], None)] + lin[1:], ln, "bf16")], None)
        ops.edge_stream_run(image, tps[r:r + 1].contiguous(), tpd[r:r + 1].contiguous(), src, dst, e, e)
    assert torch.equal(e.to_rows(), all_at_once)


def test_edge_stream_image_rejects_what_it_cannot_hold():
    gen = torch.Generator().manual_seed(0)
    lin, ln = _rand_mlp(gen, 64, 64, 2)
    good = ops.PackedMLP([(lin[0][0], None)] + lin[1:], ln, "bf16")
    ops.StreamImage([good], None)
    with pytest.raises(ops.CgnnError, match="bf16"):
        ops.StreamImage([ops.PackedMLP([(lin[0][0], None)] + lin[1:], ln, "bf16_n16")], None)
    assert not ops.StreamImage.supported(256, 256, 2, 3)
    assert not ops.StreamImage.supported(128, 64, 2, 3)
    assert not ops.StreamImage.supported(128, 128, 2, 3, enc_in=17)
    assert ops.StreamImage.supported(128, 128, 2, 10, enc_in=4)
    lin2, ln2 = _rand_mlp(gen, 64, 64, 1)
    with pytest.raises(ops.CgnnError):       # rounds of different depth
        ops.StreamImage([good, ops.PackedMLP([(lin2[0][0], None)] + lin2[1:], ln2, "bf16")], None)


def test_two_waves_per_simd_kernel_rejects_flags_it_does_not_know():
    """CGNN_STREAM_FOLDED is a promise about the image AND the Pd tables: it is accepted with fp16 tables only (the kernel
    that uses it), unknown bits are refused (include/cgnn.h)."""
    from cosmology_gnn_simulation_amd import _lib
    from cosmology_gnn_simulation_amd.ops import stream_ptr
    n, k, d = 64, 16, 128
    src, dst, ps, pd, mlps, enc, attr, e0 = _fixed_k(_problem(9, n, k, d, 2, 2, False), n, k)
    packed = [ops.PackedMLP([(lin[0][0], None)] + lin[1:], ln, "bf16") for lin, ln in mlps]
    image = ops.StreamImage(packed, None, kernel="tile32w")
    e = ops.TiledRows.from_rows(e0)
    lib = _lib.load()

    def call(tables_dtype, p_format, flags):
        tps, tpd = _s32_table(ps, tables_dtype), _s32_table(pd, tables_dtype)
        rc = lib.cgnn_edge_stream_run_w8(image.buf.data_ptr(), image.buf.numel(), d, 2, 2, 0, tps.data_ptr(), tpd.data_ptr(),
                                         tps.stride(0), src.data_ptr(), dst.data_ptr(), n * k, e.buf.data_ptr(), e.buf.data_ptr(),
                                         None, 0, 0, k, p_format, flags, stream_ptr(src.device))
        torch.cuda.synchronize()
        return rc
    assert call(torch.float16, _lib.P_F16_S32, 0) == 0
    assert call(torch.float16, _lib.P_F16_S32, _lib.STREAM_FOLDED) == 0
    assert call(torch.bfloat16, _lib.P_BF16_S32, _lib.STREAM_FOLDED) != 0
    assert call(torch.float16, _lib.P_F16_S32, 2) != 0
    assert b"flags" in lib.cgnn_last_error()

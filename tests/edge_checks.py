"""Gates for edge-latent matrices that can see ONE wrong 32-edge tile among millions (test infrastructure).

A whole-matrix relative L2 norm is blind to a bad tile: one wrong tile in 10^4 moves it by 1e-2 at most.  The gates here
are per ROW, evaluated on the device in chunks (the matrices are 8-33 GB at the BASELINE sizes):

* ``assert_rows_close(a, b, tol)``: max over rows of ||a_row - b_row|| / ||b_row|| <= tol, for two runs of the same
  arithmetic through different kernels (bf16 paths differ by rounding flips: a few 1e-3 per row; a row of a wrong tile
  differs by O(1));
* ``sample_rows(...)``: the rows a persistent tile loop is most likely to get wrong -- the last tiles of every
  workgroup's range (the kernels cut the tile range into eight XCD shares and stride through each), the final
  (possibly partial) tiles, plus uniformly random ones;
* ``emulate_edge_stream_rows(...)``: the bf16-operand / f32-accumulate arithmetic of the one-launch edge stream
  (reference graph_network.py:57,:89-90,:182 with the first Linear split into Ps[src] + Pd[dst] + We e), recomputed in
  torch for sampled rows from the SAME Ps / Pd tables the kernel consumed;
* ``corrupted_tile(...)``: a context manager that overwrites one tile of a result in place (with its neighbour's values:
  right statistics, wrong edges) and restores it -- every gate is run against it once to prove it would fail.
"""
import contextlib

import torch
import torch.nn.functional as F


def bf(t):
    return t.bfloat16().float()


def dot_bf16(a, w):
    """bf16 operands, wide accumulation (the MFMA's f32 accumulation order is not reproduced; f64 is the midpoint)."""
    return (bf(a).double() @ bf(w).double().t()).float()


def row_rel_max(a: torch.Tensor, b: torch.Tensor, chunk: int = 1 << 20):
    """-> (max over rows of ||a_row - b_row|| / ||b_row||, the row it occurs at); rows of b with zero norm count with
    their absolute difference."""
    assert a.shape == b.shape and a.dim() == 2
    worst, where = 0.0, -1
    for r0 in range(0, a.shape[0], chunk):
        x, y = a[r0:r0 + chunk], b[r0:r0 + chunk]
        num = (x - y).float().norm(dim=1)
        den = y.float().norm(dim=1).clamp_min(1e-30)
        rel = num / den
        rel = torch.where(torch.isfinite(rel), rel, torch.full_like(rel, float("inf")))
        m, i = rel.max(dim=0)
        if float(m) > worst or where < 0:
            worst, where = float(m), r0 + int(i)
    return worst, where


def assert_rows_close(a, b, tol, what="edge latents"):
    worst, where = row_rel_max(a, b)
    assert worst <= tol, f"{what}: row {where} (tile {where // 32}) differs by {worst:.3e} relative (gate {tol:.1e})"


def sample_rows(num_edges: int, n_random: int = 4096, seed: int = 0, device="cuda", tail_pairs: int = 256):
    """Row indices: two rows of every tile among the last ``tail_pairs`` tile PAIRS of each eighth of the pair range
    (covers the last iteration of every wave of a grid of up to 8 * tail_pairs / waves workgroups, for kernels that
    stride through eighths in pairs or in single tiles), all rows of the last two tiles, ``n_random`` random rows."""
    tiles = (num_edges + 31) // 32
    pairs = (tiles + 1) // 2
    picks = []
    for x in range(8):
        end = pairs * (x + 1) // 8
        lo = max(pairs * x // 8, end - tail_pairs)
        t = torch.arange(2 * lo, min(2 * end, tiles), dtype=torch.int64)
        picks += [t * 32 + 3, t * 32 + 29]
    picks.append(torch.arange(max(0, (tiles - 2) * 32), num_edges, dtype=torch.int64))
    gen = torch.Generator().manual_seed(seed)
    picks.append(torch.randint(0, num_edges, (n_random,), generator=gen, dtype=torch.int64))
    rows = torch.unique(torch.cat(picks))
    return rows[rows < num_edges].to(device)


def s32_table_to_logical(table: torch.Tensor) -> torch.Tensor:
    """CGNN_P_BF16_S32 (include/cgnn.h): feature f = 32t + 8g + 4h + c is stored at h*(H/2) + (4t + g)*4 + c.
    -> float32 values in feature order (last dimension)."""
    H = table.shape[-1]
    f = torch.arange(H, device=table.device)
    t, g, h, c = f // 32, (f % 32) // 8, (f % 8) // 4, f % 4
    pos = h * (H // 2) + (4 * t + g) * 4 + c
    return table[..., pos].float()


def _mlp_tail(h0, lins, ln):
    """h0 = pre-activation of layer 0 (bias included); lins = [(w, b)] of layers 1..; ln = (gamma, beta)."""
    h = bf(torch.relu(h0))
    for w, b in lins[:-1]:
        h = bf(torch.relu(dot_bf16(h, w) + b))
    w, b = lins[-1]
    out = dot_bf16(h, w) + b
    return F.layer_norm(out, (out.shape[1],), ln[0], ln[1], 1e-5)


def emulate_edge_stream_rows(sd: dict, rows: torch.Tensor, stream_inputs: dict, latent: int, nh: int, rounds: int,
                             with_encoder: bool = True) -> torch.Tensor:
    """The edge latents after ``rounds`` residual updates for the edge rows ``rows`` (engine numbering), from the
    reference's parameters ``sd`` (state_dict keys of graph_network.py:133-148) and the tables the kernel read."""
    dev = rows.device
    D = latent
    W = lambda k: sd[k].to(dev)      # noqa: E731
    src, dst = stream_inputs["src"][rows].long(), stream_inputs["dst"][rows].long()
    if with_encoder:
        pre = "encoder.edge_model"
        lins = [(W(f"{pre}.0.{2 * i}.weight"), W(f"{pre}.0.{2 * i}.bias")) for i in range(nh + 1)]
        attr = stream_inputs["edge_attr"][rows]
        e = _mlp_tail(dot_bf16(attr, lins[0][0]) + lins[0][1], lins[1:], (W(f"{pre}.1.weight"), W(f"{pre}.1.bias")))
    else:
        e = stream_inputs["e_in"][rows].clone()
    for r in range(rounds):
        pre = f"processor.{r}.edge_model"
        w0 = W(f"{pre}.0.0.weight")
        lins = [(W(f"{pre}.0.{2 * i}.weight"), W(f"{pre}.0.{2 * i}.bias")) for i in range(1, nh + 1)]
        ps = s32_table_to_logical(stream_inputs["ps_all"][r][src])
        pd = s32_table_to_logical(stream_inputs["pd_all"][r][dst])       # carries the layer-0 bias
        first = (ps + pd) + dot_bf16(e, w0[:, 2 * D:3 * D])
        e = e + _mlp_tail(first, lins, (W(f"{pre}.1.weight"), W(f"{pre}.1.bias")))
    return e


def assert_rows_match_emulation(got_rows: torch.Tensor, want_rows: torch.Tensor, rows: torch.Tensor, what="edge latents"):
    """Per sampled row: max-abs <= 1e-2 x the sample's scale (a bf16 rounding flip in a hidden layer moves one value by
    a few 1e-3 of the scale; a wrong bias / LayerNorm vector / fragment / tile moves whole rows by O(scale))."""
    scale = float(want_rows.abs().max())
    err = (got_rows - want_rows).abs().max(dim=1).values
    worst, i = err.max(dim=0)
    assert float(worst) <= 1e-2 * scale, (f"{what}: edge row {int(rows[int(i)])} (tile {int(rows[int(i)]) // 32}) is off by "
                                          f"{float(worst):.3e}, scale {scale:.3e}")
    assert float((got_rows - want_rows).norm() / want_rows.norm()) <= 2e-3, what


@contextlib.contextmanager
def corrupted_tile(rows_matrix: torch.Tensor, tile: int):
    """Inside the block, tile ``tile`` (32 rows) of the row-major matrix holds the NEXT tile's values (or the previous
    one's for the last tile): plausible numbers on the wrong edges, what a wrong tile index or a stale register tile
    produces.  Restored afterwards."""
    n = rows_matrix.shape[0]
    r0, r1 = tile * 32, min(tile * 32 + 32, n)
    other = r1 if r1 + (r1 - r0) <= n else r0 - 32
    assert other >= 0
    saved = rows_matrix[r0:r1].clone()
    rows_matrix[r0:r1] = rows_matrix[other:other + (r1 - r0)].clone()
    try:
        yield
    finally:
        rows_matrix[r0:r1] = saved


def must_fail(fn, *args, **kwargs):
    """The gate ``fn`` has to reject its (corrupted) arguments."""
    try:
        fn(*args, **kwargs)
    except AssertionError:
        return
    raise AssertionError(f"{getattr(fn, '__name__', fn)} did not notice a deliberately corrupted tile")

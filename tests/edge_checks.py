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

# the bf16-operand arithmetic itself is part of the oracle (oracle/bf16_stream.py: reference citations, pinned against
# cpu_ref on the CPU in every run); this file keeps the gates
from oracle.bf16_stream import bf, dot_bf16, emulate_edge_stream_rows, fold_state_dict, s32_table_to_logical  # noqa: F401


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

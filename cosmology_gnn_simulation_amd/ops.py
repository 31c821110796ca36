"""Tensor-level wrappers over the C ABI (one Python function per ``cgnn_*`` op).

Every function validates devices/dtypes, hands raw device pointers and the
current HIP stream to ``libcgnn_hip.so`` and raises :class:`CgnnError` on a
non-zero status.  Outputs are torch tensors so callers keep normal ownership.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import BF16, BF16_N16, F32, CgnnError, Linear, Mlp, check, f32c, i32c, ptr, require_device, stream_ptr


# ---- optional per-op timing with HIP events on the launch stream (bench.py) -------------------------------
_timer = None


class OpTimer:
    """``with OpTimer() as t: ...; t.summary()`` -> {op: (calls, total_ms)}.  Events are recorded on the stream
    the kernels are launched on (torch's current stream), so no extra synchronisation enters the region."""

    def __init__(self):
        self.records = {}

    def __enter__(self):
        global _timer
        _timer = self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = None
        return False

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.records.items()}


class _timed:
    """Launch context of one op: makes the tensors' device the current HIP device for the duration of the call (the
    library launches on the current device and sets kernel attributes there; the stream handed over belongs to this
    device) and, under an :class:`OpTimer`, brackets the call with events on the launch stream."""

    def __init__(self, name: str, device):
        self.name, self.device = name, torch.device(device)
        self.guard = None

    def __enter__(self):
        if self.device.type == "cuda" and self.device.index is not None and \
                self.device.index != torch.cuda.current_device():
            self.guard = torch.cuda.device(self.device)
            self.guard.__enter__()
        if _timer is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.device))

    def __exit__(self, *exc):
        if _timer is not None:
            self.b.record(torch.cuda.current_stream(self.device))
            _timer.records.setdefault(self.name, []).append((self.a, self.b))
        if self.guard is not None:
            self.guard.__exit__(*exc)
            self.guard = None
        return False


def _same_device(*tensors):
    """All tensor arguments of one op must live on one device."""
    devs = {t.device for t in tensors if t is not None and torch.is_tensor(t)}
    if len(devs) > 1:
        raise CgnnError(f"tensors of one op live on different devices: {sorted(str(d) for d in devs)}")


def _prec(p) -> int:
    if isinstance(p, int):
        return p
    try:
        return _lib.PRECISIONS[str(p).lower()]
    except KeyError:
        raise ValueError(f"unknown precision {p!r}; use 'fp32' or 'bf16'") from None


class PackedLinear:
    """A Linear layer (or a column slice of one) in MFMA-fragment order."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], precision, col0: int = 0,
                 ncols: Optional[int] = None):
        lib = _lib.load()
        w = f32c(weight.detach(), "weight")
        out_dim, ld = w.shape
        ncols = ld - col0 if ncols is None else ncols
        self.precision = _prec(precision)
        self.in_dim, self.out_dim = int(ncols), int(out_dim)
        nbytes = lib.cgnn_packed_linear_bytes(out_dim, ncols, self.precision)
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        with _timed("pack_linear", w.device):
            check(lib.cgnn_pack_linear(w.data_ptr(), out_dim, ld, col0, ncols, self.precision, self.packed.data_ptr(),
                                       stream_ptr(w.device)), "cgnn_pack_linear")
        self.bias = None if bias is None else f32c(bias.detach(), "bias").clone()

    def struct(self) -> Linear:
        return Linear(self.packed.data_ptr(), ptr(self.bias), self.in_dim, self.out_dim)


class PackedMLP:
    """``build_mlp`` (+ optional LayerNorm) ready for the kernels."""

    def __init__(self, linears: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]],
                 layer_norm: Optional[Tuple[torch.Tensor, torch.Tensor]], precision,
                 first_layer_cols: Optional[Tuple[int, int]] = None):
        nh = len(linears) - 1
        if nh < 1 or nh > _lib.MAX_HIDDEN_LAYERS:
            raise CgnnError(f"mlp_num_hidden_layers={nh} outside [1, {_lib.MAX_HIDDEN_LAYERS}]")
        self.precision = _prec(precision)
        self.layers: List[PackedLinear] = []
        for i, (w, b) in enumerate(linears):
            if i == 0 and first_layer_cols is not None:
                self.layers.append(PackedLinear(w, b, self.precision, first_layer_cols[0], first_layer_cols[1]))
            else:
                self.layers.append(PackedLinear(w, b, self.precision))
        self.gamma = self.beta = None
        if layer_norm is not None:
            self.gamma = f32c(layer_norm[0].detach(), "ln weight").clone()
            self.beta = f32c(layer_norm[1].detach(), "ln bias").clone()
        self.num_hidden_layers = nh
        self.in_dim = self.layers[0].in_dim
        self.hidden = self.layers[0].out_dim
        self.out_dim = self.layers[-1].out_dim
        m = Mlp()
        m.precision = self.precision
        m.num_hidden_layers = nh
        for i, L in enumerate(self.layers):
            m.layer[i] = L.struct()
        m.ln_gamma = ptr(self.gamma)
        m.ln_beta = ptr(self.beta)
        self._struct = m

    def struct(self) -> Mlp:
        return self._struct

    def lds_bytes(self) -> int:
        """Bytes the LDS-resident kernels need for this MLP (packed weights + bias / LayerNorm vectors)."""
        pad16 = lambda n: (n * 4 + 15) // 16 * 16  # noqa: E731
        return sum(L.packed.numel() for L in self.layers) + sum(pad16(L.out_dim) for L in self.layers) + \
            2 * pad16(self.out_dim)


class TiledRows:
    """An ``[n, width]`` float32 matrix in the engine's TILED32 layout (``cgnn_layout`` in include/cgnn.h):
    32-row tiles stored in MFMA-accumulator order so that a wavefront moves a tile with fully coalesced
    accesses.  ``buf`` has ``tiled_rows(n)`` rows; only the kernels interpret its bytes."""

    def __init__(self, n: int, width: int, device, buf: Optional[torch.Tensor] = None):
        if width % 32:
            raise CgnnError(f"TILED32 needs a width that is a multiple of 32 (got {width})")
        self.n, self.width = int(n), int(width)
        rows = ((self.n + 31) // 32) * 32
        self.buf = buf if buf is not None else torch.empty((rows, width), dtype=torch.float32, device=device)
        if self.buf.shape != (rows, width) or not self.buf.is_contiguous():
            raise CgnnError("TiledRows: buffer has the wrong shape")

    @property
    def device(self):
        return self.buf.device

    def empty_like(self) -> "TiledRows":
        return TiledRows(self.n, self.width, self.buf.device)

    def to_rows(self) -> torch.Tensor:
        return relayout(self)

    @staticmethod
    def from_rows(x: torch.Tensor) -> "TiledRows":
        return relayout(x)


def relayout(x):
    """``TiledRows -> row-major tensor`` or ``row-major tensor -> TiledRows``."""
    lib = _lib.load()
    if isinstance(x, TiledRows):
        out = torch.empty((x.n, x.width), dtype=torch.float32, device=x.device)
        if x.n:
            with _timed("relayout", x.device):
                check(lib.cgnn_relayout(x.buf.data_ptr(), _lib.TILED32, out.data_ptr(), _lib.ROWS, x.n, x.width,
                                        stream_ptr(x.device)), "cgnn_relayout")
        return out
    x = f32c(x, "x")
    t = TiledRows(x.shape[0], x.shape[1], x.device)
    if t.n:
        with _timed("relayout", x.device):
            check(lib.cgnn_relayout(x.data_ptr(), _lib.ROWS, t.buf.data_ptr(), _lib.TILED32, t.n, t.width,
                                    stream_ptr(x.device)), "cgnn_relayout")
    return t


def mlp_rows(mlp: PackedMLP, x: torch.Tensor, out=None, tiled: bool = False):
    """Row-wise MLP.  ``tiled=True`` (or ``out`` a :class:`TiledRows`) writes the result in TILED32 layout."""
    x = f32c(x, "x")
    n = x.shape[0]
    if x.dim() != 2 or x.shape[1] != mlp.in_dim:
        raise CgnnError(f"mlp_rows: input is {tuple(x.shape)}, the MLP expects [n, {mlp.in_dim}]")
    if tiled or isinstance(out, TiledRows):
        y = out if out is not None else TiledRows(n, mlp.out_dim, x.device)
        yb, layout = y.buf, _lib.TILED32
    else:
        y = out if out is not None else torch.empty((n, mlp.out_dim), dtype=torch.float32, device=x.device)
        yb, layout = y, _lib.ROWS
    with _timed("mlp_rows", x.device):
        check(_lib.load().cgnn_mlp_rows(C.byref(mlp.struct()), x.data_ptr(), n, x.stride(0), yb.data_ptr(),
                                        yb.stride(0), layout, stream_ptr(x.device)), "cgnn_mlp_rows")
    return y


def p_table_format(edge_mlp_precision) -> int:
    """``cgnn_ptable`` format the edge kernel of this precision gathers from (include/cgnn.h)."""
    return {F32: _lib.P_F32, BF16: _lib.P_BF16_S32, BF16_N16: _lib.P_BF16_S16,
            _lib.F16X2_N16: _lib.P_F32, _lib.F16X2: _lib.P_F32}[_prec(edge_mlp_precision)]


def p_format_dtype(p_format: int) -> torch.dtype:
    """torch element type of a ``cgnn_ptable`` format."""
    return {_lib.P_F32: torch.float32, _lib.P_F16_S32: torch.float16}.get(p_format, torch.bfloat16)


def p_table_dtype(edge_mlp_precision) -> torch.dtype:
    """Element type of the Ps/Pd gather tables (engine-internal layout, see include/cgnn.h)."""
    return torch.float32 if _prec(edge_mlp_precision) in (F32, _lib.F16X2_N16, _lib.F16X2) else torch.bfloat16


def project_nodes(ws: Optional[PackedLinear], wd: Optional[PackedLinear], x: torch.Tensor,
                  ps: Optional[torch.Tensor] = None, pd: Optional[torch.Tensor] = None,
                  p_format: Optional[int] = None):
    """Per-node halves of the edge model's first Linear.  ``p_format`` (``cgnn_ptable``) defaults to the format
    the edge kernel of the weights' own precision expects."""
    x = f32c(x, "x")
    n = x.shape[0]
    ref = ws if ws is not None else wd
    if p_format is None:
        p_format = p_table_format(ref.precision)
    pdt = p_format_dtype(p_format)
    if ws is not None and ps is None:
        ps = torch.empty((n, ws.out_dim), dtype=pdt, device=x.device)
    if wd is not None and pd is None:
        pd = torch.empty((n, wd.out_dim), dtype=pdt, device=x.device)
    for t in (ps, pd):
        if t is not None and (t.dtype != pdt or not t.is_contiguous()):
            raise CgnnError(f"project_nodes: tables must be contiguous {pdt} for this precision")
    s1 = ws.struct() if ws is not None else None
    s2 = wd.struct() if wd is not None else None
    with _timed("project_nodes", x.device):
        check(_lib.load().cgnn_project_nodes(C.byref(s1) if s1 is not None else None,
                                         C.byref(s2) if s2 is not None else None, ref.precision, x.data_ptr(), n,
                                         ptr(ps) if ws is not None else None, ptr(pd) if wd is not None else None,
                                         p_format, stream_ptr(x.device)), "cgnn_project_nodes")
    return ps, pd


def edge_block(mlp: PackedMLP, ps: torch.Tensor, pd: torch.Tensor, src: torch.Tensor, dst: torch.Tensor,
               e_in: TiledRows, e_out: Optional[TiledRows] = None, e_upd: Optional[TiledRows] = None,
               residual: bool = True, agg_out: Optional[torch.Tensor] = None, x_gather: Optional[torch.Tensor] = None,
               seg_k: int = 0) -> TiledRows:
    """Fused edge update on TILED32 edge latents (``e_out`` may be ``e_in`` for the in-place residual).
    ``agg_out`` (N16 kernels, fixed in-degree ``seg_k`` in {8, 16}): also write the receivers' aggregate -- of the
    sender rows ``x_gather[src]`` when given (PyG's default message), else of the edge update itself."""
    if not isinstance(e_in, TiledRows):
        raise CgnnError("edge_block: edge latents must be TiledRows (use ops.relayout / TiledRows.from_rows)")
    src, dst = i32c(src, "src"), i32c(dst, "dst")
    ne, latent = e_in.n, e_in.width
    if e_out is None:
        e_out = e_in.empty_like()
    for t, name in ((ps, "ps"), (pd, "pd")):
        require_device(t, name)
        if not t.is_contiguous() or t.dtype != p_table_dtype(mlp.precision):
            raise CgnnError(f"edge_block: {name} must be a contiguous project_nodes table of the MLP's precision")
    for t in (e_out, e_upd):
        if t is not None and (t.n != ne or t.width != latent):
            raise CgnnError("edge_block: e_out / e_upd do not match the edge latents")
    if src.numel() != ne or dst.numel() != ne:
        raise CgnnError("edge_block: src/dst length does not match the edge latents")
    if agg_out is not None:
        if x_gather is not None:
            x_gather = f32c(x_gather, "x_gather")
            if x_gather.shape[1] != latent:
                raise CgnnError("edge_block: x_gather width must equal the latent size")
        if agg_out.dtype != torch.float32 or not agg_out.is_contiguous() or agg_out.shape[1] != latent:
            raise CgnnError("edge_block: agg_out must be contiguous float32 [receivers, latent]")
    _same_device(ps, pd, src, dst, e_in.buf, e_out.buf, x_gather, agg_out)
    with _timed("edge_block", e_in.device):
        check(_lib.load().cgnn_edge_block(C.byref(mlp.struct()), ps.data_ptr(), pd.data_ptr(), src.data_ptr(),
                                          dst.data_ptr(), ne, e_in.buf.data_ptr(), e_out.buf.data_ptr(),
                                          None if e_upd is None else e_upd.buf.data_ptr(),
                                          1 if residual else 0, latent, ptr(x_gather), ptr(agg_out), seg_k,
                                          stream_ptr(e_in.device)), "cgnn_edge_block")
    return e_out


def edge_stream(mlps: Sequence[PackedMLP], ps_all: torch.Tensor, pd_all: torch.Tensor, src: torch.Tensor,
                dst: torch.Tensor, e_in: Optional[TiledRows], e_out: Optional[TiledRows] = None,
                encoder: Optional[PackedMLP] = None, edge_attr: Optional[torch.Tensor] = None) -> TiledRows:
    """All ``len(mlps)`` residual edge updates in one launch (``cgnn_edge_stream``; reference-faithful message only:
    the caller has already computed every round's ``Ps`` / ``Pd``).  ``ps_all`` / ``pd_all``: ``[rounds, N, H]`` bf16
    tables in the 16-edge kernel's format.  With ``encoder`` (the packed edge encoder) and ``edge_attr`` the initial
    latents are computed in the same launch instead of being read from ``e_in``."""
    src, dst = i32c(src, "src"), i32c(dst, "dst")
    rounds = len(mlps)
    ne = src.numel()
    if encoder is not None:
        if edge_attr is None:
            raise CgnnError("edge_stream: the encoder needs edge_attr")
        edge_attr = f32c(edge_attr, "edge_attr")
        if edge_attr.shape != (ne, encoder.in_dim) or encoder.precision != BF16_N16:
            raise CgnnError("edge_stream: edge_attr must be [E, encoder fan-in] and the encoder packed 'bf16_n16'")
        latent = encoder.out_dim
        if e_out is None:
            e_out = TiledRows(ne, latent, src.device)
    else:
        if not isinstance(e_in, TiledRows):
            raise CgnnError("edge_stream: edge latents must be TiledRows")
        latent = e_in.width
        if e_out is None:
            e_out = e_in.empty_like()
    for t, name in ((ps_all, "ps_all"), (pd_all, "pd_all")):
        require_device(t, name)
        if t.dtype != torch.bfloat16 or not t.is_contiguous() or t.dim() != 3 or t.shape[0] != rounds:
            raise CgnnError(f"edge_stream: {name} must be a contiguous bfloat16 [rounds, N, H] table")
    if any(m.precision != BF16_N16 for m in mlps):
        raise CgnnError("edge_stream: the edge models must be packed 'bf16_n16'")
    if dst.numel() != ne or e_out.n != ne or e_out.width != latent or (e_in is not None and e_in.n != ne):
        raise CgnnError("edge_stream: src/dst/e_out do not match the edge latents")
    arr = (Mlp * rounds)(*[m.struct() for m in mlps])
    enc = encoder.struct() if encoder is not None else None
    with _timed("edge_stream", src.device):
        check(_lib.load().cgnn_edge_stream(arr, rounds, ps_all.data_ptr(), pd_all.data_ptr(),
                                           ps_all.stride(0), src.data_ptr(), dst.data_ptr(), ne,
                                           e_in.buf.data_ptr() if e_in is not None else None, e_out.buf.data_ptr(), latent,
                                           C.byref(enc) if enc is not None else None, ptr(edge_attr),
                                           edge_attr.stride(0) if edge_attr is not None else 0,
                                           stream_ptr(src.device)), "cgnn_edge_stream")
    return e_out


class StreamImage:
    """All rounds' edge models (and optionally the edge encoder in front) as the contiguous chunk image
    ``cgnn_edge_stream_run`` cycles through its LDS ring (``cgnn_edge_stream_image_build``).  The models must be
    packed ``"bf16"`` with ``hidden == latent`` in {32, 64, 128}; ``supported()`` says whether a shape qualifies."""

    @staticmethod
    def supported(latent: int, hidden: int, nh: int, rounds: int, enc_in: Optional[int] = None) -> bool:
        if hidden != latent or rounds < 1 or (enc_in is not None and enc_in > 16):
            return False
        return _lib.load().cgnn_edge_stream_image_bytes(latent, nh, rounds, 1 if enc_in is not None else 0) > 0

    def __init__(self, mlps: Sequence[PackedMLP], encoder: Optional[PackedMLP] = None, kernel: str = "tile32",
                 folded: bool = False):
        """``kernel``: which kernel the image is for -- ``"tile32w"`` (``cgnn_edge_stream_run_w8``) wants every bias one chunk
        early (``cgnn_edge_stream_image_build_w8``); the images are not interchangeable.  ``folded``: the caller's promise
        that ``mlps`` / ``encoder`` were packed from folded LayerNorms (``CGNN_STREAM_FOLDED``, include/cgnn.h;
        ``graph_network.fold_edge_stream`` produces them): ``edge_stream_run`` then passes the flag to the kernels that use it."""
        lib = _lib.load()
        if kernel not in ("tile32", "tile32w"):
            raise CgnnError(f"StreamImage: unknown kernel {kernel!r}")
        self.kernel = kernel
        self.folded = bool(folded)
        self.rounds = len(mlps)
        self.latent = mlps[0].out_dim
        self.nh = mlps[0].num_hidden_layers
        self.enc_in = encoder.in_dim if encoder is not None else 0
        if any(m.precision != BF16 for m in mlps) or (encoder is not None and encoder.precision != BF16):
            raise CgnnError("StreamImage: the edge models must be packed 'bf16'")
        nbytes = lib.cgnn_edge_stream_image_bytes(self.latent, self.nh, self.rounds, 1 if encoder is not None else 0)
        if nbytes == 0:
            raise CgnnError(f"StreamImage: latent {self.latent} is not built (hidden == latent in {{32, 64, 128}})")
        dev = mlps[0].layers[0].packed.device
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        arr = (Mlp * self.rounds)(*[m.struct() for m in mlps])
        enc = encoder.struct() if encoder is not None else None
        build = lib.cgnn_edge_stream_image_build_w8 if kernel == "tile32w" else lib.cgnn_edge_stream_image_build
        check(build(arr, self.rounds, C.byref(enc) if enc is not None else None, self.latent, self.buf.data_ptr(), nbytes,
                    stream_ptr(dev)), "cgnn_edge_stream_image_build")
        self._keep = (list(mlps), encoder)     # the copies are asynchronous: keep the sources alive


def stream_w8_supported(latent: int, nh: int, fixed_k: int) -> bool:
    """Whether ``cgnn_edge_stream_run_w8`` (two waves per SIMD) is built for this shape and in-degree (receiver-sorted
    edge lists of fixed in-degree ``fixed_k``; 0 = any other edge list: not supported)."""
    return bool(_lib.load().cgnn_edge_stream_w8_supported(latent, nh, int(fixed_k)))


def edge_stream_run(image: StreamImage, ps_all: torch.Tensor, pd_all: torch.Tensor, src: torch.Tensor, dst: torch.Tensor,
                    e_in: Optional[TiledRows], e_out: Optional[TiledRows] = None,
                    edge_attr: Optional[torch.Tensor] = None, kernel: str = "tile32", lag: int = 1,
                    fixed_k: int = 0, folded: Optional[bool] = None) -> TiledRows:
    """All residual edge updates of ``image`` in one launch.  ``ps_all`` / ``pd_all``: ``[rounds, N, latent]`` bf16
    tables in ``CGNN_P_BF16_S32`` format -- or, ``"tile32w"`` with ``lag = 0`` only, float16 tables in ``CGNN_P_F16_S32``
    format (include/cgnn.h; the kernel then adds ``Ps[src] + Pd[dst]`` on the vector pipe instead of through selector
    MFMAs: what the model runs).  When the image starts with the encoder the initial latents come from
    ``edge_attr`` and ``e_in`` is ignored.  ``kernel``: ``"tile32"`` = ``cgnn_edge_stream_run`` (one wave per SIMD, two
    tiles per wave), ``"tile32w"`` = ``cgnn_edge_stream_run_w8`` (two waves per SIMD, one tile each; ``lag`` and
    ``fixed_k`` as in include/cgnn.h: the graph's fixed in-degree, ``dst[e] == e // fixed_k``; see
    ``stream_w8_supported``).  ``folded`` (default: what the image says): pass ``CGNN_STREAM_FOLDED`` where the kernel has
    the shorter LayerNorm for it (``"tile32w"`` on float16 tables); every other kernel runs a folded image as a plain one."""
    if kernel not in ("tile32", "tile32w"):
        raise CgnnError(f"edge_stream_run: unknown kernel {kernel!r}")
    if image.kernel != kernel:
        raise CgnnError(f"edge_stream_run: the image was built for {image.kernel!r}, not {kernel!r} (StreamImage(..., kernel=...))")
    src, dst = i32c(src, "src"), i32c(dst, "dst")
    ne, latent = src.numel(), image.latent
    if image.enc_in:
        if edge_attr is None:
            raise CgnnError("edge_stream_run: this image starts with the encoder and needs edge_attr")
        edge_attr = f32c(edge_attr, "edge_attr")
        if edge_attr.shape != (ne, image.enc_in):
            raise CgnnError(f"edge_stream_run: edge_attr must be [E, {image.enc_in}]")
        e_in = None
    elif not isinstance(e_in, TiledRows) or e_in.n != ne or e_in.width != latent:
        raise CgnnError("edge_stream_run: e_in must be the TiledRows edge latents")
    if e_out is None:
        e_out = TiledRows(ne, latent, src.device) if e_in is None else e_in.empty_like()
    pdt = ps_all.dtype
    if pdt == torch.float16 and (kernel != "tile32w" or lag != 0):
        raise CgnnError("edge_stream_run: float16 (CGNN_P_F16_S32) tables are for kernel='tile32w' with lag=0")
    for t, name in ((ps_all, "ps_all"), (pd_all, "pd_all")):
        require_device(t, name)
        if t.dtype != pdt or pdt not in (torch.bfloat16, torch.float16) or not t.is_contiguous() or t.dim() != 3 or \
                t.shape[0] != image.rounds or t.shape[2] != latent:
            raise CgnnError(f"edge_stream_run: {name} must be a contiguous bfloat16 (or, tile32w, float16) "
                            f"[rounds, N, latent] table")
    if dst.numel() != ne or e_out.n != ne or e_out.width != latent:
        raise CgnnError("edge_stream_run: src/dst/e_out do not match the edge latents")
    args = (image.buf.data_ptr(), image.buf.numel(), latent, image.nh, image.rounds, image.enc_in, ps_all.data_ptr(),
            pd_all.data_ptr(), ps_all.stride(0), src.data_ptr(), dst.data_ptr(), ne,
            e_in.buf.data_ptr() if e_in is not None else None, e_out.buf.data_ptr(), ptr(edge_attr),
            edge_attr.stride(0) if edge_attr is not None else 0)
    with _timed("edge_stream", src.device):
        if kernel == "tile32w":
            fold = image.folded if folded is None else bool(folded)
            check(_lib.load().cgnn_edge_stream_run_w8(*args, int(lag), int(fixed_k),
                                                      _lib.P_F16_S32 if pdt == torch.float16 else _lib.P_BF16_S32,
                                                      _lib.STREAM_FOLDED if (fold and pdt == torch.float16) else 0,
                                                      stream_ptr(src.device)), "cgnn_edge_stream_run_w8")
        else:
            check(_lib.load().cgnn_edge_stream_run(*args, stream_ptr(src.device)), "cgnn_edge_stream_run")
    return e_out


class AggregatePlan:
    """Per-graph plan of the fixed-k aggregation (``cgnn_aggregate_plan_build``): per block of 64 receivers the distinct
    sender rows and each edge's position among them, so that a round stages every distinct row once in LDS instead of
    gathering it once per edge.  Valid for the ``gather`` tensor it was built from, in its version at build time.  The plan
    holds only a WEAK reference to that tensor (``AggregatePlan.of`` caches the plan ON the tensor: a strong reference back
    would make a cycle, and a dropped graph's sender list and plan -- 130 MB at 1 M x 16 -- would wait for the cyclic
    collector instead of being freed with the graph); ``aggregate`` is handed the live tensor."""

    MIN_NODES = 8192      # below this the plain gather is launch-bound either way

    def __init__(self, gather: torch.Tensor, num_nodes: int, fixed_k: int):
        gather = i32c(gather, "gather")
        if gather.numel() != num_nodes * fixed_k:
            raise CgnnError("AggregatePlan: gather must hold fixed_k senders per receiver")
        nbytes = _lib.load().cgnn_aggregate_plan_bytes(num_nodes, fixed_k)
        if nbytes == 0:
            raise CgnnError(f"AggregatePlan: fixed_k={fixed_k} cannot be planned")
        self._gather_ref = weakref.ref(gather)
        self.gather_ptr, self.device = gather.data_ptr(), gather.device
        self.num_nodes, self.fixed_k = num_nodes, fixed_k
        self.version = gather._version
        self.blob = torch.empty(nbytes, dtype=torch.uint8, device=gather.device)
        with _timed("aggregate_plan", gather.device):
            check(_lib.load().cgnn_aggregate_plan_build(gather.data_ptr(), num_nodes, fixed_k, self.blob.data_ptr(),
                                                        stream_ptr(gather.device)), "cgnn_aggregate_plan_build")

    @staticmethod
    def supported(num_nodes: int, fixed_k: int, width: int) -> bool:
        return 0 < fixed_k <= 32 and width % 32 == 0 and num_nodes >= AggregatePlan.MIN_NODES

    @staticmethod
    def of(gather: torch.Tensor, num_nodes: int, fixed_k: int, width: int) -> Optional["AggregatePlan"]:
        """The plan cached on ``gather`` (built on first use), or None where the planned kernel does not apply."""
        if not AggregatePlan.supported(num_nodes, fixed_k, width) or gather.dtype != torch.int32 or not gather.is_cuda:
            return None
        plan = getattr(gather, "_cgnn_aggregate_plan", None)
        if plan is None or plan.version != gather._version or plan.num_nodes != num_nodes or plan.fixed_k != fixed_k:
            plan = AggregatePlan(gather, num_nodes, fixed_k)
            gather._cgnn_aggregate_plan = plan
        return plan


def aggregate(table, gather: Optional[torch.Tensor], dst: Optional[torch.Tensor], num_nodes: int,
              fixed_k: int = 0, num_edges: Optional[int] = None, out: Optional[torch.Tensor] = None,
              plan: Optional[AggregatePlan] = None) -> torch.Tensor:
    """``out[i] = sum_{e: dst[e]==i} table[gather[e] if gather is not None else e]``.  ``table`` is a row-major
    tensor, or a :class:`TiledRows` of per-edge messages (``gather`` must then be ``None``).  ``plan`` (for this
    ``gather``): the planned fixed-k kernel, bit-identical results."""
    if plan is not None:
        table = f32c(table, "table")
        if gather is None or gather.data_ptr() != plan.gather_ptr or gather.dtype != torch.int32:
            raise CgnnError("aggregate: the plan was built for another sender list")
        if plan.version != gather._version:
            raise CgnnError("aggregate: the sender list changed after the plan was built")
        if fixed_k != plan.fixed_k or num_nodes != plan.num_nodes or table.shape[1] % 32:
            raise CgnnError("aggregate: the plan does not match this call")
        if out is None:
            out = torch.empty((num_nodes, table.shape[1]), dtype=torch.float32, device=table.device)
        _same_device(table, gather, plan.blob, out)
        with _timed("aggregate", table.device):
            check(_lib.load().cgnn_aggregate_planned_rows(table.data_ptr(), table.shape[0], gather.data_ptr(),
                                                          plan.blob.data_ptr(), num_nodes, fixed_k, table.shape[1],
                                                          out.data_ptr(), stream_ptr(table.device)),
                  "cgnn_aggregate_planned")
        return out
    if isinstance(table, TiledRows):
        tb, layout, width, dev, nrows = table.buf, _lib.TILED32, table.width, table.device, table.n
    else:
        table = f32c(table, "table")
        tb, layout, width, dev, nrows = table, _lib.ROWS, table.shape[1], table.device, table.shape[0]
    if gather is not None:
        gather = i32c(gather, "gather")
    if dst is not None:
        dst = i32c(dst, "dst")
    if num_edges is None:
        num_edges = gather.numel() if gather is not None else (dst.numel() if dst is not None else nrows)
    if out is None:
        out = torch.empty((num_nodes, width), dtype=torch.float32, device=dev)
    _same_device(tb, gather, dst, out)
    with _timed("aggregate", dev):
        check(_lib.load().cgnn_aggregate(tb.data_ptr(), layout, ptr(gather), ptr(dst), num_edges, fixed_k, num_nodes,
                                         width, out.data_ptr(), stream_ptr(dev)), "cgnn_aggregate")
    return out


def node_block(mlp: PackedMLP, w_x: PackedLinear, w_agg: PackedLinear, x: torch.Tensor, agg: torch.Tensor,
               x_out: Optional[torch.Tensor] = None, residual: bool = True, next_projection=None) -> torch.Tensor:
    """Fused node update.  ``next_projection = (ws, wd, ps, pd, p_format)`` additionally fills the next round's
    Ps/Pd tables from ``x_out`` in the same call (fused into the kernel where a specialisation exists)."""
    x, agg = f32c(x, "x"), f32c(agg, "agg")
    n, latent = x.shape
    if x_out is None:
        x_out = torch.empty_like(x)
    sx, sa = w_x.struct(), w_agg.struct()
    if next_projection is not None:
        ws, wd, ps, pd, p_format = next_projection
        s1, s2 = ws.struct(), wd.struct()
        pdt = p_format_dtype(p_format)
        for t in (ps, pd):
            require_device(t, "projection table")
            if t.dtype != pdt or not t.is_contiguous() or t.shape != (n, ws.out_dim):
                raise CgnnError("node_block: projection tables have the wrong dtype/shape")
        proj = (C.byref(s1), C.byref(s2), ws.precision, ps.data_ptr(), pd.data_ptr(), p_format)
    else:
        proj = (None, None, 0, None, None, 0)
    _same_device(x, agg, x_out)
    with _timed("node_block", x.device):
        check(_lib.load().cgnn_node_block(C.byref(mlp.struct()), C.byref(sx), C.byref(sa), x.data_ptr(), agg.data_ptr(),
                                          n, x_out.data_ptr(), 1 if residual else 0, latent, *proj,
                                          stream_ptr(x.device)), "cgnn_node_block")
    return x_out


def knn_periodic(pos: torch.Tensor, box_size: float, k: int, query_ids: Optional[torch.Tensor] = None,
                 want_edge_attr: bool = True, want_order: bool = False):
    """Returns ``(senders int32 [nq*k], edge_attr float32 [nq*k, 4] | None, order int32 [n] | None)``."""
    lib = _lib.load()
    pos = f32c(pos, "pos")
    if pos.dim() != 2 or pos.shape[1] != 3:
        raise CgnnError(f"knn_periodic: pos must be [n, 3], got {tuple(pos.shape)}")
    n = pos.shape[0]
    if query_ids is not None:
        query_ids = i32c(query_ids, "query_ids")
        nq = query_ids.numel()
    else:
        nq = n
    ws_bytes = lib.cgnn_knn_workspace_bytes(n, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pos.device)
    senders = torch.empty(nq * k, dtype=torch.int32, device=pos.device)
    edge_attr = torch.empty((nq * k, 4), dtype=torch.float32, device=pos.device) if want_edge_attr else None
    st = stream_ptr(pos.device)
    with _timed("knn_periodic", pos.device):
        check(lib.cgnn_knn_periodic(pos.data_ptr(), n, float(box_size), k, ptr(query_ids), nq, senders.data_ptr(),
                                ptr(edge_attr), ws.data_ptr(), ws_bytes, st), "cgnn_knn_periodic")
    order = None
    if want_order:
        order = torch.empty(n, dtype=torch.int32, device=pos.device)
        check(lib.cgnn_knn_sorted_order(ws.data_ptr(), n, order.data_ptr(), st), "cgnn_knn_sorted_order")
    return senders, edge_attr, order


def window_features(pos_seq: torch.Tensor, temp_seq: torch.Tensor, metadata: dict, dt: float, box_size: float,
                    pos_noise: Optional[torch.Tensor] = None, temp_noise: Optional[torch.Tensor] = None):
    """``[W, N, 3]`` positions and ``[W, N(, 1)]`` temperatures -> ``(x [N, 3(W-1)+W], recent_pos [N, 3])``
    (reference data_utils.py:91-145 in one kernel).  Scalar metadata only (the reference's generator writes
    per-feature lists of length 1 for temperature, which are accepted)."""
    def scalar(v):
        t = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
        if t.numel() != 1:
            raise CgnnError("window_features: metadata statistics must be scalars")
        return float(t[0])
    pos_seq = f32c(pos_seq, "position window")
    temp_seq = f32c(temp_seq, "temperature window")
    w, n = pos_seq.shape[0], pos_seq.shape[1]
    if pos_seq.shape != (w, n, 3) or temp_seq.numel() != w * n:
        raise CgnnError(f"window_features: expected [W, N, 3] and [W, N(, 1)], got {tuple(pos_seq.shape)} / "
                        f"{tuple(temp_seq.shape)}")
    if pos_noise is not None:
        pos_noise, temp_noise = f32c(pos_noise, "position noise"), f32c(temp_noise, "temperature noise")
        if pos_noise.shape != (n, w, 3) or temp_noise.numel() != n * w:
            raise CgnnError("window_features: noise must be [N, W, 3] / [N, W(, 1)]")
    x = torch.empty((n, 3 * (w - 1) + w), dtype=torch.float32, device=pos_seq.device)
    recent = torch.empty((n, 3), dtype=torch.float32, device=pos_seq.device)
    _same_device(pos_seq, temp_seq, pos_noise, temp_noise)
    with _timed("window_features", pos_seq.device):
        check(_lib.load().cgnn_window_features(pos_seq.data_ptr(), temp_seq.data_ptr(), ptr(pos_noise), ptr(temp_noise), w, n,
                                               float(box_size), float(dt), scalar(metadata["vel_mean"]),
                                               scalar(metadata["vel_std"]), scalar(metadata["temp_mean"]),
                                               scalar(metadata["temp_std"]), x.data_ptr(), recent.data_ptr(),
                                               stream_ptr(pos_seq.device)), "cgnn_window_features")
    return x, recent


def segment_colsum(acc: torch.Tensor, batch: Optional[torch.Tensor], num_graphs: int) -> torch.Tensor:
    acc = f32c(acc, "acc")
    if batch is not None:
        batch = i32c(batch, "batch")
    n, width = acc.shape
    sums = torch.empty((num_graphs, width), dtype=torch.float64, device=acc.device)
    _same_device(acc, batch)
    with _timed("segment_colsum", acc.device):
        check(_lib.load().cgnn_segment_colsum(acc.data_ptr(), ptr(batch), n, width, num_graphs, sums.data_ptr(),
                                              stream_ptr(acc.device)), "cgnn_segment_colsum")
    return sums


def gather_rows(table: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    table, idx = f32c(table, "table"), i32c(idx, "idx")
    if out is None:
        out = torch.empty((idx.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
    _same_device(table, idx, out)
    with _timed("gather_rows", table.device):
        check(_lib.load().cgnn_gather_rows(table.data_ptr(), idx.data_ptr(), idx.numel(), table.shape[1], out.data_ptr(),
                                           stream_ptr(table.device)), "cgnn_gather_rows")
    return out


def scatter_rows(rows: torch.Tensor, idx: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    rows, idx = f32c(rows, "rows"), i32c(idx, "idx")
    require_device(table, "table")
    if not table.is_contiguous() or table.dtype != torch.float32:
        raise CgnnError("scatter_rows: table must be contiguous float32")
    _same_device(rows, idx, table)
    with _timed("scatter_rows", table.device):
        check(_lib.load().cgnn_scatter_rows(rows.data_ptr(), idx.data_ptr(), idx.numel(), table.shape[1],
                                            table.data_ptr(), stream_ptr(table.device)), "cgnn_scatter_rows")
    return table


# ---- backward of the node stream (include/cgnn.h: cgnn_mlp_backward / cgnn_weight_grad / cgnn_col_dot) -----------

class BackwardScratch:
    """The per-layer activation / gradient matrices ``cgnn_mlp_backward`` leaves behind for the parameter
    gradients (``cgnn_mlp_bwd_buffers``).  Sized for the widest MLP it will be used with and reused across calls."""

    def __init__(self, n: int, hidden: int, out_padded: int, num_hidden_layers: int, device):
        self.n, self.hidden, self.out_padded, self.nh = int(n), int(hidden), int(out_padded), int(num_hidden_layers)
        new = lambda w: torch.empty((self.n, w), dtype=torch.float32, device=device)  # noqa: E731
        self.h = [new(hidden) for _ in range(self.nh)]
        self.g_a = [new(hidden) for _ in range(self.nh)]
        self.g_o = new(out_padded)
        self.zhat = new(out_padded)

    def struct(self, nh: int, out_padded: int) -> _lib.MlpBwdBuffers:
        if nh > self.nh or out_padded > self.out_padded:
            raise CgnnError("BackwardScratch is too small for this MLP")
        b = _lib.MlpBwdBuffers()
        for l in range(nh):
            b.h[l] = self.h[l].data_ptr()
            b.g_a[l] = self.g_a[l].data_ptr()
        b.g_o = self.g_o.data_ptr()
        b.zhat = self.zhat.data_ptr()
        return b


def mlp_backward(fwd: PackedMLP, fwd2: Optional[PackedLinear], bwd: PackedMLP, bwd2: Optional[PackedLinear],
                 u1: torch.Tensor, u2: Optional[torch.Tensor], dy: torch.Tensor, scratch: BackwardScratch,
                 want_du1: bool = True, want_du2: bool = True):
    """Data gradients of ``y = [LN](MLP(cat(u1, u2)))`` given ``dy``; fills ``scratch`` (read it with
    :func:`weight_grad` / :func:`col_dot` before the next call).  ``bwd`` / ``bwd2`` hold the transposed weights.
    Returns ``(du1 | None, du2 | None)``.  Note ``scratch.g_o`` / ``scratch.zhat`` rows are ``32*ceil(out/32)`` wide."""
    u1, dy = f32c(u1, "u1"), f32c(dy, "dy")
    n = u1.shape[0]
    if n > scratch.n or fwd.hidden != scratch.hidden:
        raise CgnnError("mlp_backward: scratch does not fit (rows or hidden width)")
    if dy.shape != (n, fwd.out_dim):
        raise CgnnError(f"mlp_backward: dy is {tuple(dy.shape)}, expected {(n, fwd.out_dim)}")
    if u2 is not None:
        u2 = f32c(u2, "u2")
    du1 = torch.empty_like(u1) if want_du1 else None
    du2 = torch.empty_like(u2) if (want_du2 and u2 is not None) else None
    out_padded = (fwd.out_dim + 31) // 32 * 32
    # g_o / zhat are written with a row stride of THIS MLP's padded output width (the scratch is flat memory)
    bufs = scratch.struct(fwd.num_hidden_layers, out_padded)
    s_f2 = fwd2.struct() if fwd2 is not None else None
    s_b2 = bwd2.struct() if bwd2 is not None else None
    with _timed("mlp_backward", u1.device):
        check(_lib.load().cgnn_mlp_backward(
            C.byref(fwd.struct()), C.byref(s_f2) if s_f2 is not None else None, C.byref(bwd.struct()),
            C.byref(s_b2) if s_b2 is not None else None, u1.data_ptr(), u1.stride(0), ptr(u2),
            u2.stride(0) if u2 is not None else 0, dy.data_ptr(), dy.stride(0), n, C.byref(bufs), ptr(du1),
            du1.stride(0) if du1 is not None else 0, ptr(du2), du2.stride(0) if du2 is not None else 0,
            stream_ptr(u1.device)), "cgnn_mlp_backward")
    return du1, du2


_WGRAD_WORKSPACE = {}       # (device, stream) -> workspace of cgnn_weight_grad_x3 (17 MB, contents irrelevant between calls)


def weight_grad(g: torch.Tensor, ld_g: int, out_dim: int, a: torch.Tensor, in_dim: int, n: int, dw: torch.Tensor,
                col0: int = 0, db: Optional[torch.Tensor] = None, precision="fp32") -> torch.Tensor:
    """``dw[:, col0:col0+in_dim] += g[:n, :out_dim]^T a[:n, :in_dim]`` (``dw`` contiguous float32, pre-zeroed by the
    caller on first use); ``db`` (optional, pre-zeroed) ``+= `` the column sums of ``g``.  ``precision="fp32x3"``: a
    128 x 128 product of 16-byte-aligned operands runs on the bf16 matrix cores (three bf16 terms per operand) with a
    fixed summation order (``cgnn_weight_grad_x3``); every other shape takes the f32-MFMA kernel."""
    require_device(g, "g")
    a = f32c(a, "a")
    if dw.dtype != torch.float32 or not dw.is_contiguous() or dw.shape[0] != out_dim:
        raise CgnnError("weight_grad: dw must be contiguous float32 [out_dim, >= col0 + in_dim]")
    if _prec(precision) == _lib.F32X3 and out_dim == 128 and in_dim == 128 and ld_g % 4 == 0 and \
            a.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0 and a.data_ptr() % 16 == 0:
        lib = _lib.load()
        key = (a.device.type, a.device.index, stream_ptr(a.device))     # per stream: calls on one stream are ordered
        ws = _WGRAD_WORKSPACE.get(key)
        if ws is None:
            ws = _WGRAD_WORKSPACE[key] = torch.empty(lib.cgnn_weight_grad_x3_workspace_bytes(), dtype=torch.uint8,
                                                     device=a.device)
        with _timed("weight_grad", a.device):
            check(lib.cgnn_weight_grad_x3(g.data_ptr(), ld_g, a.data_ptr(), a.stride(0), n, dw.data_ptr(), dw.stride(0),
                                          col0, ptr(db), ws.data_ptr(), ws.numel(), stream_ptr(a.device)),
                  "cgnn_weight_grad_x3")
        return dw
    # every other shape: exact f32 MFMA, the row chunks' products added in a fixed order (cgnn_weight_grad_ordered: the same
    # bits on every run; cgnn_weight_grad itself meets in float atomics)
    lib = _lib.load()
    need = lib.cgnn_weight_grad_workspace_bytes(n, out_dim, in_dim)
    key = (a.device.type, a.device.index, stream_ptr(a.device), "ordered")
    ws = _WGRAD_WORKSPACE.get(key)
    if ws is None or ws.numel() < need:
        ws = _WGRAD_WORKSPACE[key] = torch.empty(max(need, 1), dtype=torch.uint8, device=a.device)
    with _timed("weight_grad", a.device):
        check(lib.cgnn_weight_grad_ordered(g.data_ptr(), ld_g, out_dim, a.data_ptr(), a.stride(0), in_dim, n,
                                           dw.data_ptr(), dw.stride(0), col0, ptr(db), ws.data_ptr(), ws.numel(),
                                           stream_ptr(a.device)), "cgnn_weight_grad_ordered")
    return dw


class SenderCsr:
    """Edges grouped by ``key`` (``cgnn_csr_build``): ``row_ptr`` int32 [rows + 1], ``col`` int32 [E]."""

    def __init__(self, key: torch.Tensor, val: Optional[torch.Tensor], num_rows: int):
        lib = _lib.load()
        key = i32c(key, "key")
        ne = key.numel()
        if val is not None:
            val = i32c(val, "val")
            if val.numel() != ne:
                raise CgnnError("SenderCsr: key and val differ in length")
        self.rows = int(num_rows)
        self.row_ptr = torch.empty(self.rows + 1, dtype=torch.int32, device=key.device)
        self.col = torch.empty(max(ne, 1), dtype=torch.int32, device=key.device)
        nbytes = lib.cgnn_csr_workspace_bytes(self.rows)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=key.device)
        with _timed("csr_build", key.device):
            check(lib.cgnn_csr_build(key.data_ptr(), ptr(val), ne, self.rows, self.row_ptr.data_ptr(),
                                     self.col.data_ptr(), ws.data_ptr(), nbytes, stream_ptr(key.device)), "cgnn_csr_build")


def aggregate_csr(table: torch.Tensor, csr: SenderCsr, out: Optional[torch.Tensor] = None,
                  add1: Optional[torch.Tensor] = None, add2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[r] = (add1[r] + add2[r] +) sum_{p in row r} table[csr.col[p]]``; ``out`` may be ``add1`` or ``add2``."""
    table = f32c(table, "table")
    shape = (csr.rows, table.shape[1])
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=table.device)
    for t, name in ((add1, "add1"), (add2, "add2"), (out, "out")):
        if t is not None and (tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous()
                              or t.device != table.device):
            raise CgnnError(f"aggregate_csr: {name} must be contiguous float32 {shape} on the table's device")
    with _timed("aggregate_csr", table.device):
        check(_lib.load().cgnn_aggregate_csr_add(table.data_ptr(), csr.row_ptr.data_ptr(), csr.col.data_ptr(), csr.rows,
                                                 table.shape[1], ptr(add1), ptr(add2), out.data_ptr(),
                                                 stream_ptr(table.device)), "cgnn_aggregate_csr_add")
    return out


_COLDOT_WORKSPACE = {}


def _col_dot_workspace(a: torch.Tensor, n: int, width: int) -> torch.Tensor:
    """Partial-sum scratch of the fixed-order column sums: one per (device, stream), grown on demand."""
    need = _lib.load().cgnn_col_dot_workspace_bytes(n, width)
    key = (a.device.type, a.device.index, stream_ptr(a.device))     # per stream: calls on one stream are ordered
    ws = _COLDOT_WORKSPACE.get(key)
    if ws is None or ws.numel() < need:
        ws = _COLDOT_WORKSPACE[key] = torch.empty(max(need, 1), dtype=torch.uint8, device=a.device)
    return ws


def col_dot(a: torch.Tensor, ld_a: int, b: Optional[torch.Tensor], ld_b: int, n: int, width: int,
            out: torch.Tensor) -> torch.Tensor:
    """``out[c] += sum_r a[r, c] * (b[r, c] if b is not None else 1)``, summed in a fixed order (the same bits on
    every run: ``cgnn_col_dot_ordered``)."""
    require_device(a, "a")
    if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() < width:
        raise CgnnError("col_dot: out must be contiguous float32 [width]")
    ws = _col_dot_workspace(a, n, width)
    with _timed("col_dot", a.device):
        check(_lib.load().cgnn_col_dot_ordered(a.data_ptr(), ld_a, ptr(b), ld_b, n, width, out.data_ptr(), None,
                                               ws.data_ptr(), ws.numel(), stream_ptr(a.device)), "cgnn_col_dot_ordered")
    return out


def col_dot2(a: torch.Tensor, ld_a: int, b: torch.Tensor, ld_b: int, n: int, width: int, out_ab: torch.Tensor,
             out_a: torch.Tensor):
    """``out_ab[c] += sum_r a[r, c] * b[r, c]`` and ``out_a[c] += sum_r a[r, c]`` in one pass over ``a`` (LayerNorm's
    ``dgamma`` and ``dbeta`` from ``dy`` and ``zhat``), both in a fixed order (reproducible)."""
    require_device(a, "a")
    require_device(b, "b")
    for t in (out_ab, out_a):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < width:
            raise CgnnError("col_dot2: outputs must be contiguous float32 [width]")
    ws = _col_dot_workspace(a, n, width)
    with _timed("col_dot", a.device):
        check(_lib.load().cgnn_col_dot_ordered(a.data_ptr(), ld_a, b.data_ptr(), ld_b, n, width, out_ab.data_ptr(),
                                               out_a.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(a.device)),
              "cgnn_col_dot_ordered")
    return out_ab, out_a

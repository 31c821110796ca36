"""Multi-GPU message passing: spatial tiles + one-hop halo exchange of node latents.

The reference is single-process; this is new design for one
node of 8 MI355X.  The periodic box is cut into ``world`` spatial tiles, one per rank
(one process per GPU).  An edge belongs to its receiver's rank, so edge latents never
move; what a round needs from other ranks is the latent row ``x[src]`` of every
sender that lives elsewhere (a *ghost*).  Per round:

    pack owned rows peers asked for  ->  all-to-all-v over RCCL/xGMI  ->  ghost rows

``torch.distributed.all_to_all_single`` with split sizes is exactly the grouped
send/recv this needs: in a 2x2x2 periodic tiling every rank neighbours all 7 others,
one peer per xGMI link, so all links carry traffic at once and nothing is ring-bound.
The receive buffer *is* the ghost block of the local latent table (ghosts are stored
grouped by owner rank), so there is no unpack pass.

Index bookkeeping (ownership, ghost lists, global->local maps) is host logic done once
per graph with torch indexing; the per-round data path is HIP kernels + RCCL.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import _lib, ops, synthetic


# ----------------------------------------------------------------------------
# tiling
# ----------------------------------------------------------------------------

def tile_grid(world: int) -> Tuple[int, int, int]:
    """Near-cubic factorisation: 1->(1,1,1) 2->(2,1,1) 4->(2,2,1) 8->(2,2,2) ..."""
    dims = [1, 1, 1]
    n, p = world, 2
    factors = []
    while n > 1:
        while n % p == 0:
            factors.append(p)
            n //= p
        p += 1
    for f in sorted(factors, reverse=True):
        dims[dims.index(min(dims))] *= f
    return tuple(sorted(dims, reverse=True))


def owner_of(pos: torch.Tensor, box_size: float, world: int) -> torch.Tensor:
    """Rank owning each particle: the tile of the periodic box that contains it."""
    px, py, pz = tile_grid(world)
    g = torch.tensor([px, py, pz], device=pos.device, dtype=torch.float32)
    cell = torch.floor(pos / box_size * g).to(torch.int64)
    cell = torch.minimum(cell.clamp_min_(0), (g - 1).to(torch.int64))
    return ((cell[:, 0] * py + cell[:, 1]) * pz + cell[:, 2]).to(torch.int32)


# ----------------------------------------------------------------------------
# shard description
# ----------------------------------------------------------------------------

@dataclass
class Shard:
    rank: int
    world: int
    k: int
    n_owned: int
    n_ghost: int
    owned_global: torch.Tensor        # int64 [n_owned]  global ids, in the local (spatial) order
    ghost_global: torch.Tensor        # int64 [n_ghost]  grouped by owner rank
    src_local: torch.Tensor           # int32 [n_owned*k] rows of the local table [owned | ghosts]
    dst_local: torch.Tensor           # int32 [n_owned*k]
    edge_attr: torch.Tensor           # [n_owned*k, 4]
    recv_counts: List[int]            # ghost rows coming from each rank
    want_global: List[torch.Tensor] = field(default_factory=list)   # ids requested from each rank
    send_idx: Optional[torch.Tensor] = None    # int32 local owned rows to pack, grouped by destination
    send_counts: Optional[List[int]] = None
    x_feat: Optional[torch.Tensor] = None      # [n_owned, F] encoder inputs of the owned particles
    knn_ms: float = 0.0
    n_interior: int = 0               # owned rows [0, n_interior) have only owned senders (no halo needed)

    @property
    def n_local(self) -> int:
        return self.n_owned + self.n_ghost


def tile_bounds(box_size: float, world: int, rank: int):
    """``(lo [3], hi [3])`` of rank ``rank``'s tile (the inverse of :func:`owner_of`)."""
    px, py, pz = tile_grid(world)
    ix, iy, iz = rank // (py * pz), (rank // pz) % py, rank % pz
    lo = [ix * box_size / px, iy * box_size / py, iz * box_size / pz]
    hi = [(ix + 1) * box_size / px, (iy + 1) * box_size / py, (iz + 1) * box_size / pz]
    return lo, hi


def _near_tile(pos: torch.Tensor, box_size: float, lo, hi, margin: float) -> torch.Tensor:
    """Particles within ``margin`` of the tile [lo, hi) along every axis, periodic (a superset of the margin ball)."""
    keep = torch.ones(pos.shape[0], dtype=torch.bool, device=pos.device)
    for a in range(3):
        width = hi[a] - lo[a]
        if width + 2 * margin >= box_size:
            continue                                  # the expanded tile covers this axis
        c = 0.5 * (lo[a] + hi[a])
        d = torch.abs(pos[:, a] - c)
        d = torch.minimum(d, box_size - d)            # periodic distance to the tile centre
        keep &= d <= 0.5 * width + margin
    return keep


def build_shard(pos_global: torch.Tensor, box_size: float, k: int, world: int, rank: int,
                knn_fn: Optional[Callable] = None, margin_factor: float = 2.0) -> Shard:
    """Everything rank ``rank`` can derive locally from the global positions: its owned set, their k-NN
    senders, the ghost set and the global->local renumbering.  ``knn_fn(pos, box, k, query_ids)`` defaults to
    the HIP k-NN; it returns ``(senders int32 [nq*k], edge_attr [nq*k, 4], order)``.

    The neighbour search runs over the rank's tile plus a margin, not over the whole box ("graph
    build"): margin = ``margin_factor`` x the radius that holds k particles at mean density.  It is then CHECKED --
    every owned particle's k-th neighbour must be closer than the margin, or some neighbour outside the subset could
    have been missed -- and doubled until the check holds (clustered inputs), so the result is the global one."""
    dev = pos_global.device
    n_total = pos_global.shape[0]
    owner = owner_of(pos_global, box_size, world)
    knn = knn_fn or (lambda p, b, kk, q: ops.knn_periodic(p, b, kk, query_ids=q, want_edge_attr=True,
                                                          want_order=True))
    t0 = time.perf_counter()
    search_ms = 0.0

    def timed_knn(p, b, kk, q):     # device time of the neighbour search alone (the torch glue around it is host-bound)
        nonlocal search_ms
        if dev.type != "cuda":
            return knn(p, b, kk, q)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = knn(p, b, kk, q)
        e1.record()
        e1.synchronize()
        search_ms += e0.elapsed_time(e1)
        return out

    lo, hi = tile_bounds(box_size, world, rank)
    margin = margin_factor * box_size * (3.0 * k / (4.0 * 3.141592653589793 * max(n_total, 1))) ** (1.0 / 3.0)
    while True:
        near = _near_tile(pos_global, box_size, lo, hi, margin) if world > 1 else None
        if near is not None:
            near |= owner == rank      # owner_of clamps coordinates outside [0, box) into edge tiles: owned is always searched
        whole = near is None or bool(near.all())
        sub = None if whole else torch.nonzero(near).squeeze(1)        # ascending global ids: ties order as globally
        pos_sub = pos_global if whole else pos_global[sub].contiguous()
        own_sub = owner if whole else owner[sub]
        owned_s = torch.nonzero(own_sub == rank).squeeze(1)            # indices into the subset
        # a one-query pass builds the cell grid and yields the spatial (cell-sorted) order, so that the local
        # numbering is cache friendly; then the real pass over the owned queries in that order
        if owned_s.numel():
            _, _, order = timed_knn(pos_sub, box_size, k, owned_s[:1].to(torch.int32))
            if order is not None:
                order = order.long()
                owned_s = order[own_sub[order] == rank]
        senders_s, edge_attr, _ = timed_knn(pos_sub, box_size, k, owned_s.to(torch.int32))
        if whole or owned_s.numel() == 0:
            break
        # k-th neighbour distance (minimum image) of every owned particle against the margin
        kth = senders_s.view(-1, k)[:, k - 1].long()
        dlt = torch.abs(pos_sub[kth] - pos_sub[owned_s])
        dlt = torch.minimum(dlt, box_size - dlt)
        if float(dlt.norm(dim=1).max()) <= margin:
            break
        margin *= 2.0
    owned = owned_s if whole else sub[owned_s]
    senders = senders_s.long() if whole else sub[senders_s.long()]
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    build_ms = (time.perf_counter() - t0) * 1e3          # selection of the tile + margin subset, both searches, the check
    knn_ms = search_ms if dev.type == "cuda" else build_ms
    n_owned = owned.numel()
    senders = senders.long()
    remote = owner[senders] != rank
    # interior receivers (every sender owned) first, boundary receivers last, spatial order kept inside each group:
    # a round's interior half can then run while the halo exchange is still in flight
    is_boundary = remote.view(n_owned, k).any(dim=1) if n_owned else remote.new_zeros((0,))
    n_interior = int((~is_boundary).sum())
    if 0 < n_interior < n_owned:
        regroup = torch.cat([torch.nonzero(~is_boundary).squeeze(1), torch.nonzero(is_boundary).squeeze(1)])
        owned = owned[regroup]
        senders = senders.view(n_owned, k)[regroup].reshape(-1)
        edge_attr = edge_attr.view(n_owned, k, -1)[regroup].reshape(n_owned * k, -1).contiguous()
        remote = remote.view(n_owned, k)[regroup].reshape(-1)
    ghosts = torch.unique(senders[remote])
    g_owner = owner[ghosts].long()
    perm = torch.argsort(g_owner * n_total + ghosts)      # group by owner rank, ascending id inside
    ghosts = ghosts[perm]
    g_owner = g_owner[perm]
    recv_counts = torch.bincount(g_owner, minlength=world).tolist()
    g2l = torch.full((n_total,), -1, dtype=torch.int32, device=dev)
    g2l[owned] = torch.arange(n_owned, dtype=torch.int32, device=dev)
    g2l[ghosts] = n_owned + torch.arange(ghosts.numel(), dtype=torch.int32, device=dev)
    src_local = g2l[senders].contiguous()
    dst_local = torch.arange(n_owned, dtype=torch.int32, device=dev).repeat_interleave(k)
    want = list(torch.split(ghosts, recv_counts))
    sh = Shard(rank, world, k, n_owned, ghosts.numel(), owned, ghosts, src_local, dst_local, edge_attr,
               recv_counts, want_global=want, knn_ms=knn_ms, n_interior=n_interior)
    sh._g2l = g2l
    sh.subset_build_ms = build_ms
    return sh


def finish_shard(sh: Shard, requests_from_peers: Sequence[torch.Tensor]) -> Shard:
    """``requests_from_peers[r]`` = global ids rank r wants from us -> local rows to pack, grouped by r."""
    idx = [sh._g2l[req.long()] for req in requests_from_peers]
    for r, t in enumerate(idx):
        if t.numel() and (int(t.min()) < 0 or int(t.max()) >= sh.n_owned):
            raise RuntimeError(f"rank {sh.rank}: rank {r} requested rows this rank does not own")
    sh.send_counts = [int(t.numel()) for t in idx]
    sh.send_idx = torch.cat(idx).to(torch.int32).contiguous() if idx else torch.empty(0, dtype=torch.int32)
    return sh


def _comm_device(tensor_device: torch.device, group=None) -> torch.device:
    """Device the process group moves data on: the tensors' own device with RCCL ("nccl"); the host with gloo
    (CPU tests, and the single-GPU rehearsal of the multi-rank path, which stages through host memory)."""
    import torch.distributed as dist
    return tensor_device if dist.get_backend(group) == "nccl" else torch.device("cpu")


def exchange_requests(sh: Shard, group=None) -> Shard:
    """Setup-time all-to-all of the ghost id lists (sizes, then ids)."""
    import torch.distributed as dist
    dev = sh.owned_global.device
    cdev = _comm_device(dev, group)
    counts_out = torch.tensor(sh.recv_counts, dtype=torch.int64, device=cdev)
    counts_in = torch.empty_like(counts_out)
    dist.all_to_all_single(counts_in, counts_out, group=group)
    counts_in_l = counts_in.tolist()
    recv = torch.empty(int(sum(counts_in_l)), dtype=torch.int64, device=cdev)
    dist.all_to_all_single(recv, sh.ghost_global.to(cdev).contiguous(), output_split_sizes=counts_in_l,
                           input_split_sizes=sh.recv_counts, group=group)
    return finish_shard(sh, [t.to(dev) for t in torch.split(recv, counts_in_l)])


# ----------------------------------------------------------------------------
# halo exchange (per round)
# ----------------------------------------------------------------------------

class HaloExchange:
    """Fills the ghost block ``table[n_owned:]`` from the owners' rows ``table[:n_owned]``."""

    def __init__(self, sh: Shard, group=None, pack_fn: Optional[Callable] = None):
        self.sh, self.group = sh, group
        self.pack = pack_fn or (lambda table, idx, out: ops.gather_rows(table, idx, out))
        self._buf = None

    def start(self, table: torch.Tensor):
        """Pack the rows the peers asked for and start the all-to-all into ``table``'s ghost block; returns a handle
        for :meth:`finish`.  Kernels enqueued in between run under the exchange (they must not touch ghost rows)."""
        import torch.distributed as dist
        sh = self.sh
        width = table.shape[1]
        if self._buf is None or self._buf.shape != (sh.send_idx.numel(), width) or self._buf.device != table.device:
            self._buf = torch.empty((sh.send_idx.numel(), width), dtype=table.dtype, device=table.device)
        if sh.send_idx.numel():
            self.pack(table, sh.send_idx, self._buf)
        ghosts = table[sh.n_owned:]
        cdev = _comm_device(table.device, self.group)
        if cdev == table.device:
            work = dist.all_to_all_single(ghosts, self._buf, output_split_sizes=sh.recv_counts,
                                          input_split_sizes=sh.send_counts, group=self.group, async_op=True)
            return (work, None, ghosts)
        # gloo rehearsal with device tensors: stage through host memory
        recv = torch.empty(ghosts.shape, dtype=ghosts.dtype, device=cdev)
        work = dist.all_to_all_single(recv, self._buf.to(cdev), output_split_sizes=sh.recv_counts,
                                      input_split_sizes=sh.send_counts, group=self.group, async_op=True)
        return (work, recv, ghosts)

    def finish(self, handle) -> None:
        work, recv, ghosts = handle
        work.wait()                       # RCCL: the current stream waits for the exchange
        if recv is not None:
            ghosts.copy_(recv)

    def __call__(self, table: torch.Tensor) -> None:
        self.finish(self.start(table))


# ----------------------------------------------------------------------------
# sharded forward
# ----------------------------------------------------------------------------

class ShardedForward:
    """``EncodeProcessDecode.forward`` over one spatial tile.  ``halo(table)`` must fill the ghost rows of
    ``table`` ([n_owned + n_ghost, D]) from their owners; by default it is the RCCL all-to-all above.
    Returns the predictions of the owned particles (local order; ``shard.owned_global`` maps them back)."""

    def __init__(self, model, shard: Shard, halo: Optional[Callable] = None):
        self.model, self.sh = model, shard
        self.halo = halo if halo is not None else HaloExchange(shard)
        self._bufs = None

    def _buffers(self, D: int, H: int, dev):
        sh = self.sh
        key = (D, H, dev, self.model.edge_precision, self.fused, self.p_fmt)
        if self._bufs is None or self._bufs[0] != key:
            x_all = torch.empty((sh.n_local, D), dtype=torch.float32, device=dev)
            pdt = ops.p_format_dtype(self.p_fmt) if self.P["rounds"] else torch.float32
            if self.fused:      # every round's tables are kept for the one-launch edge stream (same row stride for both)
                L = len(self.P["rounds"])
                ps = torch.empty((L, sh.n_local, H), dtype=pdt, device=dev)
                pd = torch.empty((L, sh.n_local, H), dtype=pdt, device=dev)
            else:
                ps = torch.empty((sh.n_local, H), dtype=pdt, device=dev)
                pd = torch.empty((sh.n_owned, H), dtype=pdt, device=dev)
            agg = torch.empty((sh.n_owned, D), dtype=torch.float32, device=dev)
            # fused mode updates the node latents out of place (interior rows are rewritten while boundary receivers
            # still gather the old ones): a second table, swapped every round
            x_alt = torch.empty_like(x_all) if self.fused else None
            self._bufs = (key, x_all, ps, pd, agg, x_alt)
        return self._bufs[1:]

    # the pieces are separate methods so that a single-process test can interleave several shards
    def encode(self):
        m, sh = self.model, self.sh
        P = m._pack(sh.x_feat.shape[1], sh.edge_attr.shape[1])
        D = m._latent_size
        H = P["rounds"][0].ws.out_dim if P["rounds"] else D
        self.P = P
        # reference data flow (x_j): the node stream runs round by round with its halo exchanges and leaves every
        # round's Ps / Pd behind; the edge stream then is one launch (cgnn_edge_stream), as on one GPU
        self.image = P["image"]          # cgnn_edge_stream_run's chunk image (None: first-generation kernel or per round)
        self.fused = self.image is not None or m._can_fuse_rounds(P["rounds"], D)
        # fused mode: the table format the one-launch edge stream of this shard will take (fp16 rows for the
        # two-waves-per-SIMD kernel, graph_network.stream_table_format); per round: each round's own
        self.p_fmt = P["rounds"][0].p_format if P["rounds"] else _lib.P_F32
        if self.fused and self.image is not None:
            from .graph_network import stream_table_format
            attr0 = sh.edge_attr if bool(self.image.enc_in) else None
            _, kern = m._edge_stream_plan(P, sh.k, sh.src_local.numel(), attr0)
            self.p_fmt = stream_table_format(P["rounds"], kern, int(getattr(m, "edge_stream_lag", 0)))
        self.x_all, self.ps, self.pd, self.agg, self.x_alt = self._buffers(D, H, sh.x_feat.device)
        ops.mlp_rows(P["enc_node"], sh.x_feat, out=self.x_all[:sh.n_owned])
        self._projected = False
        self._edges_pending = False
        # fused mode: the edge encoder runs inside cgnn_edge_stream when it has the rounds' shape
        self._enc_in_stream = bool(self.image.enc_in) if self.image is not None else \
            (self.fused and m._encoder_fits_stream(P))
        self.el = None if self._enc_in_stream else ops.mlp_rows(P["enc_edge"], sh.edge_attr, tiled=True)
        self.e_upd = self.el.empty_like() if m.message_source == "edge" else None

    def round(self, i: int):
        if self.fused:
            return self._round_nodes(i)
        m, sh = self.model, self.sh
        rounds = self.P["rounds"]
        p = rounds[i]
        x_own = self.x_all[:sh.n_owned]
        ps_own, ps_ghost = self.ps[:sh.n_owned], self.ps[sh.n_owned:]
        # sender projections of the ghost rows that just arrived (receivers are always owned: no Pd for ghosts)
        if sh.n_ghost:
            ops.project_nodes(p.ws, None, self.x_all[sh.n_owned:], ps_ghost, None, p.p_format)
        if not self._projected:     # first round: the owned rows too (later rounds: emitted by the node kernel)
            ops.project_nodes(p.ws, p.wd, x_own, ps_own, self.pd, p.p_format)
        edge_mode = m.message_source == "edge"
        if edge_mode and p.edge.precision == _lib.BF16_N16 and sh.k in (8, 16) and self.x_all.shape[1] <= 128:   # aggregation folded in
            ops.edge_block(p.edge, self.ps, self.pd, sh.src_local, sh.dst_local, self.el, self.el, None, True,
                           agg_out=self.agg, x_gather=None, seg_k=sh.k)
        else:
            ops.edge_block(p.edge, self.ps, self.pd, sh.src_local, sh.dst_local, self.el, self.el,
                           self.e_upd if edge_mode else None, True)
            if edge_mode:
                ops.aggregate(self.e_upd, None, sh.dst_local, sh.n_owned, sh.k, sh.src_local.numel(), self.agg)
            else:
                ops.aggregate(self.x_all, sh.src_local, sh.dst_local, sh.n_owned, sh.k, sh.src_local.numel(),
                              self.agg, plan=ops.AggregatePlan.of(sh.src_local, sh.n_owned, sh.k, self.x_all.shape[1]))
        nxt = None
        if i + 1 < len(rounds):
            q = rounds[i + 1]
            if q.p_format == p.p_format and q.p_dtype == self.ps.dtype:
                fused_ok = p.node.precision in _lib.N16_NODE and q.ws_fused.precision == _lib.BF16_N16
                nxt = (q.ws_fused if fused_ok else q.ws, q.wd_fused if fused_ok else q.wd, ps_own, self.pd, q.p_format)
        ops.node_block(p.node, p.wx, p.wa, x_own, self.agg, x_own, True, nxt)
        self._projected = nxt is not None

    def _interior_launch_rows(self) -> int:
        """Rows of the "interior" launches: ``n_interior`` rounded DOWN to whole grid waves of the node kernel.  A launch of
        s 128-row steps on c workgroups (one per CU) takes ceil(s / c) step times, so the remainder of the interior rows cost
        a whole extra step time there -- 849 steps on 256 CUs: four step times for 3.3 of work at 125 k rows per rank -- and
        nothing in the boundary launch, which is far from full; rows that need no ghost may run after the exchange just as well."""
        ni = self.sh.n_interior
        cached = self.__dict__.get("_ni_launch")
        if cached is not None and cached[0] == ni:
            return cached[1]
        dev = self.sh.x_feat.device
        rows = ni
        if dev.type == "cuda":
            wave_rows = 128 * torch.cuda.get_device_properties(dev).multi_processor_count
            if ni >= wave_rows:
                rows = ni - ni % wave_rows
        self._ni_launch = (ni, rows)
        return rows

    def _src_part(self, a: int, b: int) -> torch.Tensor:
        """The sender list of owned receivers [a, b) as one tensor OBJECT per part (the aggregation plan is cached on it)."""
        parts = self.__dict__.setdefault("_src_parts", {})
        if (a, b) not in parts:
            k = self.sh.k
            parts[(a, b)] = self.sh.src_local[a * k:b * k]
        return parts[(a, b)]

    def _round_nodes(self, i: int, part: str = "all"):
        """Fused mode: the node half of round ``i`` for the owned rows of ``part``: ``"interior"`` (receivers whose
        senders are all owned: needs no ghost row, runs under the halo exchange), ``"boundary"`` (the rest, after the
        exchange; also projects the ghost rows) or ``"all"``.  Reads ``x_all``, writes ``x_alt``; the tables swap
        when the round is complete."""
        sh = self.sh
        rounds = self.P["rounds"]
        p = rounds[i]
        no, ni, k = sh.n_owned, self._interior_launch_rows(), sh.k
        a, b = {"all": (0, no), "interior": (0, ni), "boundary": (ni, no)}[part]
        if part != "interior" and sh.n_ghost:
            ops.project_nodes(p.ws, None, self.x_all[no:], self.ps[i][no:], None, self.p_fmt)
        if b > a:
            x_in = self.x_all[a:b]
            if i == 0:
                ops.project_nodes(p.ws, p.wd, x_in, self.ps[0][a:b], self.pd[0][a:b], self.p_fmt)
            src_part = self._src_part(a, b)
            ops.aggregate(self.x_all, src_part, None, b - a, k, (b - a) * k, self.agg[a:b],
                          plan=ops.AggregatePlan.of(src_part, b - a, k, self.x_all.shape[1]))
            nxt = None
            if i + 1 < len(rounds):
                q = rounds[i + 1]
                fused_ok = p.node.precision in _lib.N16_NODE and q.ws_fused.precision == _lib.BF16_N16
                nxt = (q.ws_fused if fused_ok else q.ws, q.wd_fused if fused_ok else q.wd, self.ps[i + 1][a:b],
                       self.pd[i + 1][a:b], self.p_fmt)
            ops.node_block(p.node, p.wx, p.wa, x_in, self.agg[a:b], self.x_alt[a:b], True, nxt)
        if part != "interior":
            self.x_all, self.x_alt = self.x_alt, self.x_all
            self._edges_pending = i + 1 == len(rounds)

    def finish_edges(self):
        """Fused mode: all edge updates in one launch, once the last round's node half has run."""
        if self.fused and self._edges_pending:
            sh = self.sh
            enc = self.P["enc_edge"] if self._enc_in_stream else None
            if self.image is not None:
                attr = sh.edge_attr if self._enc_in_stream else None
                image, kernel = self.model._edge_stream_plan(self.P, sh.k, sh.src_local.numel(), attr)
                self.el = ops.edge_stream_run(image, self.ps, self.pd, sh.src_local, sh.dst_local, self.el, self.el, attr,
                                              kernel=kernel, lag=int(getattr(self.model, "edge_stream_lag", 0)), fixed_k=sh.k)
            else:
                self.el = ops.edge_stream([p.edge for p in self.P["rounds"]], self.ps, self.pd, sh.src_local,
                                          sh.dst_local, self.el, self.el, enc, sh.edge_attr if enc is not None else None)
            self._edges_pending = False

    def decode(self) -> dict:
        self.finish_edges()
        x_own = self.x_all[:self.sh.n_owned]
        return {"acceleration": ops.mlp_rows(self.P["dec_acc"], x_own),
                "temp_rate": ops.mlp_rows(self.P["dec_tr"], x_own)}

    def __call__(self) -> dict:
        with torch.no_grad():
            self.encode()
            n_rounds = len(self.P["rounds"])
            overlap = self.fused and hasattr(self.halo, "start") and 0 < self.sh.n_interior
            for i in range(n_rounds):
                # x_j aggregation and the sender projections both read ghost latents of the current round
                if overlap:     # interior receivers need no ghost row: their half of the round hides the exchange
                    handle = self.halo.start(self.x_all)
                    self._round_nodes(i, "interior")
                    self.halo.finish(handle)
                    self._round_nodes(i, "boundary")
                else:
                    self.halo(self.x_all)
                    self.round(i)
            return self.decode()


# ----------------------------------------------------------------------------
# synthetic shard for bench.py (every rank regenerates the same global box from the seed)
# ----------------------------------------------------------------------------

def build_synthetic_shard(particles_per_gpu: int, world: int, rank: int, k: int, seed: int, device, metadata: dict,
                          group=None) -> Shard:
    n_total = particles_per_gpu * world
    # the box of synthetic.make_snapshot(n_total, seed), bit for bit, but only one global frame (positions: ownership and the
    # neighbour search need all of them) and the feature window of the OWNED particles are built and uploaded: per rank the
    # host work is the random draws plus N + 6 N / world elements, not the whole [6, N, 3] trajectory
    snap = synthetic.LazySnapshot(n_total, seed=seed)
    box, dt = metadata["box_size"], metadata["dt"]
    W = 5
    pos = torch.remainder(snap.frame(W - 1).to(device), box).contiguous()     # the window's last frame
    sh = build_shard(pos, box, k, world, rank)
    sh = exchange_requests(sh, group)
    coords, energy = snap.window_of(sh.owned_global)
    # node features of the owned particles: the same kernel data_utils.preprocess uses
    sh.x_feat, _ = ops.window_features(coords[:W].to(device).contiguous(), energy[:W].to(device).contiguous(), metadata, dt, box)
    return sh

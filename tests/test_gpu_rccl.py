"""GPU: the multi-GPU path's RCCL calls, executed for real on one GPU.

No reference counterpart (the reference is single-process).  A one-rank ``nccl`` process group is brought up in
this process (``init_process_group("nccl", device_id=...)`` on 127.0.0.1) and the code that ``bench.py --gpus N`` runs per
rank goes through it unchanged: ``build_synthetic_shard`` (setup all-to-all of the ghost lists on device tensors),
``HaloExchange.start`` / ``finish`` (``all_to_all_single(..., async_op=True)`` into a view of the node table, the work
handle's ``wait()`` ordering the launch stream behind RCCL's) and ``ShardedForward``.  World 1 has no peers, so besides the
empty exchange of the real shard a hand-built shard sends rows to ITSELF: real bytes through RCCL, checked bit for bit.
Several ranks on several GPUs only run in the driver's scaling bench; their index logic is covered by the gloo tests
(tests/test_dist_cpu.py) and the single-GPU loopback (tests/test_gpu_dist.py)."""
import socket

import pytest
import torch

from cosmology_gnn_simulation_amd import data_utils, dist as cdist, graph_network, ops, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world_of_one():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    dev = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def test_halo_exchange_moves_rows_through_rccl(nccl_world_of_one):
    """A shard whose only peer is itself: rows 3, 0, 7, ... of the owned block must land in the ghost block, through
    all_to_all_single on device tensors with async_op=True, with kernels enqueued between start() and finish()."""
    dev = nccl_world_of_one
    n_owned, n_ghost, width = 4096, 1000, 128
    gen = torch.Generator(device=dev).manual_seed(3)
    send_idx = torch.randint(0, n_owned, (n_ghost,), device=dev, generator=gen, dtype=torch.int32)
    sh = cdist.Shard(rank=0, world=1, k=16, n_owned=n_owned, n_ghost=n_ghost,
                     owned_global=torch.arange(n_owned, device=dev), ghost_global=send_idx.long(),
                     src_local=torch.zeros(1, dtype=torch.int32, device=dev), dst_local=torch.zeros(1, dtype=torch.int32, device=dev),
                     edge_attr=torch.zeros(1, 4, device=dev), recv_counts=[n_ghost], send_idx=send_idx, send_counts=[n_ghost])
    table = torch.randn(n_owned + n_ghost, width, device=dev, generator=gen)
    table[n_owned:] = float("nan")
    want = table[:n_owned][send_idx.long()].clone()
    halo = cdist.HaloExchange(sh)
    for _ in range(3):                                  # repeated use of the pack buffer and of the group
        table[n_owned:] = float("nan")
        handle = halo.start(table)
        busy = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)      # work under the exchange
        halo.finish(handle)
        got = table[n_owned:].clone()                   # enqueued behind work.wait(): must see the exchanged rows
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        assert bool(torch.isfinite(busy).all())


def test_sharded_forward_over_rccl_equals_unsharded(nccl_world_of_one):
    """bench.py's per-rank path on a world of one: setup all-to-all, per-round (empty) halo exchanges with the
    interior / boundary overlap, one-launch edge stream; owned rows bit-identical to the plain forward."""
    dev = nccl_world_of_one
    n, k, d, L = 30000, 16, 128, 3
    meta = synthetic.make_metadata()
    sh = cdist.build_synthetic_shard(n, 1, 0, k, 77, dev, meta)
    assert sh.n_owned == n and sh.n_ghost == 0 and sh.send_counts == [0] and sh.recv_counts == [0]
    model = graph_network.EncodeProcessDecode(d, d, 2, L, 3)
    model.load_state_dict(synthetic.make_state_dict(d, d, 2, L, 3))
    model = model.to(dev).eval()
    model.edge_precision, model.node_precision = "bf16", "fp16x2"       # bench.py's preset
    runner = cdist.ShardedForward(model, sh)                             # default halo: HaloExchange over the default group
    out = runner()
    assert isinstance(runner.halo, cdist.HaloExchange) and runner.fused
    snap = synthetic.make_snapshot(n, seed=77)
    g = data_utils.preprocess(snap["Coordinates"][:5], snap["InternalEnergy"][:5], meta, None, None, 0.0, k, meta["dt"],
                              meta["box_size"], device=dev)
    with torch.no_grad():
        want = model(g)
    torch.cuda.synchronize()
    for key in ("acceleration", "temp_rate"):
        assert torch.equal(out[key], want[key][sh.owned_global]), key

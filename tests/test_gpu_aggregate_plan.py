"""GPU: the planned fixed-k aggregation (cgnn_aggregate_planned: a block's distinct sender rows staged once in LDS)
against cgnn_aggregate's fixed-k kernel -- same summation order, so the results must be bit-identical -- and against
the oracle's propagate_add (reference graph_network.py:92)."""
import pytest
import torch

from cosmology_gnn_simulation_amd import data_utils, ops, synthetic
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _knn_senders(n, k, seed):
    gen = torch.Generator().manual_seed(seed)
    pos = torch.rand(n, 3, generator=gen).to(DEV)
    snd, _, order = ops.knn_periodic(pos, 1.0, k, want_edge_attr=False, want_order=True)
    # particles in the engine's spatial (cell) order, as the model's locality plan arranges them
    inv = torch.empty_like(order)
    inv[order.long()] = torch.arange(n, device=DEV, dtype=order.dtype)
    return inv[snd.view(n, k)[order.long()].long()].reshape(-1).contiguous().int()


@pytest.mark.parametrize("n,k,width", [(20000, 16, 128), (9001, 16, 128), (12345, 8, 64), (10000, 5, 32), (8200, 32, 256),
                                       (16384, 16, 256)])
def test_planned_aggregation_is_bit_identical_on_spatial_graphs(n, k, width):
    src = _knn_senders(n, k, n + k)
    x = torch.randn(n, width, device=DEV)
    plain = ops.aggregate(x, src, None, n, k)
    plan = ops.AggregatePlan(src, n, k)
    got = ops.aggregate(x, src, None, n, k, plan=plan)
    assert torch.equal(got, plain)
    rows = 64 if k in (8, 16) else 32                                  # receivers per block
    counts = plan.blob[: 4 * ((n + rows - 1) // rows)].view(torch.int32)
    assert int((counts > 0).sum()) == counts.numel()                   # no block exceeds the list ...
    assert float((counts <= 352).float().mean()) > 0.9                 # ... nearly all are staged in LDS ...
    assert float(counts.float().mean()) < 0.6 * rows * k               # ... and hold far fewer rows than references
    dst = torch.arange(n).repeat_interleave(k)
    want = cpu_ref.propagate_add(x.cpu(), torch.stack([src.cpu().long(), dst]))
    assert float((got.cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("k", [16, 8, 11])
def test_blocks_with_too_many_distinct_senders_take_the_direct_path(k):
    n, width = 10000, 128
    gen = torch.Generator().manual_seed(k)
    src = torch.randint(0, n, (n * k,), generator=gen).int().to(DEV)     # 64 k references per block, almost all distinct
    x = torch.randn(n, width, device=DEV)
    plan = ops.AggregatePlan(src, n, k)
    rows = 64 if k in (8, 16) else 32
    counts = plan.blob[: 4 * ((n + rows - 1) // rows)].view(torch.int32)
    assert int(((counts < 0) | (counts > 352)).sum()) > 0 or rows * k <= 352
    assert torch.equal(ops.aggregate(x, src, None, n, k, plan=plan), ops.aggregate(x, src, None, n, k))


def test_plan_is_bound_to_its_sender_list():
    n, k = 9000, 16
    src = _knn_senders(n, k, 3)
    x = torch.randn(n, 128, device=DEV)
    plan = ops.AggregatePlan.of(src, n, k, 128)
    assert plan is ops.AggregatePlan.of(src, n, k, 128)                  # cached on the tensor
    other = src.clone()
    with pytest.raises(ops.CgnnError, match="another sender list"):
        ops.aggregate(x, other, None, n, k, plan=plan)
    src[0] = src[1]                                                      # in-place change: the cached plan is rebuilt
    plan2 = ops.AggregatePlan.of(src, n, k, 128)
    assert plan2 is not plan
    assert torch.equal(ops.aggregate(x, src, None, n, k, plan=plan2), ops.aggregate(x, src, None, n, k))
    with pytest.raises(ops.CgnnError, match="changed"):
        ops.aggregate(x, src, None, n, k, plan=plan)
    assert ops.AggregatePlan.of(src[: 100 * k].contiguous(), 100, k, 128) is None        # too small to be worth a plan
    assert ops.AggregatePlan.of(src, n, k, 36) is None


def test_dropping_a_sender_list_frees_its_plan_without_the_cyclic_collector():
    """AggregatePlan.of caches the plan on the sender tensor; the plan must not hold the tensor in return (a rollout builds a
    new graph, sender list and 130-MB plan per step: a reference cycle would leave them to Python's cyclic collector)."""
    import gc
    n, k = 20000, 16
    src = _knn_senders(n, k, 9)
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        s2 = src.clone()
        plan = ops.AggregatePlan.of(s2, n, k, 128)
        assert plan is not None and torch.cuda.memory_allocated() >= base + plan.blob.numel()
        del plan, s2
        assert torch.cuda.memory_allocated() == base
    finally:
        gc.enable()


def test_model_forward_is_unchanged_by_the_plan():
    """The whole forward with and without planned aggregation: bit-identical outputs."""
    from cosmology_gnn_simulation_amd import graph_network
    n, k, d = 12000, 16, 128
    snap, meta = synthetic.make_snapshot(n, seed=5), synthetic.make_metadata()
    g = data_utils.preprocess(snap["Coordinates"][:5], snap["InternalEnergy"][:5], meta, None, None, 0.0, k, 0.01, 1.0)
    m = graph_network.EncodeProcessDecode(d, d, 2, 3, 3)
    m.load_state_dict(synthetic.make_state_dict(d, d, 2, 3, 3))
    m = m.to(DEV).eval()
    m.edge_precision, m.node_precision = "bf16", "fp16x2"
    with torch.no_grad():
        a = m(g)
        saved = ops.AggregatePlan.MIN_NODES
        try:
            ops.AggregatePlan.MIN_NODES = 1 << 60                         # no plan
            g2 = data_utils.preprocess(snap["Coordinates"][:5], snap["InternalEnergy"][:5], meta, None, None, 0.0, k, 0.01,
                                       1.0)
            b = m(g2)
        finally:
            ops.AggregatePlan.MIN_NODES = saved
    assert torch.equal(a["acceleration"], b["acceleration"]) and torch.equal(a["temp_rate"], b["temp_rate"])


def test_table_longer_than_the_receivers_part():
    """A spatial shard's table holds ghost rows behind the owned (receiver) rows and its senders point into them: the
    branch-free kernel addresses rows by buffer offsets and must not take the receiver count for the table's length
    (round 3: ghost senders read as zeros until cgnn_aggregate_planned_rows was given the table's row count)."""
    n, ghosts, k, width = 9000, 700, 16, 128
    gen = torch.Generator().manual_seed(11)
    src = _knn_senders(n, k, 5)
    # redirect a tenth of the references to ghost rows
    pick = torch.rand(n * k, generator=gen).to(DEV) < 0.1
    ghost_ids = torch.randint(n, n + ghosts, (n * k,), generator=gen).int().to(DEV)
    src = torch.where(pick, ghost_ids, src).contiguous()
    table = torch.randn(n + ghosts, width, device=DEV)
    plain = ops.aggregate(table, src, None, n, k)
    plan = ops.AggregatePlan(src, n, k)
    got = ops.aggregate(table, src, None, n, k, plan=plan)
    assert got.shape == (n, width) and torch.equal(got, plain)
    dst = torch.arange(n).repeat_interleave(k)
    want = torch.zeros(n, width).index_add_(0, dst, table.cpu()[src.cpu().long()])
    assert float((got.cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max())

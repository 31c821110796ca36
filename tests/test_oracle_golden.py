"""CPU: the oracle (oracle/cpu_ref.py) against the committed golden fixtures, which were produced by the
reference's own source lines (oracle/make_golden.py).  Bit-exact for indices, <=1e-6 for floats."""
import numpy as np
import torch

from conftest import edge_index_from, rel_err
from oracle import cpu_ref

W = 5


def _window(g):
    c, e = torch.from_numpy(g["coords"]), torch.from_numpy(g["energy"])
    return c[:W].clone(), e[:W].clone(), c[W].clone(), e[W].clone()


def test_preprocess_matches_reference(golden):
    g = golden
    c, e, nc, ne = _window(g)
    r = cpu_ref.preprocess(c, e, g["metadata"], nc, ne, 0.0, int(g["k"]), g["metadata"]["dt"], g["metadata"]["box_size"])
    assert torch.equal(r["edge_index"], edge_index_from(g))          # integer work: bit exact
    for key in ("x", "edge_attr", "y_acc", "y_temp_rate", "pos"):
        assert torch.equal(r[key], torch.from_numpy(g[key])), key


def test_graph_layout_facts(golden):
    """receiver-sorted, fixed in-degree, self loop first with zero features."""
    g = golden
    n, k = int(g["n"]), int(g["k"])
    snd = g["senders"].reshape(n, k)
    assert np.array_equal(snd[:, 0], np.arange(n))
    assert np.all(g["edge_attr"].reshape(n, k, 4)[:, 0] == 0)
    d = g["edge_attr"].reshape(n, k, 4)[:, :, 3]
    # not minimum-image: some edges are about a box length long
    assert d.max() > 0.5 * float(g["box"])


def test_forward_matches_reference(golden):
    g = golden
    sd = g["state_dict"]
    with torch.no_grad():
        o = cpu_ref.encode_process_decode(sd, torch.from_numpy(g["x"]), edge_index_from(g),
                                          torch.from_numpy(g["edge_attr"]), int(g["nh"]), int(g["steps"]))
    assert rel_err(o["acceleration"], torch.from_numpy(g["acceleration"])) <= 1e-6
    assert rel_err(o["temp_rate"], torch.from_numpy(g["temp_rate"])) <= 1e-6


def test_edge_stream_is_dead_in_reference_mode(golden_tiny):
    """with PyG's default message the outputs ignore edge_attr entirely."""
    g = golden_tiny
    sd = g["state_dict"]
    ei = edge_index_from(g)
    with torch.no_grad():
        a = cpu_ref.encode_process_decode(sd, torch.from_numpy(g["x"]), ei, torch.from_numpy(g["edge_attr"]), 2, 3)
        b = cpu_ref.encode_process_decode(sd, torch.from_numpy(g["x"]), ei,
                                          torch.randn(ei.shape[1], 4) * 10, 2, 3)
        c = cpu_ref.encode_process_decode(sd, torch.from_numpy(g["x"]), ei, torch.from_numpy(g["edge_attr"]), 2, 3,
                                          message_source="edge")
    assert torch.equal(a["acceleration"], b["acceleration"])
    assert not torch.allclose(a["acceleration"], c["acceleration"])


def test_interaction_block_matches_reference(golden_tiny):
    g = golden_tiny
    with torch.no_grad():
        bx, be = cpu_ref.interaction_network(g["state_dict"], "processor.0", torch.from_numpy(g["enc_x"]),
                                             edge_index_from(g), torch.from_numpy(g["enc_edge"]), 2)
    assert rel_err(bx, torch.from_numpy(g["block0_x"])) <= 1e-6
    assert rel_err(be, torch.from_numpy(g["block0_edge"])) <= 1e-6


def test_one_step_and_momentum_match_reference(golden):
    g = golden
    c, e, nc, ne = _window(g)
    acc, tr = torch.from_numpy(g["acceleration"]), torch.from_numpy(g["temp_rate"])
    s = cpu_ref.one_step(acc, tr, c, e, nc, ne, g["metadata"])
    assert torch.equal(s["new_position"], torch.from_numpy(g["new_position"]))
    assert torch.equal(s["new_temp"], torch.from_numpy(g["new_temp"]))
    assert abs(s["position_mse"] - float(g["position_mse"])) <= 1e-12
    assert abs(s["temperature_mse"] - float(g["temperature_mse"])) <= 1e-12
    m = float(cpu_ref.momentum_conservation_loss(acc, torch.zeros(int(g["n"]), dtype=torch.long), 1,
                                                 g["metadata"]["dt"], 1.0))
    assert abs(m - float(g["momentum"])) <= 1e-6 * abs(float(g["momentum"]))


def test_driver_functions_match_reference(harness):
    """The reference's own validate_one_step / momentum_conservation_loss / rollout outputs (harness.npz)."""
    h = harness
    meta, nh, steps, k = h["metadata"], int(h["nh"]), int(h["steps"]), int(h["k"])
    c, e = torch.from_numpy(h["coords"]), torch.from_numpy(h["energy"])
    W1 = int(h["one_step_window"])
    sd = h["state_dict_one_step"]
    for t, pe, te in zip(h["one_step_tested"], h["one_step_position_errors"], h["one_step_temperature_errors"]):
        s0 = int(t) - W1
        g = cpu_ref.preprocess(c[s0:s0 + W1].clone(), e[s0:s0 + W1].clone(), meta, None, None, 0.0, k, meta["dt"],
                               meta["box_size"])
        with torch.no_grad():
            o = cpu_ref.encode_process_decode(sd, g["x"], g["edge_index"], g["edge_attr"], nh, steps)
        st = cpu_ref.one_step(o["acceleration"], o["temp_rate"], c[s0:s0 + W1], e[s0:s0 + W1], c[s0 + W1], e[s0 + W1], meta)
        assert abs(st["position_mse"] - float(pe)) <= 1e-6 * float(pe)
        assert abs(st["temperature_mse"] - float(te)) <= 1e-6 * float(te)
    assert float(h["one_step_position_error"]) == float(np.mean(h["one_step_position_errors"]))
    # ragged three-graph batch: forward + momentum term
    sizes = [int(v) for v in h["batch_sizes"]]
    batch = torch.cat([torch.full((m,), i, dtype=torch.long) for i, m in enumerate(sizes)])
    acc = torch.from_numpy(h["batch_acceleration"])
    m = float(cpu_ref.momentum_conservation_loss(acc, batch, len(sizes), meta["dt"], float(h["momentum_weight"])))
    assert abs(m - float(h["batch_momentum"])) <= 1e-6 * abs(float(h["batch_momentum"]))
    # rollout: bit-identical trajectory
    W2, R = int(h["rollout_window"]), int(h["rollout_steps"])
    got = cpu_ref.rollout(h["state_dict_rollout"], nh, steps, c, e, meta, W2, 16, W2 + R)
    assert torch.equal(got["Coordinates"], torch.from_numpy(h["rollout_coords"]))
    assert torch.equal(got["InternalEnergy"], torch.from_numpy(h["rollout_energy"]))


def test_knn_restatement_is_exact_vs_bruteforce():
    """The cKDTree-assisted k-NN must equal an exhaustive float32 search (ties by extended index)."""
    gen = torch.Generator().manual_seed(3)
    pos = torch.rand(200, 3, generator=gen)
    ext, _ = cpu_ref.extend_positions(pos, 1.0)
    got = cpu_ref.knn_extended(ext, pos, 8)[1].reshape(200, 8).numpy()
    e, q = ext.numpy(), pos.numpy()
    d2 = cpu_ref._sqdist_f32(e[None, :, :], q[:, None, :])
    want = np.lexsort((np.broadcast_to(np.arange(e.shape[0]), d2.shape), d2), axis=1)[:, :8]
    assert np.array_equal(got, want)

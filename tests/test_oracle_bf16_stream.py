"""CPU, every run: the bf16 edge-stream restatement (oracle/bf16_stream.py) against the f32 oracle on the
reference-generated fixture graphs.  This is what ties the yardstick of the one-launch edge-stream kernels to the
reference: the GPU tests compare the kernels with the restatement row by row (1e-2 of the scale), this file holds the
restatement itself within the bf16 bound of ``cpu_ref`` (3e-2 relative L2, SURVEY F8: a pure-bf16 restatement of the
reference measured 1-2.6e-2) -- and shows that it is NOT the f32 arithmetic (a restatement that forgot to round would pass
the bound trivially)."""
import pytest
import torch

from conftest import edge_index_from, load_golden, rel_l2
from oracle import bf16_stream, cpu_ref


def _node_latents_per_round(sd, x, edge_index, edge_attr, nh, steps):
    """xs[r] = the node latents round r starts with, from cpu_ref's own functions (graph_network.py:166-183)."""
    xl, el = cpu_ref.graph_independent(sd, "encoder", x, edge_attr, nh)
    xs = []
    for i in range(steps):
        xs.append(xl)
        dx, de = cpu_ref.interaction_network(sd, f"processor.{i}", xl, edge_index, el, nh)
        xl, el = xl + dx, el + de
    return xs, el


@pytest.mark.parametrize("name", ["tiny", "tiny_k16_box25", "cfg1"])
@pytest.mark.parametrize("table_dtype", [torch.bfloat16, torch.float16])
def test_bf16_stream_restatement_stays_within_the_bf16_bound_of_the_f32_oracle(name, table_dtype):
    g = load_golden(name)
    sd, nh, steps, latent = g["state_dict"], int(g["nh"]), int(g["steps"]), int(g["latent"])
    x, ea = torch.from_numpy(g["x"]), torch.from_numpy(g["edge_attr"])
    ei = edge_index_from(g)
    with torch.no_grad():
        xs, e_f32 = _node_latents_per_round(sd, x, ei, ea, nh, steps)
        # the loop above IS the oracle's forward: same edge latents as encode_process_decode, bit for bit
        ref = cpu_ref.encode_process_decode(sd, x, ei, ea, nh, steps, return_latents=True)
        assert torch.equal(e_f32, ref["edge_latent"])
        e_bf = bf16_stream.emulate_from_node_latents(sd, xs, ei, ea, latent, nh, table_dtype)
    err = rel_l2(e_bf, e_f32)
    assert err <= 3e-2, err
    assert err >= 1e-4, f"{err}: the restatement does not round like a bf16 path"
    # fp16 tables carry three more significand bits than bf16 ones: never further from the f32 oracle by more than noise
    if table_dtype == torch.float16:
        with torch.no_grad():
            e_b = bf16_stream.emulate_from_node_latents(sd, xs, ei, ea, latent, nh, torch.bfloat16)
        assert err <= 1.2 * rel_l2(e_b, e_f32)


def test_row_subset_form_equals_the_all_edges_form():
    """emulate_edge_stream_rows (what the full-size GPU gates use: sampled rows, the kernel's own P tables) == the
    all-edges form restricted to those rows, given the same tables."""
    g = load_golden("tiny")
    sd, nh, steps, latent = g["state_dict"], int(g["nh"]), int(g["steps"]), int(g["latent"])
    x, ea = torch.from_numpy(g["x"]), torch.from_numpy(g["edge_attr"])
    ei = edge_index_from(g)
    with torch.no_grad():
        xs, _ = _node_latents_per_round(sd, x, ei, ea, nh, steps)
        want = bf16_stream.emulate_from_node_latents(sd, xs, ei, ea, latent, nh, torch.bfloat16)
        D = latent
        ps_all, pd_all = [], []
        for r, xr in enumerate(xs):
            w0, b0 = sd[f"processor.{r}.edge_model.0.0.weight"], sd[f"processor.{r}.edge_model.0.0.bias"]
            ps_all.append(bf16_stream.dot_bf16(xr, w0[:, 0:D]).bfloat16())
            pd_all.append((bf16_stream.dot_bf16(xr, w0[:, D:2 * D]) + b0).bfloat16())
        # logical order -> CGNN_P_BF16_S32 order (the inverse of s32_table_to_logical)
        pos = bf16_stream.s32_position(D, torch.bfloat16)
        def to_s32(p):
            out = torch.empty_like(p)
            out[..., pos] = p
            return out
        stream_inputs = {"src": ei[0].int(), "dst": ei[1].int(), "edge_attr": ea,
                         "ps_all": torch.stack([to_s32(p) for p in ps_all]), "pd_all": torch.stack([to_s32(p) for p in pd_all])}
        rows = torch.arange(0, ei.shape[1], 7)
        got = bf16_stream.emulate_edge_stream_rows(sd, rows, stream_inputs, latent, nh, steps)
    assert torch.equal(got, want[rows])


@pytest.mark.parametrize("name", ["tiny", "tiny_k16_box25", "cfg1"])
def test_folded_parameters_are_the_same_model(name):
    """fold_state_dict (CGNN_STREAM_FOLDED, include/cgnn.h: centred output Linears, LayerNorm shifts carried forward into the
    next round's first-Linear bias) leaves the final edge latents AND the node outputs of the f32 oracle unchanged -- the
    intermediate edge latents differ by B_r -- and the bf16 restatement of the folded model stays within the same bound."""
    g = load_golden(name)
    sd, nh, steps, latent = g["state_dict"], int(g["nh"]), int(g["steps"]), int(g["latent"])
    x, ea = torch.from_numpy(g["x"]), torch.from_numpy(g["edge_attr"])
    ei = edge_index_from(g)
    fsd = bf16_stream.fold_state_dict(sd, latent, nh, steps)
    changed = {k for k in sd if not torch.equal(sd[k], fsd[k])}
    assert changed and all(".edge_model." in k for k in changed), changed
    with torch.no_grad():
        ref = cpu_ref.encode_process_decode(sd, x, ei, ea, nh, steps, return_latents=True)
        fold = cpu_ref.encode_process_decode(fsd, x, ei, ea, nh, steps, return_latents=True)
        assert rel_l2(fold["edge_latent"], ref["edge_latent"]) <= 2e-6
        assert torch.equal(fold["acceleration"], ref["acceleration"]) and torch.equal(fold["x_latent"], ref["x_latent"])
        # the folded model's LayerNorm inputs have no mean, and only its last round shifts
        xs, _ = _node_latents_per_round(sd, x, ei, ea, nh, steps)
        e_bf = bf16_stream.emulate_from_node_latents(fsd, xs, ei, ea, latent, nh, torch.float16, folded=True)
        # LayerNorm without a mean on centred weights: the plain LayerNorm up to what bf16 leaves of the mean (~1e-4 sigma),
        # amplified by the operand-rounding flips of the following rounds
        e_ln = bf16_stream.emulate_from_node_latents(fsd, xs, ei, ea, latent, nh, torch.float16)
        assert rel_l2(e_bf, e_ln) <= 5e-3
    for r in range(steps):
        w = fsd[f"processor.{r}.edge_model.0.{2 * nh}.weight"]
        assert float(w.sum(dim=0).abs().max()) <= 1e-5 * float(w.abs().max()) * w.shape[0]
        beta = fsd[f"processor.{r}.edge_model.1.bias"]
        assert (r + 1 == steps) or not beta.any()
    err = rel_l2(e_bf, ref["edge_latent"])
    assert 1e-4 <= err <= 3e-2, err


@pytest.mark.parametrize("name", ["tiny", "cfg1"])
def test_the_engines_fold_is_the_oracles_fold(name):
    """graph_network.fold_edge_stream (what the model packs for CGNN_STREAM_FOLDED) and oracle fold_state_dict (the yardstick's
    restatement of it) are written independently and must produce the same parameters, bit for bit: the GPU gates compare
    the kernel on the first with the emulation on the second."""
    from cosmology_gnn_simulation_amd import graph_network
    g = load_golden(name)
    sd, nh, steps, latent = g["state_dict"], int(g["nh"]), int(g["steps"]), int(g["latent"])
    hidden = sd["processor.0.edge_model.0.0.weight"].shape[0]
    out_dim = sd[f"decoder_acc.{2 * nh}.weight"].shape[0]
    m = graph_network.EncodeProcessDecode(latent, hidden, nh, steps, out_dim)
    m._materialize_all(int(g["x"].shape[1]), int(g["edge_attr"].shape[1]))
    m.load_state_dict(sd)
    folded = graph_network.fold_edge_stream([net.edge_model for net in m.processor], latent)
    fsd = bf16_stream.fold_state_dict(sd, latent, nh, steps)
    for r, (wb, (gamma, beta)) in enumerate(folded):
        pre = f"processor.{r}.edge_model"
        for i, (w, b) in enumerate(wb):
            assert torch.equal(w.detach(), fsd[f"{pre}.0.{2 * i}.weight"]), (r, i)
            assert torch.equal(b.detach(), fsd[f"{pre}.0.{2 * i}.bias"]), (r, i)
        assert torch.equal(gamma.detach(), fsd[f"{pre}.1.weight"]) and torch.equal(beta.detach(), fsd[f"{pre}.1.bias"]), r
    w, b = graph_network._centred_output(graph_network._split_mlp(m.encoder.edge_model)[0][-1])
    assert torch.equal(w, fsd[f"encoder.edge_model.0.{2 * nh}.weight"]) and torch.equal(b, fsd[f"encoder.edge_model.0.{2 * nh}.bias"])

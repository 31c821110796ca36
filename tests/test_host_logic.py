"""CPU: host-side logic of the product (no kernel launches)."""
import os
import re

import pytest
import torch

from conftest import ROOT
from cosmology_gnn_simulation_amd import _lib, graph_network, synthetic
from cosmology_gnn_simulation_amd.graph import Batch, Data


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()                                   # dlopen works without a GPU
    header = open(os.path.join(ROOT, "include", "cgnn.h")).read()
    declared = set(re.findall(r"\b(cgnn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cgnn_version() == 100
    assert lib.cgnn_arch() == b"gfx950"
    assert lib.cgnn_packed_linear_bytes(128, 128, _lib.BF16) == 128 * 128 * 2
    assert lib.cgnn_packed_linear_bytes(3, 17, _lib.F32) == 32 * 32 * 4      # padded to whole 32x32 tiles


def test_struct_layout_matches_header():
    import ctypes as C
    assert C.sizeof(_lib.Linear) == 24
    assert C.sizeof(_lib.Mlp) == 8 + 7 * 24 + 16
    assert C.sizeof(_lib.MlpBwdBuffers) == (6 + 6 + 2) * 8


def test_state_dict_contract_and_load_before_forward():
    m = graph_network.EncodeProcessDecode(128, 128, 2, 10, 3)
    keys = list(m.state_dict().keys())
    assert len(keys) == 188
    shapes = synthetic.state_dict_shapes(128, 128, 2, 10, 3)
    assert keys == list(shapes.keys())
    sd = synthetic.make_state_dict(128, 128, 2, 10, 3)
    m.load_state_dict(sd)                                 # before any forward (one_step_test.py:21)
    assert tuple(m.state_dict()["processor.3.edge_model.0.0.weight"].shape) == (128, 384)
    assert sum(v.numel() for v in sd.values()) == 1623428


def test_no_cpu_fallback():
    m = graph_network.EncodeProcessDecode(32, 32, 2, 1, 3)
    m.load_state_dict(synthetic.make_state_dict(32, 32, 2, 1, 3))
    g = Data(x=torch.zeros(4, 17), edge_index=torch.zeros(2, 8, dtype=torch.long), edge_attr=torch.zeros(8, 4))
    with torch.no_grad(), pytest.raises(_lib.CgnnError):
        m(g)


def test_training_paths_without_backward_kernels_are_refused_loudly():
    g = Data(x=torch.zeros(4, 17), edge_index=torch.zeros(2, 8, dtype=torch.long), edge_attr=torch.zeros(8, 4))
    m = graph_network.EncodeProcessDecode(32, 32, 2, 1, 3)
    m.message_source = "edge"                  # the engine's extension has no backward
    with pytest.raises(NotImplementedError):
        m(g)
    with pytest.raises(NotImplementedError):   # stand-alone sub-modules are inference-only
        m.encoder(g)
    m.message_source = "x_j"
    with pytest.raises(_lib.CgnnError):        # training exists, but only on the device: no CPU path
        m(g)


def test_edge_attr_none_raises_value_error():
    net = graph_network.InteractionNetwork(torch.nn.Identity(), torch.nn.Identity())
    with pytest.raises(ValueError):
        net(Data(x=torch.zeros(2, 4), edge_index=torch.zeros(2, 2, dtype=torch.long), edge_attr=None))


def test_data_and_batch_surface():
    d = Data(x=torch.zeros(3, 2), edge_index=torch.tensor([[0, 1], [1, 2]]), edge_attr=torch.zeros(2, 4))
    assert not hasattr(d, "globals")
    d2 = Data(x=torch.ones(2, 2), edge_index=torch.tensor([[0], [1]]), edge_attr=torch.zeros(1, 4))
    b = Batch.from_data_list([d, d2])
    assert b.num_graphs == 2 and b.x.shape == (5, 2)
    assert b.edge_index.tolist() == [[0, 1, 3], [1, 2, 4]]
    assert b.batch.tolist() == [0, 0, 0, 1, 1]


def test_synthetic_generators_are_deterministic():
    a = synthetic.make_snapshot(100, seed=5)
    b = synthetic.make_snapshot(100, seed=5)
    assert torch.equal(a["Coordinates"], b["Coordinates"]) and a["Coordinates"].shape == (6, 100, 3)
    assert float(a["Coordinates"].min()) >= 0.0 and float(a["Coordinates"].max()) < 1.0
    assert set(synthetic.make_metadata()) == set(synthetic.METADATA_KEYS)


def test_snapshot_io_and_metadata(tmp_path):
    from cosmology_gnn_simulation_amd import snapshot_io
    snap = synthetic.make_snapshot(50, seed=2)
    path = str(tmp_path / "s.npz")
    snapshot_io.write_snapshot(path, snap)
    back = snapshot_io.read_snapshot(path)
    assert torch.equal(back["Coordinates"], snap["Coordinates"]) and float(back["BoxSize"]) == 1.0
    meta = snapshot_io.generate_metadata(back, str(tmp_path / "m.json"))
    assert set(meta) == set(synthetic.METADATA_KEYS)                   # the 10 keys of generate_metadata.py:32-43
    assert isinstance(meta["temp_mean"], list) and isinstance(meta["vel_std"], float)
    e = back["InternalEnergy"].double().numpy()
    assert abs(meta["temp_std"][0] - e.std(axis=(0, 1))[0]) < 1e-12
    with pytest.raises(ValueError):
        snapshot_io.read_snapshot(str(tmp_path / "x.bin"))


def test_reference_module_names_resolve_to_the_engine(tmp_path):
    """``from graph_network import EncodeProcessDecode`` / ``from data_utils import preprocess`` (reference
    train.py:15-16, one_step_test.py:9-10) with ``compat/`` on PYTHONPATH; ``torch_geometric`` names with
    ``compat/pyg_free`` as well.  Run in a child interpreter so this process's modules stay untouched."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import graph_network, data_utils, inspect\n"
        "import cosmology_gnn_simulation_amd.graph_network as eng, cosmology_gnn_simulation_amd.data_utils as du\n"
        "assert graph_network.EncodeProcessDecode is eng.EncodeProcessDecode\n"
        "assert graph_network.InteractionNetwork is eng.InteractionNetwork and graph_network.build_mlp is eng.build_mlp\n"
        "assert data_utils.preprocess is du.preprocess and data_utils.extend_positions_torch is du.extend_positions_torch\n"
        "p = list(inspect.signature(data_utils.preprocess).parameters)[:9]\n"
        "assert p == ['position_seq', 'temperature_seq', 'metadata', 'target_position', 'target_temperature',\n"
        "             'noise_std', 'num_neighbors', 'dt', 'box_size'], p\n"
        "import torch_geometric as pyg\n"
        "from torch_geometric.loader import DataLoader\n"
        "from cosmology_gnn_simulation_amd.graph import Batch\n"
        "assert pyg.data.Batch is Batch\n"
        "print('ok')\n")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(root, "compat"), os.path.join(root, "compat", "pyg_free")])
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_generate_metadata_matches_reference_driver():
    """The ten statistics against the reference's own generate_metadata.generate_metadata (generate_metadata.py:6-48),
    run behind the h5py stand-in by oracle/make_golden.py (tests/golden/harness.npz): same keys, same list-vs-scalar
    shapes, same values (the reference reduces in float32, this restatement in float64)."""
    import numpy as np
    from conftest import load_harness
    from cosmology_gnn_simulation_amd import snapshot_io
    h = load_harness()
    snap = {"Coordinates": torch.from_numpy(h["coords"]), "InternalEnergy": torch.from_numpy(h["energy"]),
            "Velocities": torch.from_numpy(h["velocities"]), "HydroAcceleration": torch.from_numpy(h["hydro_acceleration"]),
            "BoxSize": torch.tensor(h["metadata"]["box_size"]), "TimeStep": torch.tensor(h["metadata"]["dt"])}
    got = snapshot_io.generate_metadata(snap)
    want = h["generated_metadata"]
    assert sorted(got) == sorted(want)
    for key, w in want.items():
        g = np.asarray(got[key], dtype=np.float64)
        assert g.shape == w.shape, key                       # temperature statistics are length-1 lists, the rest scalars
        assert np.allclose(g, w, rtol=1e-5, atol=1e-7), (key, g, w)


def test_kernels_with_hand_issued_loads_use_no_scratch_and_pad_their_own_hazards(tmp_path):
    """The ring kernels and the one-launch edge stream request rows with inline-asm loads and hand the registers over
    behind a counted `s_waitcnt` of their own.  The compiler does not know that those registers are still in flight: if it
    spilled one between the request and the wait it would store a stale value (and its own reloads wait vmcnt(0),
    draining the prefetches).  So these kernels must compile without scratch: checked here from hipcc's resource remarks.
    Nor can it pad hazards inside an asm block: a vector-memory instruction there that reads an SGPR a VALU instruction
    wrote fewer than five wait states earlier (v_readlane of a spilled base pointer, v_readfirstlane) goes out with the
    old address -- scripts/dev/scan_asm_hazards.py reads the listing for that."""
    import concurrent.futures
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "cosmology_gnn_simulation_amd", "csrc")
    sys.path.insert(0, os.path.join(ROOT, "scripts", "dev"))
    try:
        import scan_asm_hazards
    finally:
        sys.path.pop(0)

    def listing(name, extra=()):
        out = str(tmp_path / (name + ".s"))
        cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
               "-Wno-unused-function", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", *extra, "-S",
               os.path.join(csrc, name), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return name, r.stderr, out

    # (file, extra flags, which of its kernels must be free of scratch: a regular expression on the mangled name, "" = all)
    jobs = [("node_block_f2.hip", (), ""), ("edge_block_f2.hip", (), ""), ("edge_block_ring256.hip", (), ""),
            ("edge_stream32.hip", ("-mllvm", "-amdgpu-mfma-vgpr-form=1"), ""),
            # round 1's 16-row kernels: rows requested by hand-issued global_load_dwordx4 (n16.hpp) behind manual waits
            ("edge_stream.hip", (), ""), ("node_block_n16.hip", (), ""), ("edge_block.hip", (), "n16_kernel"),
            # the bench's edge stream (two waves per SIMD): its encoder form (ENC = true: "ELb1E") is the one that runs in the
            # model; a spill there reloads behind vmcnt(0) and drains the ring's prefetches
            # (template arguments DT, NH, ENC, LAG, PF16: "ILi4ELi<NH>ELb1E..." = ENC true)
            ("edge_stream32w.hip", ("-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"), r"kernelILi4ELi\dELb1E"),
            # planned aggregation: branch-free buffer loads / stores (round 3).  Its 16-byte buffer stores are what the second
            # scan below is about
            ("aggregate_plan.hip", (), "planned_kernelILi16E")]
    with concurrent.futures.ThreadPoolExecutor(4) as pool:
        for (name, text, path), (_, _, must) in zip(pool.map(lambda j: listing(*j[:2]), jobs), jobs):
            sizes = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", text)]
            spills = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", text)]
            kernels = re.findall(r"Function Name: (\S+)", text)
            assert sizes and len(sizes) == len(spills) == len(kernels), name
            bad = [k for k, s_, v in zip(kernels, sizes, spills) if re.search(must, k) and (s_ != 0 or v != 0)]
            assert not bad, (name, bad)
            assert any(re.search(must, k) for k in kernels), name
            assert scan_asm_hazards.scan(path) == [], name
            # a 16-byte buffer store with a REGISTER in its scalar-offset field, data registers overwritten by the very next
            # vector instruction: hipcc pads that only for a constant soffset, gfx950 needs the wait state either way (wrong
            # sums in round 3's first branch-free aggregation kernel; scripts/dev/scan_asm_hazards.py)
            assert scan_asm_hazards.scan_store_hazard(path) == [], name


def test_store_hazard_scanner_sees_an_unpadded_wide_buffer_store(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts", "dev"))
    try:
        import scan_asm_hazards
    finally:
        sys.path.pop(0)
    body = ("k:\n\tbuffer_store_dwordx4 v[12:15], v135, s[8:11], %s offen\n%s\tv_pk_add_f32 v[12:13], v[148:149], v[152:153]\n"
            "\ts_endpgm\n")
    cases = {"reg": ("s19", ""), "const": ("0", ""), "padded": ("s19", "\ts_nop 0\n"), "other": ("s19", "\tv_mov_b32_e32 v1, v2\n")}
    found = {}
    for name, (soff, between) in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(body % (soff, between))
        found[name] = len(scan_asm_hazards.scan_store_hazard(str(f)))
    assert found == {"reg": 1, "const": 0, "padded": 0, "other": 0}


def test_asm_hazard_scanner_sees_a_short_gap(tmp_path):
    """The scanner itself: a base pointer read back from a VGPR lane two instructions before an asm load is a finding,
    the same listing with the five wait states is not."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts", "dev"))
    try:
        import scan_asm_hazards
    finally:
        sys.path.pop(0)
    body = ("k:\n\tv_readlane_b32 s4, v255, 2\n\tv_readlane_b32 s5, v255, 3\n\tv_mov_b32_e32 v1, v2\n\t;;#ASMSTART\n"
            "\ts_mov_b32 m0, s6\n\ts_nop %d\n\tglobal_load_lds_dwordx4 v3, s[4:5]\n\t;;#ASMEND\n\ts_endpgm\n")
    short, padded = tmp_path / "short.s", tmp_path / "padded.s"
    short.write_text(body % 0)
    padded.write_text(body % 3)
    assert len(scan_asm_hazards.scan(str(short))) == 2          # s4 after 4 states, s5 after 3
    assert scan_asm_hazards.scan(str(padded)) == []


def test_committed_pmc_traffic_belongs_to_the_committed_kernels():
    """profiles/traffic.json is keyed by a hash of each kernel's sources (bench.kernel_source_sha16); bench.py drops an
    entry whose kernel changed since the PMC pass.  At a commit the entries of the headline configuration should be live:
    a kernel edit after the last profile round shows up here, not as a silent `traffic: null` in the bench line."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)
    db = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    stale = [k for k, v in db.items() if isinstance(v, dict) and bench.kernel_source_sha16(v["source"]) != v["source_sha16"]]
    assert not stale, f"re-run scripts/profile_round.sh: PMC traffic entries measured on older kernel sources: {stale}"
    for key in ("edge_stream+enc:1000000:16:128:10", "node_block:1000000:128", "aggregate:1000000:16:128"):
        assert bench._traffic(key, "") and bench._traffic(key, "") > 0, key


def test_integration_notes_name_every_abi_entry():
    """INTEGRATION.md's table is where a maintainer of the reference looks up what each C entry replaces: every function
    include/cgnn.h declares must appear there."""
    header = open(os.path.join(ROOT, "include", "cgnn.h")).read()
    declared = set(re.findall(r"\b(cgnn_[a-z0-9_]+)\s*\(", header))
    notes = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert len(declared) > 30
    assert [d for d in sorted(declared) if d not in notes] == []


def test_bench_presets_follow_baseline_json(monkeypatch):
    """bench.py's --config presets are BASELINE.json's configs (particles, neighbours, latent, rounds), seeds 1234 + the
    configuration number; N > 1 defaults to weak scaling WITH the strong-scaling leg of the headline box
    (BASELINE.json: 'at 1M particles ... 1/2/4/8 GPU'), cfg4 / cfg5 to strong scaling of their own totals."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))

    def parse(*argv):
        monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
        return bench.parse_args()

    a = parse()
    assert (a.gpus, a.particles, a.neighbors, a.latent, a.mp_steps, a.edge_precision, a.seed) == (1, 1_000_000, 16, 128, 10, "bf16", 1237)
    want = {"cfg1": (4096, 8, 64, 5), "cfg2": (262144, 16, 128, 10), "cfg3": (1_000_000, 16, 128, 10),
            "cfg4": (4_000_000, 16, 128, 10), "cfg5": (1_000_000, 32, 256, 15)}
    mult = {"k": 1024, "M": 1_000_000}      # "4k" = 4,096 and "256k" = 262,144, "1M" = 1,000,000
    for i, (name, shape) in enumerate(want.items()):
        a = parse("--config", name)
        assert (a.particles, a.neighbors, a.latent, a.mp_steps) == shape, name
        assert a.seed == 1234 + i + 1, name
        m = re.match(r"(\d+)([kM]) particles, k=(\d+), latent=(\d+), (\d+) MP steps", base["configs"][i])
        assert m, base["configs"][i]
        assert (int(m.group(1)) * mult[m.group(2)], int(m.group(3)), int(m.group(4)), int(m.group(5))) == shape, name
    assert parse("--gpus", "8").scaling == "weak" and not parse("--gpus", "8").no_strong_leg
    assert parse("--gpus", "8", "--config", "cfg4").scaling == "strong"
    assert parse("--gpus", "8", "--config", "cfg5").scaling == "strong"
    assert parse("--gpus", "8", "--scaling=strong").scaling == "strong"

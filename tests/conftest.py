import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library is a build artefact (git-ignored): build it once if a fresh checkout has none (hipcc
    cross-compiles gfx950 without a GPU; about a minute)."""
    lib = os.path.join(ROOT, "cosmology_gnn_simulation_amd", "libcgnn_hip.so")
    if not os.path.isfile(lib):
        import __graft_entry__
        __graft_entry__.build()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    g = {k: z[k] for k in z.files}
    sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w:")}
    g = {k: v for k, v in g.items() if not k.startswith("w:")}
    g["state_dict"] = sd
    g["metadata"] = dict(vel_mean=0.0, vel_std=1.0, acc_mean=0.0, acc_std=1.0, temp_mean=1.0, temp_std=0.5,
                         temp_rate_mean=0.0, temp_rate_std=2.0, dt=float(g["dt"]), box_size=float(g["box"]))
    return g


def load_harness():
    """tests/golden/harness.npz: outputs of the reference's own driver functions (validate_one_step,
    momentum_conservation_loss on a ragged batch, rollout), see oracle/make_golden.py::run_harness."""
    z = np.load(os.path.join(GOLDEN_DIR, "harness.npz"))
    g = {k: z[k] for k in z.files if not k.startswith(("w1:", "w2:", "meta:", "genmeta:"))}
    g["generated_metadata"] = {k[8:]: z[k] for k in z.files if k.startswith("genmeta:")}
    g["state_dict_one_step"] = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w1:")}
    g["state_dict_rollout"] = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w2:")}
    g["metadata"] = {k[5:]: float(z[k]) for k in z.files if k.startswith("meta:")}
    return g


@pytest.fixture(scope="session")
def harness():
    return load_harness()


@pytest.fixture(scope="session", params=["tiny", "tiny_k16_box25", "cfg1"])
def golden(request):
    return load_golden(request.param)


@pytest.fixture(scope="session")
def golden_tiny():
    return load_golden("tiny")


def edge_index_from(g):
    n, k = int(g["n"]), int(g["k"])
    snd = torch.from_numpy(g["senders"].astype(np.int64))
    rcv = torch.arange(n, dtype=torch.int64).repeat_interleave(k)
    return torch.stack([snd, rcv], dim=0)


def rel_err(a, b):
    """The larger of the two norms SURVEY 8(d) names for the f32 gates: max-abs / max-abs and relative L2."""
    a, b = a.double(), b.double()
    linf = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    l2 = float((a - b).norm() / b.norm().clamp_min(1e-30))
    return max(linf, l2)


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

"""Developer aid: error patterns of cgnn_edge_stream_run against the bf16 emulation (by feature, by edge)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_edge_stream32 as T  # noqa: E402

torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)


def report(name, prob):
    got = T._run(*prob)
    want = T._emulate(*prob)
    err = (got - want).abs()
    d = got.shape[1]
    print(f"== {name}: max err {float(err.max()):.4f} scale {float(want.abs().max()):.3f} relL2 {float((got-want).norm()/want.norm()):.2e}")
    pf = err.max(dim=0).values
    print(" per-feature max err:", [round(float(v), 2) for v in pf[:min(d, 128)]])
    pe = err.max(dim=1).values
    print(" per-edge max err (first 70):", [round(float(v), 2) for v in pe[:70]])
    return got, want


def variant(prob, **kw):
    src, dst, ps, pd, mlps, enc, attr, e0 = prob
    mlps = [([(w.clone(), b.clone()) for w, b in lin], (ln[0].clone(), ln[1].clone())) for lin, ln in mlps]
    if kw.get("zero_p"):
        ps, pd = torch.zeros_like(ps), torch.zeros_like(pd)
    if kw.get("zero_we"):
        for lin, ln in mlps:
            lin[0][0].zero_()
    if kw.get("unit_ln"):
        for lin, ln in mlps:
            ln[0].fill_(1.0)
            ln[1].zero_()
    if kw.get("zero_bias"):
        for lin, ln in mlps:
            for w, b in lin:
                b.zero_()
    return src, dst, ps, pd, mlps, enc, attr, e0


base = T._problem(1, 8, 8, 128, 1, 1, False, 0)
report("full nh=1 L=1", base)
report("zero P", variant(base, zero_p=True))
report("zero We", variant(base, zero_we=True))
report("zero P, unit LN, zero bias", variant(base, zero_p=True, unit_ln=True, zero_bias=True))
report("zero We, unit LN, zero bias", variant(base, zero_we=True, unit_ln=True, zero_bias=True))
b2 = T._problem(2, 8, 8, 128, 2, 1, False, 0)
report("nh=2 L=1", b2)
b3 = T._problem(3, 8, 8, 64, 1, 1, False, 0)
report("latent 64 nh=1 L=1", b3)

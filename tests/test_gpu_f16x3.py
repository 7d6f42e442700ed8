"""Precision mode "f16x3" (GEMMs on the fp16 x3 split MFMA) end to end against the same goldens as the exact-fp32
mode.  Bar: 2e-5 relative L2 on every stage and hidden state (measured ~3e-6; north_star allows 1e-3; plain
fp16 would be 1.6e-3)."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import la, model, rel_l2

TOL = 2e-5


def run(lengths, mask=True, hidden=False, taps=False):
    m, sd = model()
    enc = m.speecht5.encoder
    x, msk = la.synth.batch(lengths)
    st = {} if taps else None
    enc.precision = "f16x3"
    try:
        out = enc(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda() if mask else None,
                  output_hidden_states=hidden, stage_taps=st)
        torch.cuda.synchronize()
    finally:
        enc.precision = "f32"
    return out, st, (x, msk, sd)


def test_g2_ragged_every_stage():
    g = golden("g2_5s_3s.npz")
    rows = torch.from_numpy(g["rows"])
    out, st, _ = run(g["lengths"], hidden=True, taps=True)
    for name in ("conv_stack", "feature_projection", "prenet"):
        assert rel_l2(st[name][:, rows], g[name]) < TOL, name
    for i, h in enumerate(out.hidden_states):
        assert rel_l2(h[:, rows], g["hidden_states"][i]) < TOL, i


def test_g3_headline_shape_and_g5_long():
    g = golden("g3_30s_x2.npz")
    out, _, _ = run(g["lengths"], hidden=True)
    for i in (0, 6, 12):
        assert rel_l2(out.hidden_states[i][:, torch.from_numpy(g["rows"])], g["hidden_states"][i]) < TOL, i
    g = golden("g5_T4096.npz")
    out, _, _ = run(g["lengths"], mask=False)
    assert rel_l2(out.last_hidden_state[:, torch.from_numpy(g["rows"])], g["hidden_states"][12]) < TOL


def test_against_fp64_oracle_and_fp32_mode(oracle):
    lengths = [30000, 17000, 400]
    out, _, (x, msk, sd) = run(lengths)
    ref64 = oracle.encode(x, msk, sd, dtype=torch.float64)
    e16 = rel_l2(out.last_hidden_state, ref64)
    m, _ = model()
    y32 = m.speecht5.encoder(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda()).last_hidden_state
    e32 = rel_l2(y32, ref64)
    print(f"rel L2 vs fp64 oracle: f16x3 {e16:.2e}, f32 {e32:.2e}")
    assert e32 < 3e-6 and e16 < 1e-5
    assert not torch.equal(y32, out.last_hidden_state)  # the mode really switched


def test_deterministic():
    a, _, _ = run([32000, 32000])
    b, _, _ = run([32000, 32000])
    assert torch.equal(a.last_hidden_state, b.last_hidden_state)

"""Opt-in precision mode "f16x2" (VERDICT r1, next-round 6b): the weights of every projection / conv GEMM rounded to fp16 (after
their per-tensor power-of-two scale), activations still carried as hi + lo, the attention products and the relative-position
table still three-term.  A third fewer matrix instructions in the GEMMs; NO LONGER fp32 class: the bar here is north_star's
1e-3 relative L2 against the HuggingFace goldens and the fp64 oracle, and the achieved figures are printed.  The default stays
"f16x3" (asserted below)."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import la, model, rel_l2

BAR = 1e-3


def run(lengths, mask=True, hidden=False, precision="f16x2"):
    m, sd = model(precision=precision)
    x, msk = la.synth.batch(lengths)
    out = m.speecht5.encoder(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda() if mask else None,
                             output_hidden_states=hidden)
    torch.cuda.synchronize()
    assert not m.speecht5.encoder.last_range_fallback
    return out, (x, msk, sd)


def test_default_precision_is_still_fp32_class():
    assert la.SpeechT5ForSpeechToTextMI355X(1).speecht5.encoder.precision == "f16x3"
    with pytest.raises(ValueError):
        la.SpeechT5ForSpeechToTextMI355X(1, precision="fp16")


def test_goldens_within_the_1e3_bar():
    worst = 0.0
    g = golden("g1_1s.npz")
    out, _ = run(g["lengths"])
    worst = max(worst, rel_l2(out.last_hidden_state, g["last_hidden_state"]))
    g = golden("g2_5s_3s.npz")
    rows = torch.from_numpy(g["rows"])
    out, _ = run(g["lengths"], hidden=True)
    per_layer = [rel_l2(h[:, rows], g["hidden_states"][i]) for i, h in enumerate(out.hidden_states)]
    worst = max(worst, max(per_layer))
    g = golden("g3_30s_x2.npz")
    rows = torch.from_numpy(g["rows"])
    out, _ = run(g["lengths"], hidden=True)
    per_layer3 = [rel_l2(out.hidden_states[i][:, rows], g["hidden_states"][i]) for i in range(13)]
    worst = max(worst, max(per_layer3))
    g = golden("g5_T4096.npz")
    out, _ = run(g["lengths"], mask=False, hidden=True)
    e5 = rel_l2(out.hidden_states[12][:, torch.from_numpy(g["rows"])], g["hidden_states"][12])
    worst = max(worst, e5)
    print(f"f16x2 vs HF goldens: 5s+3s per layer {['%.1e' % e for e in per_layer]}, 30s x2 last layer {per_layer3[12]:.2e}, T=4096 {e5:.2e}; worst {worst:.2e}")
    assert worst < BAR


def test_against_fp64_oracle_and_the_default_mode(oracle):
    lengths = [30000, 17000, 400]
    out, (x, msk, sd) = run(lengths)
    ref64 = oracle.encode(x, msk, sd, dtype=torch.float64)
    e2 = rel_l2(out.last_hidden_state, ref64)
    y3, _ = run(lengths, precision="f16x3")
    e3 = rel_l2(y3.last_hidden_state, ref64)
    print(f"rel L2 vs fp64 oracle: f16x2 {e2:.2e}, f16x3 {e3:.2e}")
    assert e3 < 1e-5 < e2 < BAR  # really another arithmetic, inside the bar
    again, _ = run(lengths)
    assert torch.equal(again.last_hidden_state, out.last_hidden_state)  # deterministic

"""BASELINE.json configs[3] without its chunking: ONE 60-minute recording as a single clip, T = 179 999 frames (the reference
windows such audio to 10 minutes because HuggingFace cannot hold the attention of anything longer; this library can, so it must
either be right at that size or refuse it).  57.6 M samples -> 11.52 M conv0 frames x 512 channels = 5.9e9 elements per fp16 plane:
the first size at which an element index leaves 32 bits, six times the sequence length of configs[2].  Parity is established as in
test_gpu_long.py -- each stage's input comes from the GPU run, its output rows are recomputed by the oracle in fp64 with K / V over
all 179 999 frames, at both ends and the middle of every buffer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import la, model, rel_l2

L60 = 57_600_000
T60 = 179_999
ROWS = [0, 1, 63, 64, 159, 160, 161, 90000, 179000, 179838, 179839, 179997, 179998]


@pytest.fixture(scope="module")
def hour_run():
    m, sd = model()
    assert la.synth.conv_out_length(L60) == T60
    x = torch.from_numpy(la.synth.clip(91, L60))[None]
    st = {}
    enc = m.speecht5.encoder
    out = enc(input_values=x.cuda(), output_hidden_states=True, stage_taps=st)
    torch.cuda.synchronize()
    assert not enc.last_range_fallback
    yield x[0], out, st, sd
    del out, st
    torch.cuda.empty_cache()


def test_hour_shapes_and_finiteness(hour_run):
    _, out, st, _ = hour_run
    assert tuple(out.last_hidden_state.shape) == (1, T60, 768) and len(out.hidden_states) == 13
    assert bool(torch.isfinite(out.last_hidden_state).all())
    assert st["frames"].cpu().tolist() == [T60]
    # the far end of the output is not a copy of the near end (an index that wrapped at 2^32 elements would make it one)
    assert not torch.equal(out.last_hidden_state[0, -64:], out.last_hidden_state[0, :64])


def test_hour_conv_stack_windows(hour_run, oracle):
    x, _, st, sd = hour_run
    cache = {}
    # 131 071 / 131 072: conv0 frames 8 388 544.. -- where the element index of its [frames, 512] planes crosses 2^32
    for lo, hi in ((0, 6), (131070, 131075), (T60 - 5, T60)):
        ref = oracle.feature_encoder_window(x, sd, lo, hi, stats_cache=cache)
        assert rel_l2(st["conv_stack"][0, lo:hi], ref) < 1e-5, (lo, hi)


def test_hour_pos_conv_rows(hour_run, oracle):
    _, _, st, sd = hour_run
    ref = oracle.pos_conv_rows(st["feature_projection"][0].cpu(), ROWS, T60, sd)
    assert rel_l2(st["prenet"][0, ROWS], ref) < 1e-5


@pytest.mark.parametrize("layer", [0, 11])
def test_hour_encoder_layer_rows(hour_run, oracle, layer):
    _, out, _, sd = hour_run
    pe_k = torch.from_numpy(sd["wrapped_encoder.embed_positions.pe_k.weight"])
    xin = out.hidden_states[layer][0].cpu()
    ref = oracle.encoder_layer_rows(xin, ROWS, None, sd, f"wrapped_encoder.layers.{layer}.", pe_k)
    assert rel_l2(out.hidden_states[layer + 1][0, ROWS], ref) < 1e-5

"""Precision mode "f16x3" outside the friendly numeric range of the synthetic weights.

Every GEMM operand of that mode is a pair of fp16 planes: hi = fp16(x) overflows at 65504 and lo = fp16(x - hi) falls into
subnormals for small x.  The library's policy (include/loco_asr.h, "numeric range of precision mode f16x3"):

  * weights are stored as W * 2^k per tensor (max|W| -> [2^13, 2^14)), undone exactly in the epilogue: their level is free;
  * activation planes whose range follows from the weights (LayerNorm / GroupNorm outputs) are checked at load time, the
    unbounded ones (GELU outputs of the conv stack and of the feed-forward, q|k|v, the feature projection) report max|x|
    per stage from the GEMM epilogue that writes them; a forward whose stages all lie in [2^-6, 65504) is guaranteed to fp32
    class, anything else is detected (LOCO_E_RANGE) and -- default policy -- the batch is run again on the library's
    exact-fp32 MFMA kernels.  Never a CPU / PyTorch fallback, never NaNs handed to the caller.

All comparisons are against the fp64 evaluation of the CPU oracle on the same (modified) weights; bars are stated per test
(north_star: 1e-3)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import la, rel_l2
    _libmod = __import__("importlib").import_module("loco-asr_amd._lib")

LAYERS = 2
LENGTHS = [24000, 16000, 9000]


def build(mod):
    """2-layer encoder with `mod(sd)` applied to the synthetic state dict (numpy, HF names with sub-module prefixes)."""
    sd = la.synth.encoder_state_dict(0, LAYERS)
    mod(sd)
    pre, enc = la.synth.split_state_dict(sd)
    m = la.SpeechT5ForSpeechToTextMI355X(LAYERS)
    m.speecht5.encoder.wrapped_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in enc.items()})
    m.speecht5.encoder.prenet.load_state_dict({k: torch.from_numpy(v) for k, v in pre.items()})
    return m.to("cuda").speecht5.encoder, sd


def run(enc, sd, oracle, lengths=LENGTHS):
    x, msk = la.synth.batch(lengths)
    out = enc(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda()).last_hidden_state
    torch.cuda.synchronize()
    ref = oracle.encode(x, msk, sd, dtype=torch.float64)
    return out, rel_l2(out, ref)


def scale(sd, key, f):
    sd[key] = (sd[key] * np.float32(f)).astype(np.float32)


def stage_amax(enc, name, layer=-1):
    return max(a for n, l, a in enc.range_report() if name in n and l == layer)


FFN1 = "wrapped_encoder.layers.0.feed_forward.intermediate_dense."


def test_synthetic_model_sits_inside_the_range(oracle):
    """No stage of the ordinary test model is anywhere near a limit: the fp32 re-run must never be what makes other tests pass."""
    enc, sd = build(lambda sd: None)
    _, err = run(enc, sd, oracle)
    assert not enc.last_range_fallback and err < 1e-5
    rep = enc.range_report()
    # conv layers 0-5 (conv0's GroupNorm + GELU output is measured, not bounded: its bound grows with sqrt(frames per clip)), the
    # pos-conv input, and per encoder layer q|k|v + the GELU'd intermediate
    assert len(rep) == 6 + 1 + 2 * LAYERS and "conv_layers.0" in rep[0][0]
    for name, layer, amax in rep:
        assert 0.05 < amax < 500.0, (name, layer, amax)


@pytest.mark.parametrize("factor,expect", [(200.0, 1e3), (2000.0, 1e4)])
def test_large_activations_inside_the_range(oracle, factor, expect):
    """(a) feed-forward intermediate of 1e3 / 1e4: still f16x3, still fp32 class."""
    def mod(sd):
        scale(sd, FFN1 + "weight", factor)
        scale(sd, FFN1 + "bias", factor)
    enc, sd = build(mod)
    _, err = run(enc, sd, oracle)
    amax = stage_amax(enc, "feed_forward intermediate", 0)
    print(f"FFN1 x{factor:g}: max|intermediate| = {amax:.3g}, rel L2 vs fp64 = {err:.2e}")
    assert expect / 3 < amax < 65504 and not enc.last_range_fallback
    assert err < 2e-5


def test_overflow_is_detected_and_rerun_in_fp32(oracle):
    """(a) beyond 65504: hi would be inf.  Default policy: the batch is run again on the exact-fp32 kernels; 'raise': LOCO_E_RANGE
    naming the stage; 'off': the asynchronous forward, after which loco_forward_status reports the same."""
    def mod(sd):
        scale(sd, FFN1 + "weight", 40000.0)
        scale(sd, FFN1 + "bias", 40000.0)
    enc, sd = build(mod)
    out, err = run(enc, sd, oracle)
    assert enc.last_range_fallback and torch.isfinite(out).all()
    print(f"overflow case after the fp32 re-run: rel L2 vs fp64 = {err:.2e}")
    assert err < 1e-5
    enc.range_policy = "raise"
    with pytest.raises(_libmod.LocoError, match=r"feed_forward intermediate.*above"):
        run(enc, sd, oracle)
    enc.range_policy = "off"
    x, msk = la.synth.batch(LENGTHS)
    y = enc(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda()).last_hidden_state
    torch.cuda.synchronize()
    buf = C.create_string_buffer(400)
    assert enc._lib.loco_forward_status(enc._handle, buf, 400) == -5 and b"layers.0" in buf.value
    assert stage_amax(enc, "feed_forward intermediate", 0) > 65504
    assert not enc.last_range_fallback  # nothing was re-run: y is what the fp16 planes gave (inf / NaN inside)
    assert not torch.isfinite(y).all()


def test_tiny_activations_are_detected_and_rerun_in_fp32(oracle):
    """(b) conv weights of 1e-6: the conv stack's outputs sink into fp16's subnormals before LayerNorm(512) brings them back.
    Detected as 'below the range' -> fp32 re-run -> fp32-class result (the fp16 planes alone would be off by > 1e-3)."""
    key = "prenet.feature_encoder.conv_layers.3.conv.weight"

    def mod(sd):
        scale(sd, key, 1e-4)
    enc, sd = build(mod)
    assert float(np.abs(sd[key]).max()) < 1e-5
    out, err = run(enc, sd, oracle)
    assert enc.last_range_fallback
    print(f"tiny conv weights after the fp32 re-run: rel L2 vs fp64 = {err:.2e}")
    assert err < 1e-5
    enc.range_policy = "off"
    _, err_planes = run(enc, sd, oracle)
    print(f"  (the fp16 planes alone: {err_planes:.2e})")
    assert err_planes > 10 * err


@pytest.mark.parametrize("f", [1e-6, 1e6])
def test_weight_level_is_free(oracle, f):
    """(b) weights of magnitude 1e-7 ... 1e+4 where the ACTIVATIONS stay ordinary: the last conv layer feeds LayerNorm(512) in
    fp32, so scaling its weights changes no plane tensor -- only the weight planes, which the per-tensor 2^k absorbs."""
    key = "prenet.feature_encoder.conv_layers.6.conv.weight"
    enc, sd = build(lambda sd: scale(sd, key, f))
    print(f"conv6 weights x{f:g}: max|W| = {float(np.abs(sd[key]).max()):.3g}")
    _, err = run(enc, sd, oracle)
    assert not enc.last_range_fallback
    print(f"  rel L2 vs fp64 = {err:.2e}")
    assert err < 2e-5


def test_weights_beyond_fp16_maximum(oracle):
    """out_proj weights of 6e5 (> 65504): their fp16 planes exist only thanks to the per-tensor scale; the output of that GEMM is
    fp32 and the next LayerNorm brings it back to O(1)."""
    def mod(sd):
        scale(sd, "wrapped_encoder.layers.0.attention.out_proj.weight", 1e7)
        scale(sd, "wrapped_encoder.layers.0.attention.out_proj.bias", 1e7)
    enc, sd = build(mod)
    assert float(np.abs(sd["wrapped_encoder.layers.0.attention.out_proj.weight"]).max()) > 65504
    _, err = run(enc, sd, oracle)
    assert not enc.last_range_fallback
    print(f"out_proj weights of 6e5: rel L2 vs fp64 = {err:.2e}")
    assert err < 2e-5


def test_outlier_channels(oracle):
    """(c) one channel x1000 in feature_projection and in FFN1 (what wav2vec-style checkpoints look like): in range, fp32 class."""
    def mod(sd):
        for key in ("prenet.feature_projection.projection.", FFN1):
            sd[key + "weight"][5] *= np.float32(1000.0)
            sd[key + "bias"][5] *= np.float32(1000.0)
    enc, sd = build(mod)
    _, err = run(enc, sd, oracle)
    a1, a2 = stage_amax(enc, "input of pos_conv_embed"), stage_amax(enc, "feed_forward intermediate", 0)
    print(f"outlier channels: max|feature projection| = {a1:.3g}, max|FFN intermediate| = {a2:.3g}, rel L2 vs fp64 = {err:.2e}")
    assert a1 > 300 and a2 > 300 and not enc.last_range_fallback
    assert err < 2e-5


def test_weight_determined_ranges_are_checked_at_load_time(oracle):
    """LayerNorm / GroupNorm outputs are bounded by their affine weights (sqrt(D) max|gamma| + max|beta|): a gamma of 1e4 in an
    encoder LayerNorm cannot be guaranteed below 65504, gammas of 1e-4 sink its output under 2^-6 -- both are known before any
    forward runs, both end in the fp32 kernels."""
    for f in (1e4, 1e-4):
        def mod(sd, f=f):
            scale(sd, "wrapped_encoder.layers.0.layer_norm.weight", f)
            scale(sd, "wrapped_encoder.layers.0.layer_norm.bias", f)
        enc, sd = build(mod)
        out, err = run(enc, sd, oracle)
        assert enc.last_range_fallback and err < 1e-5, (f, err)
        enc.range_policy = "raise"
        with pytest.raises(_libmod.LocoError, match=r"layers\.0\.layer_norm"):
            run(enc, sd, oracle)
    # conv0's GroupNorm output is MEASURED where it is written (the bound sqrt(frames per clip) * max|gamma| would send
    # every 10-minute batch with gamma >= 47 through f16x3 + sync + fp32): gamma x 2000 gives max|x| ~ 1.7e4 -- inside the range,
    # no re-run; gamma x 1e5 really overflows the planes and is caught by the tracked stage
    for f, fallback in ((2000.0, False), (1.0e5, True)):
        def mod(sd, f=f):
            scale(sd, "prenet.feature_encoder.conv_layers.0.layer_norm.weight", f)
        enc, sd = build(mod)
        _, err = run(enc, sd, oracle, lengths=[16000])
        amax = stage_amax(enc, "conv_layers.0") if not fallback else None
        assert enc.last_range_fallback == fallback and err < 2e-5, (f, err, amax)
        if not fallback:
            assert 5e3 < amax < 65504
    enc.range_policy = "raise"
    with pytest.raises(_libmod.LocoError, match=r"conv_layers\.0"):
        run(enc, sd, oracle, lengths=[16000])


def test_non_finite_output_is_caught_even_when_every_tracked_stage_is_in_range():
    """The range words are maxima taken with fmaxf, which a NaN never enters: an inf / NaN born INSIDE a stage (round 3: the
    attention kernel's row maximum, see test_attention_f16x3_row_max_covers_both_lane_halves) sailed past them and came out as
    NaN embeddings with status OK.  The forward's last LayerNorm now raises a flag for any row whose statistics are not finite.
    Exercised with the one source of NaNs a test can plant without breaking a kernel: a NaN sample in the waveform (HF propagates
    it too) -- 'raise' names the finite check, 'fp32' re-runs (and returns HF's NaNs), 'off' reports through the status call."""
    enc, sd = build(lambda sd: None)
    x, msk = la.synth.batch([16000, 12000])
    x[1, 5000] = np.nan
    xd, md = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
    enc.range_policy = "raise"
    with pytest.raises(_libmod.LocoError, match="non-finite values"):
        enc(input_values=xd, attention_mask=md)
    enc.range_policy = "fp32"
    y = enc(input_values=xd, attention_mask=md).last_hidden_state
    assert enc.last_range_fallback and bool(torch.isfinite(y[0]).all()) and not bool(torch.isfinite(y[1]).all())
    y = enc(input_values=xd[:1], attention_mask=md[:1]).last_hidden_state  # the clean clip alone: nothing flagged
    assert not enc.last_range_fallback and bool(torch.isfinite(y).all())


def test_c_abi_checked_forward_reports_instead_of_rerunning_under_policy_0():
    def mod(sd):
        scale(sd, FFN1 + "weight", 40000.0)
        scale(sd, FFN1 + "bias", 40000.0)
    enc, sd = build(mod)
    x, msk = la.synth.batch([16000])
    xd, md = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
    enc(input_values=xd, attention_mask=md)  # loads the weights, sizes the workspace
    lib, h = enc._lib, enc._handle
    out = torch.empty((1, 49, 768), device="cuda")
    used = C.c_int32(7)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (h, C.c_void_p(xd.data_ptr()), C.c_void_p(md.data_ptr()), 1, 16000, C.c_void_p(out.data_ptr()), None, None,
            C.c_void_p(enc._workspace.data_ptr()), enc._workspace.numel(), st, C.byref(used))
    assert lib.loco_set_range_policy(h, 0) == 0
    assert lib.loco_forward_checked(*args) == -5 and used.value == 0
    assert b"activation range" in lib.loco_last_error()
    assert lib.loco_set_range_policy(h, 1) == 0
    assert lib.loco_forward_checked(*args) == 0 and used.value == 1 and torch.isfinite(out).all()
    assert lib.loco_set_range_policy(h, 2) == -1

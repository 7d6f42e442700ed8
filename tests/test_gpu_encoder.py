"""End-to-end parity of the MI355X encoder path against the golden fixtures (HuggingFace outputs generated
in the build container, tests/golden/make_goldens.py) and against the CPU oracle run on the GPU box.

Bar (BASELINE.json north_star): embeddings within 1e-3 relative L2 of the HF CPU fp32 path.  Both precision modes
are held to 2e-5 on full outputs and on every intermediate stage: "f32" (exact fp32 MFMA) lands at ~1e-6, the
default "f16x3" (three fp16 MFMAs per fp32-class product) at ~3e-6.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import la, model, rel_l2

TOL = 2e-5
PRECISIONS = ["f16x3", "f32"]


def run(lengths, mask=True, hidden=False, taps=False, layers=12, precision=None):
    m, sd = model(layers, precision=precision)
    x, msk = la.synth.batch(lengths)
    st = {} if taps else None
    out = m.speecht5.encoder(input_values=torch.from_numpy(x).cuda(),
                             attention_mask=torch.from_numpy(msk).cuda() if mask else None,
                             output_hidden_states=hidden, stage_taps=st)
    torch.cuda.synchronize()
    return out, st, (x, msk, sd)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_g1_one_second_clip_full_output(precision):
    g = golden("g1_1s.npz")
    out, st, _ = run(g["lengths"], taps=True, precision=precision)
    assert tuple(out.last_hidden_state.shape) == (1, 49, 768)
    assert rel_l2(st["conv_stack"], g["conv_stack"]) < TOL
    assert rel_l2(st["prenet"], g["prenet"]) < TOL
    assert rel_l2(out.last_hidden_state, g["last_hidden_state"]) < TOL


@pytest.mark.parametrize("precision", PRECISIONS)
def test_g2_ragged_batch_every_stage_and_layer(precision):
    g = golden("g2_5s_3s.npz")
    rows = torch.from_numpy(g["rows"])
    out, st, _ = run(g["lengths"], hidden=True, taps=True, precision=precision)
    assert st["frames"].cpu().tolist() == [249, 149]
    for name in ("conv_stack", "feature_projection", "prenet"):
        assert rel_l2(st[name][:, rows], g[name]) < TOL, name
    assert len(out.hidden_states) == 13
    for i, h in enumerate(out.hidden_states):
        assert rel_l2(h[:, rows], g["hidden_states"][i]) < TOL, i
        assert abs(float(h.double().norm()) / g["hidden_stats"][i, 0] - 1) < 1e-5, i
    assert torch.equal(out.hidden_states[-1], out.last_hidden_state)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_g10_second_weight_family_hf_init_with_outlier_channels(precision):
    """a golden whose weights follow HF's own initialisation distributions, with log-normal LayerNorm / GroupNorm
    gains and x100 outlier channels in every encoder LayerNorm (hidden states reach |x| ~ 300 in single channels) -- every stage
    and all 13 hidden states, both fp32-class modes, and the f16x3 range guard must not have re-run anything in fp32."""
    g = golden("g10_hf_init_outliers.npz")
    sd = la.synth.encoder_state_dict_hf_init(0)
    pre, enc_sd = la.synth.split_state_dict(sd)
    m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                         {k: torch.from_numpy(v) for k, v in enc_sd.items()}, precision=precision).cuda()
    enc = m.speecht5.encoder
    x, msk = la.synth.batch(g["lengths"], first_index=int(g["first_index"]))
    rows = torch.from_numpy(g["rows"])
    st = {}
    out = enc(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda(), output_hidden_states=True, stage_taps=st)
    assert st["frames"].cpu().tolist() == [249, 162]
    assert not enc.last_range_fallback, "the outlier-channel model must stay inside the f16x3 range"
    for name in ("conv_stack", "feature_projection", "prenet"):
        assert rel_l2(st[name][:, rows], g[name]) < TOL, name
    # With x100 outlier channels fp32 itself is the limit: HF's own fp32 pass sits up to 5.9e-5 from HF run in float64 (fixture key
    # hf_fp32_error).  The hidden states are therefore compared with the FLOAT64 rows, against a FIXED bar of 1e-4 (north_star:
    # 1e-3), and must not be worse than twice HF's own fp32 figure at the last layer; the actual figures go into the message.
    errs = [rel_l2(h[:, rows], g["hidden_states_fp64"][i]) for i, h in enumerate(out.hidden_states)]
    msg = f"g10 ({precision}) rel L2 vs HF float64 per hidden state: {[f'{e:.1e}' for e in errs]}; HF fp32: {[f'{e:.1e}' for e in g['hf_fp32_error']]}"
    print(msg)
    from conftest import record_figure
    record_figure("g10_hidden_states_vs_hf_float64", precision=precision, rel_l2=errs, hf_fp32=[float(e) for e in g["hf_fp32_error"]])
    # Fixed bars, one per mode (measured: f16x3 1.8e-5 at layer 7 and 7.1e-5 at the last; f32 1.0e-4 and 4.3e-5).  f16x3 adds 32
    # products per MFMA in a tree and 24 such partial sums per K = 768; the exact-fp32 MFMA (32x32x2) chains two products per
    # instruction -- since round 3 in blocks of 64 roundings folded into a second accumulator (gemm_f32.hip; one chain of 384 ... 1 536
    # before: 2.0e-4 at layer 7) -- and the network amplifies a layer's error ~100x by layer 7 (tools/g10_probe.py).
    # round 4: the GELU GEMMs of the f32 mode (FFN1, conv 1-6: the longest K) no longer run on the variant without blocked
    # accumulation -- 1.0e-4 -> 3.9e-5 at layer 7, 4.3e-5 -> 1.7e-5 at the last: the range fallback is not the weaker arithmetic any more
    bar = {"f16x3": 1e-4, "f32": 5e-5}[precision]
    assert max(errs) < bar, msg
    assert errs[-1] < 2 * float(g["hf_fp32_error"][-1]), msg
    for i in range(6):
        assert rel_l2(out.hidden_states[i][:, rows], g["hidden_states"][i]) < TOL, (i, msg)
    if precision == "f16x3":
        rep = dict(((n, l), a) for n, l, a in enc.range_report())
        assert max(rep.values()) < 65504 and min(rep.values()) > 2 ** -6


@pytest.mark.parametrize("precision", PRECISIONS)
def test_g3_headline_shape_batch2(precision):
    g = golden("g3_30s_x2.npz")
    rows = torch.from_numpy(g["rows"])
    out, st, _ = run(g["lengths"], hidden=True, taps=True, precision=precision)
    assert tuple(out.last_hidden_state.shape) == (2, 1499, 768)
    assert rel_l2(st["conv_stack"][:, rows], g["conv_stack"]) < TOL
    assert rel_l2(st["prenet"][:, rows], g["prenet"]) < TOL
    for i in range(13):
        assert rel_l2(out.hidden_states[i][:, rows], g["hidden_states"][i]) < TOL, i
        assert abs(float(out.hidden_states[i].double().norm()) / g["hidden_stats"][i, 0] - 1) < 1e-5, i


@pytest.mark.parametrize("precision", PRECISIONS)
def test_g3r_ragged_30s(precision):
    g = golden("g3r_30s_ragged.npz")
    out, _, _ = run(g["lengths"], precision=precision)
    y = out.last_hidden_state
    assert rel_l2(y[:, torch.from_numpy(g["rows"])], g["last_hidden_state"]) < TOL
    assert abs(float(y.double().norm()) / g["out_stats"][0] - 1) < 1e-5


@pytest.mark.parametrize("precision", PRECISIONS)
def test_g5_T4096_long_clip(precision):
    g = golden("g5_T4096.npz")
    rows = torch.from_numpy(g["rows"])
    out, _, _ = run(g["lengths"], mask=False, hidden=True, precision=precision)
    assert tuple(out.last_hidden_state.shape) == (1, 4096, 768)
    for i in (0, 1, 6, 12):
        assert rel_l2(out.hidden_states[i][:, rows], g["hidden_states"][i]) < TOL, i
    assert abs(float(out.last_hidden_state.double().norm()) / g["hidden_stats"][12, 0] - 1) < 1e-5


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("lengths,mask", [([400], True), ([719], False), ([16000, 9600, 400], True), ([30000, 30000], False),
                                          ([80000], True)])  # [1, 80000]: BASELINE.json configs[0], the single 5 s utterance, as such
def test_against_oracle_on_the_box(lengths, mask, oracle, precision):
    out, _, (x, msk, sd) = run(lengths, mask=mask, precision=precision)
    ref = oracle.encode(x, msk if mask else None, sd)
    assert out.last_hidden_state.shape == ref.shape
    assert rel_l2(out.last_hidden_state, ref) < TOL


def test_padded_frames_are_part_of_the_contract():
    """The reference pickles the padded frames too (…base…py:109-113); batch composition changes them
    (SURVEY.md §7 hard part 5) -- so the short clip alone must NOT equal its rows inside the batch."""
    out, _, _ = run([80000, 48000])
    alone, _, _ = run([48000])
    y = out.last_hidden_state
    assert float(y[1, 149:].abs().max()) > 0.1
    m, _ = model()
    # clip index differs (synth.batch numbers clips from 0), so rebuild the same clip alone
    x1 = torch.from_numpy(la.synth.clip(1, 48000))[None].cuda()
    alone = m.speecht5.encoder(input_values=x1).last_hidden_state
    assert rel_l2(alone[0], y[1, :149]) > 1e-2


def test_deterministic_and_batch_independent():
    """Equal-length clips are independent units: a clip's rows do not depend on its
    batch neighbours, and two runs are bitwise identical."""
    m, _ = model()
    x, msk = la.synth.batch([32000] * 3)
    xs = torch.from_numpy(x).cuda()
    a = m.speecht5.encoder(input_values=xs).last_hidden_state
    b = m.speecht5.encoder(input_values=xs).last_hidden_state
    assert torch.equal(a, b)
    c = m.speecht5.encoder(input_values=xs[[2, 0]]).last_hidden_state
    assert torch.equal(c[0], a[2]) and torch.equal(c[1], a[0])


def test_errors_follow_hf():
    m, _ = model()
    enc = m.speecht5.encoder
    with pytest.raises(ValueError):
        enc(input_values=torch.zeros(1, 399, device="cuda"))  # shorter than one frame
    with pytest.raises(ValueError):
        enc(input_values=torch.zeros(2, 1000, device="cuda"), attention_mask=torch.ones(2, 999, device="cuda"))
    with pytest.raises(RuntimeError):
        enc(input_values=torch.zeros(1, 1000))  # CPU tensor: no fallback path


def test_state_dict_roundtrip_and_legacy_key_names():
    """load_state_dict contract of the reference (…base…py:99-100) incl. the 4.30.2 weight_g/weight_v names
    and the extra pos_sinusoidal_embed.weights entry its pickles carry (map_speecht5_hf.py:164-166)."""
    sd = la.synth.encoder_state_dict(0, layers=2)
    pre, enc = la.synth.split_state_dict(sd)
    pre = {k: torch.from_numpy(v) for k, v in pre.items()}
    enc = {k: torch.from_numpy(v) for k, v in enc.items()}
    legacy = dict(pre)
    legacy["pos_conv_embed.conv.weight_g"] = legacy.pop("pos_conv_embed.conv.parametrizations.weight.original0")
    legacy["pos_conv_embed.conv.weight_v"] = legacy.pop("pos_conv_embed.conv.parametrizations.weight.original1")
    legacy["pos_sinusoidal_embed.weights"] = torch.zeros(4004, 768)
    a = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts(pre, enc, layers=2).to("cuda")
    b = la.SpeechT5ForSpeechToTextMI355X(2)
    res = b.speecht5.encoder.prenet.load_state_dict(legacy)
    assert not res.missing_keys and not res.unexpected_keys
    b.speecht5.encoder.wrapped_encoder.load_state_dict(enc)
    b = b.to("cuda")
    x = torch.from_numpy(la.synth.batch([8000])[0]).cuda()
    assert torch.equal(a.speecht5.encoder(x).last_hidden_state, b.speecht5.encoder(x).last_hidden_state)
    assert set(a.speecht5.encoder.prenet.state_dict()) == set(pre)
    with pytest.raises(RuntimeError):
        b.speecht5.encoder.wrapped_encoder.load_state_dict({"nope": torch.zeros(1)})


def test_headline_batch_32x30s_properties():
    """BASELINE config 2 at full size (32 x 30 s): the first two clips must reproduce the 30 s x 2 golden (equal
    lengths = independent units), every output is finite, and a second run is bitwise identical."""
    g = golden("g3_30s_x2.npz")
    rows = torch.from_numpy(g["rows"])
    m, _ = model()
    x, _ = la.synth.batch([480000] * 32)
    xs = torch.from_numpy(x).cuda()
    y = m.speecht5.encoder(input_values=xs).last_hidden_state
    assert tuple(y.shape) == (32, 1499, 768)
    assert bool(torch.isfinite(y).all())
    assert rel_l2(y[:2, rows], g["hidden_states"][12]) < TOL
    y2 = m.speecht5.encoder(input_values=xs).last_hidden_state
    assert torch.equal(y, y2)


def test_the_c_abi_forward_is_stream_capturable_and_replays_bit_identically():
    """The boundary allocates nothing and never synchronises inside loco_forward_async, so it can be captured into a hipGraph
    (tools/graph_capture.py) and the replay is bit-identical -- also with new data of the same shape, and the captured forward
    writes its own status block on every replay.  No timing claim: a replay takes what the eager forward takes (2.017 vs 2.016 ms
    for one 5 s utterance, profiles/r03_hipgraph_trace.txt) -- the path is bound by its chain of small dependent kernels, not by
    launches -- which is why the encoder module has no use_graphs switch."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from graph_capture import CapturedForward
    m, _ = model()
    enc = m.speecht5.encoder
    x, msk = la.synth.batch([80000])
    xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
    eager = enc(input_values=xs, attention_mask=ms).last_hidden_state
    cap = CapturedForward(enc, xs, ms)
    for _ in range(3):
        out = cap.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager) and cap.status_code() == 0
    x2 = torch.from_numpy(la.synth.batch([80000], first_index=5)[0]).cuda()
    out = cap.replay(x2).clone()  # same shape, new data -> same graph
    torch.cuda.synchronize()
    assert torch.equal(out, enc(input_values=x2, attention_mask=ms).last_hidden_state)


def test_two_stream_half_batches_are_bit_identical():
    """loco_set_streams(2): a large batch runs as two half-batches on two HIP streams (fork/join with events).  Clips are
    independent, so the result must equal the single-stream pass bit for bit -- ragged lengths, odd batch size, consecutive
    calls that reuse the workspace, and hipGraph capture of the fork/join included."""
    m, _ = model(layers=2)
    enc = m.speecht5.encoder
    lens = [480000 - 1731 * i for i in range(23)]  # 23 clips of ~30 s: 12 + 11 (halves of >= 1024 frames run on two streams)
    x, msk = la.synth.batch(lens)
    xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
    enc.streams = 1
    one = enc(input_values=xs, attention_mask=ms).last_hidden_state
    frames_one = enc.last_frames.clone() if getattr(enc, "last_frames", None) is not None else None
    enc.streams = 2
    try:
        two = enc(input_values=xs, attention_mask=ms).last_hidden_state
        assert torch.equal(one, two)
        # the caller's stream must see the join: a kernel queued right after reads finished data
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            three = enc(input_values=xs, attention_mask=ms).last_hidden_state
            chk = three.sum()
        s.synchronize()
        assert torch.equal(three, one) and float(chk) == float(one.sum())
        nomask = enc(input_values=xs).last_hidden_state
        enc.streams = 1
        assert torch.equal(nomask, enc(input_values=xs).last_hidden_state)
        enc.streams = 2
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from graph_capture import CapturedForward
        enc(input_values=xs, attention_mask=ms)  # loco_set_streams(2) is in effect for the capture
        cap = CapturedForward(enc, xs, ms.to(torch.int32))  # the fork / join events of the two-stream schedule are capturable
        a = cap.replay().clone()
        b = cap.replay().clone()
        torch.cuda.synchronize()
        assert torch.equal(a, one) and torch.equal(b, one)
        if frames_one is not None:
            assert torch.equal(cap.frames, frames_one)
    finally:
        enc.streams = 2


@pytest.mark.parametrize("B,T", [(1, 1), (1, 2), (3, 63), (2, 64), (2, 65), (1, 127), (3, 128), (1, 129), (2, 256), (2, 257), (4, 256),
                                 (4, 257), (1, 1023), (1, 1025), (8, 1024), (8, 1025)])
def test_shape_sweep_around_tile_and_dispatch_boundaries(B, T, oracle):
    """Frame counts around every size the kernels branch on: the 64-key attention tile, the 128/256-row GEMM tiles, the
    M = 512 / 1024 / 8192 dispatch thresholds (128-row tiles, split-K), ragged masks included.  Two encoder layers, fp32
    oracle, default precision."""
    m, sd = model(layers=2)
    enc = m.speecht5.encoder
    L = 320 * T + 80
    assert la.synth.conv_out_length(L) == T
    lengths = [L - (173 * i * L) // (4 * max(B, 2)) // 7 * 7 for i in range(B)]  # clip 0 full length, the others shorter
    lengths = [max(400, n) for n in lengths]
    x, msk = la.synth.batch(lengths, first_index=T % 17)
    y = enc(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda()).last_hidden_state
    assert tuple(y.shape) == (B, T, 768) and bool(torch.isfinite(y).all())
    ref = oracle.encode(x, msk, sd)
    assert rel_l2(y, ref) < TOL, (B, T)
    assert enc.last_frames.tolist() == [la.synth.conv_out_length(n) for n in lengths]


def test_c_abi_error_codes():
    """What a C caller sees for bad arguments: negative codes and a message, never a crash or a silent wrong result."""
    import ctypes as C

    from gpu_util import lib, ptr, stream
    m, _ = model(layers=1)
    enc = m.speecht5.encoder
    enc(input_values=torch.zeros(1, 16000, device="cuda"))  # creates + finalises the handle
    L_ = lib()
    h = enc._handle
    x = torch.zeros(2, 16000, device="cuda")
    out = torch.empty(2, 49, 768, device="cuda")
    need = int(L_.loco_workspace_bytes(h, 2, 16000))
    ws = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
    ok = L_.loco_forward(h, ptr(x), None, 2, 16000, ptr(out), None, None, ptr(ws), need, stream())
    assert ok == 0
    # workspace one byte short -> LOCO_E_WORKSPACE, message names both sizes
    rc = L_.loco_forward(h, ptr(x), None, 2, 16000, ptr(out), None, None, ptr(ws), need - 1, stream())
    assert rc < 0 and str(need).encode() in L_.loco_last_error()
    # misaligned workspace
    rc = L_.loco_forward(h, ptr(x), None, 2, 16000, ptr(out), None, None, C.c_void_p(ws.data_ptr() + 4), need, stream())
    assert rc < 0 and b"aligned" in L_.loco_last_error()
    # null pointers, empty batch, input shorter than one frame
    assert L_.loco_forward(h, None, None, 2, 16000, ptr(out), None, None, ptr(ws), need, stream()) < 0
    assert L_.loco_forward(h, ptr(x), None, 0, 16000, ptr(out), None, None, ptr(ws), need, stream()) < 0
    assert L_.loco_forward(h, ptr(x), None, 2, 399, ptr(out), None, None, ptr(ws), need, stream()) < 0
    assert L_.loco_workspace_bytes(h, 2, 399) == 0 and L_.loco_output_frames(399) == 0 and L_.loco_output_frames(400) == 1
    # loco_forward_async: a status block is mandatory and 8-byte aligned, the precision argument is -1 or a documented mode, and a
    # failed enqueue leaves the block invalid (loco_status_check refuses it) rather than describing some earlier forward
    status = torch.zeros(int(L_.loco_status_bytes()) + 8, dtype=torch.uint8).pin_memory()
    out2 = torch.empty_like(out)  # the checks below must not disturb `out`, which the end of the test compares with
    async_args = lambda prec, ws_bytes, st_ptr: (h, prec, ptr(x), None, 2, 16000, ptr(out2), None, None, ptr(ws), ws_bytes, stream(), st_ptr)
    assert L_.loco_forward_async(*async_args(-1, need, None)) == -1
    assert L_.loco_forward_async(*async_args(-1, need, C.c_void_p(status.data_ptr() + 4))) == -1 and b"aligned" in L_.loco_last_error()
    assert L_.loco_forward_async(*async_args(5, need, C.c_void_p(status.data_ptr()))) == -1 and b"f16x2" in L_.loco_last_error()
    assert L_.loco_forward_async(*async_args(-1, need, C.c_void_p(status.data_ptr()))) == 0
    torch.cuda.synchronize()
    assert L_.loco_status_check(C.c_void_p(status.data_ptr()), None, 0) == 0
    assert L_.loco_forward_async(*async_args(-1, need - 1, C.c_void_p(status.data_ptr()))) == -3  # LOCO_E_WORKSPACE
    assert L_.loco_status_check(C.c_void_p(status.data_ptr()), None, 0) == -1 and b"status block" in L_.loco_last_error()
    assert L_.loco_forward_async(*async_args(0, need, C.c_void_p(status.data_ptr()))) == 0  # exact-fp32 mode for this call only
    torch.cuda.synchronize()
    assert L_.loco_status_check(C.c_void_p(status.data_ptr()), None, 0) == 0 and L_.loco_get_precision(h) == 1
    assert not torch.equal(out2, out) and rel_l2(out2, out) < 1e-5  # another arithmetic, the same embeddings
    # knobs reject values outside their domain
    assert L_.loco_set_streams(h, 3) < 0 and L_.loco_set_streams(h, 2) == 0
    assert L_.loco_set_precision(h, 7) < 0 and L_.loco_set_precision(h, 1) == 0
    # an unknown weight name / a wrong shape are reported by name
    w = torch.zeros(3, 3, device="cuda")
    shp = (C.c_int64 * 2)(3, 3)
    assert L_.loco_set_weight(h, b"wrapped_encoder.layers.0.nope.weight", ptr(w), shp, 2) < 0 and b"nope" in L_.loco_last_error()
    assert L_.loco_set_weight(h, b"wrapped_encoder.layer_norm.weight", ptr(w), shp, 2) < 0 and b"size mismatch" in L_.loco_last_error()
    # the handle still works after all of that
    y = enc(input_values=x).last_hidden_state
    assert torch.equal(y, out)


def test_concurrent_forwards_on_two_user_streams_are_bit_identical():
    """A C caller may enqueue independent forwards of one handle on different streams (from one host thread).  Two half
    batches on two streams, three steps back to back with no synchronisation in between -- so that one stream's front-end
    kernels share CUs with the other's late-layer kernels -- must reproduce the single pass bit for bit, every time.
    (Regression: a packed-math variant of the conv0 kernel did not; tools/race_probe.py is the stand-alone form.)"""
    import ctypes as C

    from gpu_util import lib
    m, _ = model()
    enc = m.speecht5.encoder
    B, L = 32, 480000
    x, msk = la.synth.batch([L] * B)
    xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda().int()
    enc.streams = 1
    try:
        ref = enc(input_values=xs, attention_mask=ms).last_hidden_state
        L_, h = lib(), enc._handle
        L_.loco_set_streams(h, 1)
        T = int(L_.loco_output_frames(L))
        need = int(L_.loco_workspace_bytes(h, B // 2, L))
        wss = [torch.empty(need, dtype=torch.uint8, device="cuda") for _ in range(2)]
        out = torch.empty(B, T, 768, device="cuda")
        streams = [torch.cuda.Stream() for _ in range(2)]
        # forwards of one handle that overlap go through loco_forward_async, each with its own status block (include/loco_asr.h:
        # loco_forward keeps its status in the handle and must not overlap with itself)
        stat = [torch.zeros(int(L_.loco_status_bytes()), dtype=torch.uint8).pin_memory() for _ in range(2)]
        for trial in range(8):
            out.zero_()
            torch.cuda.synchronize()
            for _ in range(3):
                for i in range(2):
                    a, b = i * (B // 2), (i + 1) * (B // 2)
                    rc = L_.loco_forward_async(h, -1, C.c_void_p(xs[a:b].data_ptr()), C.c_void_p(ms[a:b].data_ptr()), b - a, L,
                                               C.c_void_p(out[a:b].data_ptr()), None, None, C.c_void_p(wss[i].data_ptr()), need,
                                               C.c_void_p(streams[i].cuda_stream), C.c_void_p(stat[i].data_ptr()))
                    assert rc == 0, L_.loco_last_error()
            torch.cuda.synchronize()
            assert all(L_.loco_status_check(C.c_void_p(st_.data_ptr()), None, 0) == 0 for st_ in stat)
            bad = [i for i in range(B) if not torch.equal(out[i], ref[i])]
            assert not bad, (trial, bad)
    finally:
        enc.streams = 2


def test_two_streams_on_a_mid_size_batch():
    """16 x 5 s: halves of 1992 frames take the split-K GEMM path with other K slices than the whole batch would, so the two
    schedules agree to fp32 summation order, not bit for bit; each is deterministic and within the parity bar."""
    m, sd = model()
    enc = m.speecht5.encoder
    lens = [80000 - 997 * i for i in range(16)]
    x, msk = la.synth.batch(lens)
    xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
    try:
        enc.streams = 1
        one = enc(input_values=xs, attention_mask=ms).last_hidden_state
        enc.streams = 2
        two = enc(input_values=xs, attention_mask=ms).last_hidden_state
        again = enc(input_values=xs, attention_mask=ms).last_hidden_state
        assert torch.equal(two, again)
        assert rel_l2(two, one) < 5e-6
    finally:
        enc.streams = 2


def test_from_pretrained_gives_the_bits_of_from_state_dicts(tmp_path):
    """a checkpoint directory on disk (model.safetensors with the keys of SpeechT5ForSpeechToText, decoder tensors
    beside them) through from_pretrained -- the fine-tuned script's call, …finetuned…py:95 -- against the same weights handed over
    as the two state dicts (the base script's calls, …base…py:99-100): bit-identical output."""
    from safetensors.torch import save_file
    layers = 3
    sd = la.synth.encoder_state_dict(5, layers=layers)
    named = {"speecht5.encoder." + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    named["speecht5.decoder.wrapped_decoder.layer_norm.weight"] = torch.ones(768)
    save_file(named, str(tmp_path / "model.safetensors"))
    a = la.SpeechT5ForSpeechToTextMI355X.from_pretrained(str(tmp_path)).cuda()
    pre, enc_sd = la.synth.split_state_dict(sd)
    b = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                         {k: torch.from_numpy(v) for k, v in enc_sd.items()}, layers=layers).cuda()
    x, m = la.synth.batch([40000, 25000], first_index=9)
    xa, ma = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
    ya = a.speecht5.encoder(input_values=xa, attention_mask=ma).last_hidden_state
    yb = b.speecht5.encoder(input_values=xa, attention_mask=ma).last_hidden_state
    assert a.speecht5.encoder.num_layers == layers and torch.equal(ya, yb) and bool(torch.isfinite(ya).all())

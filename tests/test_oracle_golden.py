"""The CPU oracle against the golden fixtures generated from HuggingFace (tests/golden/make_goldens.py).

This is what pins the oracle: the reference has no tests, so the committed HF
outputs on seeded inputs are the ground truth the restatement must reproduce.  fp32 tolerance:
relative L2 <= 5e-6 (both sides are fp32 evaluations of the same maths; observed ~8e-7).
"""
import numpy as np
import torch

from conftest import golden, rel_l2

TOL = 5e-6


def test_g1_one_second_full_output(oracle, synth, state_dict):
    g = golden("g1_1s.npz")
    x, m = synth.batch(g["lengths"])
    taps = {}
    y = oracle.encode(x, m, state_dict, taps=taps)
    assert y.shape == (1, 49, 768)
    assert rel_l2(taps["conv_stack"], g["conv_stack"]) < TOL
    assert rel_l2(taps["prenet"], g["prenet"]) < TOL
    assert rel_l2(y, g["last_hidden_state"]) < TOL


def test_g2_ragged_batch_every_stage(oracle, synth, state_dict):
    g = golden("g2_5s_3s.npz")
    x, m = synth.batch(g["lengths"])
    rows = g["rows"]
    taps, hs = {}, []
    y = oracle.encode(x, m, state_dict, taps=taps, hidden_states=hs)
    assert y.shape == (2, 249, 768)
    for name in ("conv_stack", "feature_projection", "prenet"):
        assert rel_l2(taps[name][:, rows], g[name]) < TOL, name
    assert len(hs) == 13
    for i, h in enumerate(hs):
        assert rel_l2(h[:, rows], g["hidden_states"][i]) < TOL, i
        assert abs(float(h.double().norm()) / g["hidden_stats"][i, 0] - 1) < 1e-6
    # padded frames of the short clip are NOT zero and are part of the contract (SURVEY.md §7 hard part 5)
    assert float(y[1, 149:].abs().max()) > 0.1


def test_g10_second_weight_family_hf_init_with_outlier_channels(oracle, synth):
    """HF's own init distributions + log-normal norm gains + x100 outlier channels (synth.encoder_state_dict_hf_init): the oracle
    restates the path, not the first weight family."""
    g = golden("g10_hf_init_outliers.npz")
    sd = synth.encoder_state_dict_hf_init(0)
    x, m = synth.batch(g["lengths"], first_index=int(g["first_index"]))
    rows = g["rows"]
    taps, hs = {}, []
    y = oracle.encode(x, m, sd, taps=taps, hidden_states=hs)
    assert y.shape == (2, 249, 768)
    for name in ("conv_stack", "feature_projection", "prenet"):
        assert rel_l2(taps[name][:, rows], g[name]) < TOL, name
    # The outlier channels make the function ill-conditioned for fp32: HF's own fp32 pass is up to 5.9e-5 off HF run in float64
    # (recorded in the fixture), so the hidden states are compared with the FLOAT64 rows.  Fixed bars: 5e-5 for this fp32 restatement
    # (observed: <= 1.8e-5, 3.6e-6 at the last layer), and the fixture's own fp32 figure stays below 1e-4.
    errs = [rel_l2(h[:, rows], g["hidden_states_fp64"][i]) for i, h in enumerate(hs)]
    assert max(errs) < 5e-5, [f"{e:.1e}" for e in errs]
    assert 2e-5 < g["hf_fp32_error"].max() < 1e-4
    for i in range(6):  # the early layers are well conditioned: there the fp32 rows agree to the usual bar too
        assert rel_l2(hs[i][:, rows], g["hidden_states"][i]) < 6e-6, i
    assert g["hidden_stats"][:, 2].max() > 100  # the outlier channels are there


def test_g3_headline_shape_batch2(oracle, synth, state_dict):
    g = golden("g3_30s_x2.npz")
    x, m = synth.batch(g["lengths"])
    hs = []
    y = oracle.encode(x, m, state_dict, hidden_states=hs)
    assert y.shape == (2, 1499, 768)
    for i in (0, 1, 6, 12):
        assert rel_l2(hs[i][:, g["rows"]], g["hidden_states"][i]) < TOL, i


def test_g3r_ragged_30s(oracle, synth, state_dict):
    g = golden("g3r_30s_ragged.npz")
    assert list(g["lengths"]) == synth.mixed_lengths(3, 480000)
    x, m = synth.batch(g["lengths"])
    y = oracle.encode(x, m, state_dict)
    assert rel_l2(y[:, g["rows"]], g["last_hidden_state"]) < TOL
    assert abs(float(y.double().norm()) / g["out_stats"][0] - 1) < 1e-6


def test_g4_attention_module_with_key_padding(oracle, synth, state_dict):
    g = golden("g4_attention_l3.npz")
    h = torch.from_numpy(synth.hashed_uniform("g4.hidden", (2, 200, 768), 7)) * 1.7
    pe_k = torch.from_numpy(state_dict["wrapped_encoder.embed_positions.pe_k.weight"])
    y = oracle.attention(h, torch.from_numpy(g["frames"]), state_dict, "wrapped_encoder.layers.3.", pe_k, q_block=64)
    assert rel_l2(y, g["out"]) < TOL


def test_g5_T4096_blocked_attention(oracle, synth, state_dict):
    g = golden("g5_T4096.npz")
    x, _ = synth.batch(g["lengths"])
    hs = []
    y = oracle.encode(x, None, state_dict, hidden_states=hs, q_block=256)
    assert y.shape == (1, 4096, 768)
    for i in (0, 1, 12):
        assert rel_l2(hs[i][:, g["rows"]], g["hidden_states"][i]) < TOL, i


def test_frame_counts(oracle, synth):
    # 5 s -> 249, 30 s -> 1499, 10 min -> 29 999
    assert oracle.feat_extract_output_lengths(80000) == 249 == synth.conv_out_length(80000)
    assert oracle.feat_extract_output_lengths(480000) == 1499
    assert oracle.feat_extract_output_lengths(9600000) == 29999
    assert oracle.feat_extract_output_lengths(400) == 1
    m = torch.zeros(2, 1000, dtype=torch.int32)
    m[0, :1000] = 1
    m[1, :400] = 1
    assert oracle.frame_counts(m, 2).tolist() == [2, 1]


def _text_sd(synth, state_dict):
    sd = dict(state_dict)
    sd.update(synth.text_prenet_state_dict(0))
    return sd


def test_g6_text_encoder_without_and_with_mask(oracle, synth, state_dict):
    """"next" row f-4: SpeechT5EncoderWithTextPrenet as the reference's text branch calls it (ids only: pads attend), and
    with the tokenizer's right-padding mask; plus a row of max_text_positions tokens."""
    g = golden("g6_text.npz")
    sd = _text_sd(synth, state_dict)
    lengths = list(g["lengths"])
    ids, mask = synth.token_ids(3, 57, lengths=lengths)
    rows = g["short_rows"]
    pre = oracle.text_prenet(ids, sd)
    assert pre.shape == (3, 57, 768)
    assert rel_l2(pre[:, rows], g["prenet"]) < 1e-7  # gather + one multiply-add: exact up to the last bit
    hs = []
    y = oracle.encode_text(ids, None, sd, hidden_states=hs)
    assert rel_l2(y[:, rows], g["nomask_last"]) < TOL
    assert len(hs) == 13
    for i, h in enumerate(hs):
        assert abs(float(h.double().norm()) / g["nomask_hidden_stats"][i, 0] - 1) < 1e-6, i
    hs = []
    ym = oracle.encode_text(ids, mask, sd, hidden_states=hs)
    assert rel_l2(ym, g["masked_last"]) < TOL
    for i, h in enumerate(hs):
        assert abs(float(h.double().norm()) / g["masked_hidden_stats"][i, 0] - 1) < 1e-6, i
    assert rel_l2(ym[1, :31], y[1, :31]) > 1e-3  # the mask matters: pads are keys for every row when it is absent
    long_ids, _ = synth.token_ids(1, synth.MAX_TEXT_POSITIONS, seed=11)
    yl = oracle.encode_text(long_ids, None, sd)
    assert rel_l2(yl[:, g["long_rows"]], g["long_last"]) < TOL
    assert abs(float(yl.double().norm()) / g["long_stats"][0] - 1) < 1e-6

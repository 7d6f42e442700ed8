"""Direct comparison of the oracle with the installed HuggingFace module (skipped where transformers is absent).
Complements the committed goldens: a different seed, a different ragged batch, and the 4.30.2 key spelling."""
import importlib

import pytest
import torch

from conftest import rel_l2

tr = pytest.importorskip("transformers")


@pytest.fixture(scope="module")
def hf_and_sd(synth):
    sd = synth.encoder_state_dict(3)
    enc = tr.SpeechT5ForSpeechToText(tr.SpeechT5Config()).eval().speecht5.encoder
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return enc, sd


def test_ragged_batch(hf_and_sd, oracle, synth):
    enc, sd = hf_and_sd
    x, m = synth.batch([40000, 23456, 31999], first_index=10)
    with torch.no_grad():
        ref = enc(input_values=torch.from_numpy(x), attention_mask=torch.from_numpy(m)).last_hidden_state
    assert rel_l2(oracle.encode(x, m, sd), ref) < 5e-6


def test_legacy_weight_norm_key_names(hf_and_sd, oracle, synth):
    enc, sd = hf_and_sd
    legacy = dict(sd)
    legacy["prenet.pos_conv_embed.conv.weight_g"] = legacy.pop("prenet.pos_conv_embed.conv.parametrizations.weight.original0")
    legacy["prenet.pos_conv_embed.conv.weight_v"] = legacy.pop("prenet.pos_conv_embed.conv.parametrizations.weight.original1")
    x, _ = synth.batch([16000])
    with torch.no_grad():
        ref = enc(input_values=torch.from_numpy(x)).last_hidden_state
    assert rel_l2(oracle.encode(x, None, legacy), ref) < 5e-6


def test_frame_mask_helper(hf_and_sd, oracle):
    enc, _ = hf_and_sd
    m = torch.zeros(3, 50000, dtype=torch.int32)
    for i, n in enumerate((50000, 400, 33333)):
        m[i, :n] = 1
    T = int(oracle.feat_extract_output_lengths(50000))
    hf = enc.prenet._get_feature_vector_attention_mask(T, m)
    fr = oracle.frame_counts(m, T)
    assert hf.sum(1).tolist() == fr.tolist()

"""Checkpoints on disk: the two ways the reference gets its weights --
``SpeechT5ForSpeechToText.from_pretrained("microsoft/speecht5_asr")`` (extract_speecht5_finetuned_embeddings_slurp.py:95) and three
pickles made from a fairseq ``speecht5_base.pt`` (extract_speecht5_base_embeddings_slurp.py:40-49, map_speecht5_hf.py:157-181) -- as one
call / one command each, exercised on synthetic checkpoints written here (no real checkpoint is reachable offline)."""
import importlib
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

la = importlib.import_module("loco-asr_amd")
cm = importlib.import_module("loco-asr_amd.checkpoint_map")
LAYERS = 2


def _hf_named(sd, spelling="5.x"):
    """synth weights under the key names of a SpeechT5ForSpeechToText checkpoint, plus tensors of other sub-modules"""
    out = {}
    for k, v in sd.items():
        if spelling == "4.30.2":
            k = k.replace("pos_conv_embed.conv.parametrizations.weight.original0", "pos_conv_embed.conv.weight_g")
            k = k.replace("pos_conv_embed.conv.parametrizations.weight.original1", "pos_conv_embed.conv.weight_v")
        out["speecht5.encoder." + k] = torch.from_numpy(np.ascontiguousarray(v))
    out["speecht5.decoder.wrapped_decoder.layers.0.self_attn.q_proj.weight"] = torch.zeros(4, 4)
    out["text_decoder_postnet.lm_head.weight"] = torch.zeros(3, 5)
    return out


@pytest.fixture(scope="module")
def sd():
    return la.synth.encoder_state_dict(3, layers=LAYERS)


def _check_loaded(model, sd):
    enc = model.speecht5.encoder
    assert enc.num_layers == LAYERS
    pre, wrapped = la.synth.split_state_dict(sd)
    got_pre, got_enc = enc.prenet.state_dict(), enc.wrapped_encoder.state_dict()
    assert set(got_pre) == set(pre) and set(got_enc) == set(wrapped)
    for k, v in pre.items():
        assert torch.equal(got_pre[k], torch.from_numpy(v)), k
    for k, v in wrapped.items():
        assert torch.equal(got_enc[k], torch.from_numpy(v)), k


@pytest.mark.parametrize("spelling", ["5.x", "4.30.2"])
def test_from_pretrained_safetensors_directory(tmp_path, sd, spelling):
    from safetensors.torch import save_file
    save_file(_hf_named(sd, spelling), str(tmp_path / "model.safetensors"))
    with open(tmp_path / "config.json", "w") as fh:
        json.dump({"model_type": "speecht5", "hidden_size": 768, "encoder_layers": LAYERS, "encoder_ffn_dim": 3072, "conv_dim": [512] * 7}, fh)
    _check_loaded(la.SpeechT5ForSpeechToTextMI355X.from_pretrained(str(tmp_path)), sd)
    _check_loaded(la.SpeechT5ForSpeechToTextMI355X.from_pretrained(str(tmp_path / "model.safetensors"), precision="f32"), sd)


def test_from_pretrained_pytorch_bin_and_sharded_index(tmp_path, sd):
    named = _hf_named(sd)
    a = tmp_path / "bin"
    os.makedirs(a)
    torch.save(named, a / "pytorch_model.bin")
    _check_loaded(la.SpeechT5ForSpeechToTextMI355X.from_pretrained(str(a)), sd)
    # two safetensors shards behind an index, as transformers writes checkpoints above its shard size
    from safetensors.torch import save_file
    b = tmp_path / "sharded"
    os.makedirs(b)
    keys = sorted(named)
    shards = {"model-00001-of-00002.safetensors": keys[: len(keys) // 2], "model-00002-of-00002.safetensors": keys[len(keys) // 2:]}
    for name, ks in shards.items():
        save_file({k: named[k] for k in ks}, str(b / name))
    with open(b / "model.safetensors.index.json", "w") as fh:
        json.dump({"metadata": {}, "weight_map": {k: name for name, ks in shards.items() for k in ks}}, fh)
    _check_loaded(la.SpeechT5ForSpeechToTextMI355X.from_pretrained(str(b)), sd)


def test_from_pretrained_fails_by_name(tmp_path, sd):
    from safetensors.torch import save_file
    named = _hf_named(sd)
    missing = "speecht5.encoder.wrapped_encoder.layers.1.feed_forward.output_dense.bias"
    del named[missing]
    save_file(named, str(tmp_path / "model.safetensors"))
    with pytest.raises(RuntimeError, match="layers.1.feed_forward.output_dense.bias"):
        la.SpeechT5ForSpeechToTextMI355X.from_pretrained(str(tmp_path))
    # a hub name that is not in the local cache: a clear refusal, never a download
    with pytest.raises(FileNotFoundError, match="never downloads"):
        la.SpeechT5ForSpeechToTextMI355X.from_pretrained("microsoft/speecht5_asr")
    # another geometry: refused from config.json, before any tensor is read
    other = tmp_path / "large"
    os.makedirs(other)
    with open(other / "config.json", "w") as fh:
        json.dump({"hidden_size": 1024, "encoder_attention_heads": 16}, fh)
    with pytest.raises(ValueError, match="hidden_size"):
        la.SpeechT5ForSpeechToTextMI355X.from_pretrained(str(other))
    # a checkpoint of something else
    junk = tmp_path / "junk"
    os.makedirs(junk)
    save_file({"bert.embeddings.weight": torch.zeros(2, 2)}, str(junk / "model.safetensors"))
    with pytest.raises(KeyError, match="not a SpeechT5"):
        la.SpeechT5ForSpeechToTextMI355X.from_pretrained(str(junk))


def test_hub_name_is_served_from_the_local_cache_only(tmp_path, sd, monkeypatch):
    from safetensors.torch import save_file
    snap = tmp_path / "hub" / "models--microsoft--speecht5_asr" / "snapshots" / "abc123"
    os.makedirs(snap)
    save_file(_hf_named(sd), str(snap / "model.safetensors"))
    monkeypatch.setenv("HF_HUB_CACHE", str(tmp_path / "hub"))
    _check_loaded(la.SpeechT5ForSpeechToTextMI355X.from_pretrained("microsoft/speecht5_asr"), sd)


def test_fairseq_pt_to_the_three_pickles_command(tmp_path, sd):
    """`python -m loco-asr_amd.checkpoint_map speecht5_base.pt --out extracted/speecht5/mapping/` writes what …base…py:40-49 opens;
    extract.py's loader reads them back into the weights the checkpoint held."""
    import argparse
    pre, enc = la.synth.split_state_dict(sd)
    fair = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in cm.to_fairseq_names(pre, enc).items()}
    tp = la.synth.text_prenet_state_dict(0)
    fair["text_encoder_prenet.encoder_prenet.0.weight"] = torch.from_numpy(tp["text_prenet.embed_tokens.weight"])
    fair["text_encoder_prenet.encoder_prenet.1.alpha"] = torch.tensor(1.37)
    fair["decoder.layers.0.fc1.weight"] = torch.zeros(2, 2)
    ckpt = {"model": fair, "args": argparse.Namespace(arch="t5_transformer_base"), "cfg": None, "extra_state": {"epoch": 3}}
    pt = tmp_path / "speecht5_base.pt"
    torch.save(ckpt, pt)
    out = tmp_path / "extracted" / "speecht5" / "mapping"
    r = subprocess.run([sys.executable, "-m", "loco-asr_amd.checkpoint_map", str(pt), "--out", str(out)], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(out)) == ["encoder_state_dict.pickle", "speech_prenet_state_dict.pickle", "text_prenet_state_dict.pickle"]
    extract = importlib.import_module("loco-asr_amd.extract")
    enc_l = extract.load_state_dict_file(str(out / "encoder_state_dict.pickle"))
    pre_l = extract.load_state_dict_file(str(out / "speech_prenet_state_dict.pickle"))
    txt_l = extract.load_state_dict_file(str(out / "text_prenet_state_dict.pickle"))
    assert "pos_sinusoidal_embed.weights" in pre_l and tuple(pre_l["pos_sinusoidal_embed.weights"].shape) == (4004, 768)
    with open(out / "encoder_state_dict.pickle", "rb") as fh:
        assert all(torch.is_tensor(v) for v in pickle.load(fh).values())  # tensors, as the reference's Mapping stores them
    model = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts(pre_l, enc_l, layers=LAYERS)
    # the reference's pickles spell the weight-norm pair weight_g / weight_v (transformers 4.30.2); the module re-keys them
    _check_loaded(model, sd)
    assert torch.equal(txt_l["embed_tokens.weight"], fair["text_encoder_prenet.encoder_prenet.0.weight"]) and float(txt_l["encode_positions.alpha"]) == pytest.approx(1.37)
    # a speech-path key the rules do not know must stop the command, not vanish
    fair["encoder.layers.0.mystery.weight"] = torch.zeros(1)
    torch.save({"model": fair}, pt)
    r = subprocess.run([sys.executable, "-m", "loco-asr_amd.checkpoint_map", str(pt), "--out", str(out)], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "encoder.layers.0.mystery.weight" in r.stderr

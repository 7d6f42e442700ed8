#!/usr/bin/env python3
"""Golden key tables for the checkpoint import ("next" row f-3), produced by THE REFERENCE'S OWN `Mapping` class.

Runs in the build container only: it imports /root/reference/speech_text/map_speecht5_hf.py (plain Python, no
dependencies of its own) and drives it the way the reference's notebook does --
``Mapping(model_asr, model_tts, ckpt)`` with ``model_asr = SpeechT5ForSpeechToText``, ``model_tts =
SpeechT5ForTextToSpeech`` and ``ckpt = torch.load(<fairseq speecht5_base.pt>)``.  No fairseq checkpoint is reachable
offline, and `Mapping` looks at NAMES only (the one shape it reads, map_speecht5_hf.py:35, is never used), so ``ckpt['model']``
here is a synthetic dict: the fairseq SpeechT5 key names (microsoft/SpeechT5 `speecht5/models/speecht5.py`: modules
``encoder``, ``speech_encoder_prenet``, ``text_encoder_prenet``, decoder and post-nets) with a distinct integer as each
value, so that every entry of the produced dicts can be traced back to the checkpoint key it was taken from.  A few
keys of other sub-modules and two keys `Mapping` has no rule for are included on purpose.

The HF models are instantiated locally from ``SpeechT5Config()`` (random init; only ``named_parameters()`` names matter).
Two runs are stored:

  "hf_installed"  against the installed transformers (5.x): its weight-norm parameters are spelled
                  ``pos_conv_embed.conv.parametrizations.weight.original0/1``, which the reference's 4.30.2-era rule
                  (map_speecht5_hf.py:139-145: fairseq ``pos_conv.0.weight_g`` -> ``pos_conv_embed.conv.weight_g``) cannot
                  match -- the fixture records exactly what it does produce (the two keys are MISSING from the prenet dict).
  "hf_4_30_2"     against a view of the same models whose ``named_parameters()`` use the 4.30.2 spelling
                  (``...conv.weight_g`` / ``...conv.weight_v``, the reference's pinned version, requirements.txt:151):
                  a pure rename of those two names, nothing else touched.  This is the table the reference's pickles
                  (extracted/speecht5/mapping/*.pickle) were built with.

    python tests/golden/make_mapping_goldens.py        # writes tests/golden/g9_mapping.json
"""
import importlib.util
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/speech_text/map_speecht5_hf.py"
LAYERS = 12


def fairseq_names():
    """Key names of a fairseq SpeechT5-base checkpoint's ``['model']`` dict that concern the encoder side, in module order."""
    names = []
    p = "speech_encoder_prenet."
    names += [p + "mask_emb"]
    names += [p + f"feature_extractor.conv_layers.{i}.0.weight" for i in range(7)]
    names += [p + "feature_extractor.conv_layers.0.2.weight", p + "feature_extractor.conv_layers.0.2.bias"]
    names += [p + "layer_norm.weight", p + "layer_norm.bias", p + "post_extract_proj.weight", p + "post_extract_proj.bias"]
    names += [p + "pos_conv.0.bias", p + "pos_conv.0.weight_g", p + "pos_conv.0.weight_v"]
    names += [p + "pos_sinusoidal_embed._float_tensor"]  # fairseq SinusoidalPositionalEmbedding buffer: no rule in Mapping
    names += ["text_encoder_prenet.encoder_prenet.0.weight", "text_encoder_prenet.encoder_prenet.1.alpha"]
    e = "encoder."
    names += [e + "version"]  # fairseq TransformerEncoder buffer: no rule in Mapping
    for l in range(LAYERS):
        b = f"{e}layers.{l}."
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            names += [b + f"self_attn.{proj}.weight", b + f"self_attn.{proj}.bias"]
        names += [b + "self_attn_layer_norm.weight", b + "self_attn_layer_norm.bias"]
        names += [b + "fc1.weight", b + "fc1.bias", b + "fc2.weight", b + "fc2.bias"]
        names += [b + "final_layer_norm.weight", b + "final_layer_norm.bias"]
    names += [e + "layer_norm.weight", e + "layer_norm.bias", e + "pos_emb.pe_k.weight"]
    # other sub-modules (ignored by Mapping)
    names += ["decoder.layers.0.fc1.weight", "decoder.layers.0.self_attn.q_proj.weight", "speech_decoder_postnet.feat_out.weight",
              "text_decoder_postnet.output_projection.weight"]
    return names


class RenamedView:
    """``named_parameters()`` of `model` with the two weight-norm names spelled as transformers 4.30.2 spells them."""

    def __init__(self, model):
        self._m = model
        self.speecht5 = model.speecht5

    def named_parameters(self):
        for n, p in self._m.named_parameters():
            yield (n.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v"), p)


def run(mapping_cls, model_asr, model_tts, ckpt):
    m = mapping_cls(model_asr, model_tts, ckpt)
    names = {int(v): k for k, v in ckpt["model"].items()}

    def table(sd):
        out = {}
        for k, v in sd.items():
            if torch.is_tensor(v) and v.numel() == 1 and v.dtype == torch.int64 and int(v) in names:
                out[k] = names[int(v)]  # taken from the checkpoint
            else:
                out[k] = "<model parameter, not from the checkpoint>"
        return out

    return {"encoder_state_dict": table(m.encoder_state_dict),
            "speech_prenet_state_dict": table(m.speech_prenet_state_dict),
            "text_prenet_state_dict": table(m.text_prenet_state_dict),
            "encoder_unmatched": sorted(k for k, v in m.encoder_values.items() if v is None)}


def main():
    from transformers import SpeechT5Config, SpeechT5ForSpeechToText, SpeechT5ForTextToSpeech
    import transformers

    spec = importlib.util.spec_from_file_location("ref_map_speecht5_hf", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    ckpt = {"model": {n: torch.tensor(1000 + i, dtype=torch.int64) for i, n in enumerate(fairseq_names())}}
    cfg = SpeechT5Config()
    asr, tts = SpeechT5ForSpeechToText(cfg), SpeechT5ForTextToSpeech(cfg)
    out = {"transformers_installed": transformers.__version__, "fairseq_keys": fairseq_names(),
           "hf_installed": run(ref.Mapping, asr, tts, ckpt),
           "hf_4_30_2": run(ref.Mapping, RenamedView(asr), RenamedView(tts), ckpt)}
    path = os.path.join(HERE, "g9_mapping.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")
    for k in ("hf_installed", "hf_4_30_2"):
        print(k, {n: len(v) for n, v in out[k].items()})


if __name__ == "__main__":
    main()

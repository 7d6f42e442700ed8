"""Weight import: fairseq SpeechT5-base checkpoint keys -> the three HuggingFace-named state dicts the encoders load.

Restates what the reference's `Mapping` class produces (/root/reference/speech_text/map_speecht5_hf.py:34-99 encoder,
:101-166 speech prenet, :168-181 text prenet) as explicit rename tables instead of its nested string searches.  The tables
are pinned by tests/golden/g9_mapping.json, which tests/golden/make_mapping_goldens.py wrote by running THE REFERENCE'S
`Mapping` itself on a synthetic fairseq-named checkpoint and locally instantiated HF models (tests/test_host_logic.py).
The fairseq names are

    encoder.pos_emb.pe_k.weight                                   -> embed_positions.pe_k.weight
    encoder.layer_norm.{weight,bias}                              -> layer_norm.*
    encoder.layers.N.self_attn.{q,k,v,out}_proj.{weight,bias}     -> layers.N.attention.*
    encoder.layers.N.self_attn_layer_norm.*                       -> layers.N.layer_norm.*
    encoder.layers.N.fc1.* / fc2.*                                -> layers.N.feed_forward.intermediate_dense.* / output_dense.*
    encoder.layers.N.final_layer_norm.*                           -> layers.N.final_layer_norm.*
    speech_encoder_prenet.mask_emb                                -> masked_spec_embed
    speech_encoder_prenet.layer_norm.*                            -> feature_projection.layer_norm.*
    speech_encoder_prenet.post_extract_proj.*                     -> feature_projection.projection.*
    speech_encoder_prenet.feature_extractor.conv_layers.N.0.weight-> feature_encoder.conv_layers.N.conv.weight
    speech_encoder_prenet.feature_extractor.conv_layers.N.2.*     -> feature_encoder.conv_layers.N.layer_norm.*
    speech_encoder_prenet.pos_conv.0.{bias,weight_g,weight_v}     -> pos_conv_embed.conv.{bias,weight_g,weight_v}

    text_encoder_prenet.encoder_prenet.0.weight                   -> embed_tokens.weight            (text prenet dict)

(`weight_g/weight_v` is the transformers-4.30.2 spelling, the reference's pin; the encoder's load_state_dict accepts it and
the 5.x `parametrizations.weight.original0/1` spelling alike.  Run against transformers 5.x the reference's own rule for
those two keys finds no partner and silently drops them -- recorded in the fixture's "hf_installed" table.)
"""
from __future__ import annotations

import re
from typing import Dict, Tuple

_ENC_RULES = (
    (r"^encoder\.pos_emb\.(.+)$", r"embed_positions.\1"),
    (r"^encoder\.layer_norm\.(weight|bias)$", r"layer_norm.\1"),
    (r"^encoder\.layers\.(\d+)\.self_attn\.(q_proj|k_proj|v_proj|out_proj)\.(weight|bias)$", r"layers.\1.attention.\2.\3"),
    (r"^encoder\.layers\.(\d+)\.self_attn_layer_norm\.(weight|bias)$", r"layers.\1.layer_norm.\2"),
    (r"^encoder\.layers\.(\d+)\.fc1\.(weight|bias)$", r"layers.\1.feed_forward.intermediate_dense.\2"),
    (r"^encoder\.layers\.(\d+)\.fc2\.(weight|bias)$", r"layers.\1.feed_forward.output_dense.\2"),
    (r"^encoder\.layers\.(\d+)\.final_layer_norm\.(weight|bias)$", r"layers.\1.final_layer_norm.\2"),
)
_PRE_RULES = (
    (r"^speech_encoder_prenet\.mask_emb$", r"masked_spec_embed"),
    (r"^speech_encoder_prenet\.layer_norm\.(weight|bias)$", r"feature_projection.layer_norm.\1"),
    (r"^speech_encoder_prenet\.post_extract_proj\.(weight|bias)$", r"feature_projection.projection.\1"),
    (r"^speech_encoder_prenet\.feature_extractor\.conv_layers\.(\d+)\.0\.weight$", r"feature_encoder.conv_layers.\1.conv.weight"),
    (r"^speech_encoder_prenet\.feature_extractor\.conv_layers\.(\d+)\.2\.(weight|bias)$", r"feature_encoder.conv_layers.\1.layer_norm.\2"),
    (r"^speech_encoder_prenet\.pos_conv\.0\.(bias|weight_g|weight_v)$", r"pos_conv_embed.conv.\1"),
)


def _apply(rules, key):
    for pat, rep in rules:
        if re.match(pat, key):
            return re.sub(pat, rep, key)
    return None


def map_fairseq_speecht5(ckpt_model: Dict[str, object]) -> Tuple[Dict[str, object], Dict[str, object], list]:
    """(encoder_state_dict, speech_prenet_state_dict, unmapped_keys) from ``ckpt['model']`` of a fairseq
    SpeechT5 checkpoint.  Keys of other sub-modules (decoder, text pre/post-nets, quantizer ...) are ignored;
    speech-path keys that match no rule are returned in ``unmapped_keys`` so that nothing is dropped silently
    (the reference records them as ``encoder_mapping[name] = None``, map_speecht5_hf.py:68,75)."""
    enc, pre, unmapped = {}, {}, []
    for k, v in ckpt_model.items():
        if k.startswith("encoder."):
            nk = _apply(_ENC_RULES, k)
            if nk is None:
                unmapped.append(k)
            else:
                enc[nk] = v
        elif k.startswith("speech_encoder_prenet."):
            nk = _apply(_PRE_RULES, k)
            if nk is None:
                unmapped.append(k)
            else:
                pre[nk] = v
    return enc, pre, unmapped


def map_text_prenet(ckpt_model: Dict[str, object], tts_prenet_state: Dict[str, object] | None = None) -> Dict[str, object]:
    """The reference's third dict (map_speecht5_hf.py:168-181): ``embed_tokens.weight`` is taken from the fairseq checkpoint
    (``text_encoder_prenet.encoder_prenet.0.weight``); every other entry -- ``encode_positions.alpha`` and the sinusoid
    buffer ``encode_positions.pe`` -- is copied from the HF TTS model's text prenet (``tts_prenet_state`` =
    ``model_tts.speecht5.encoder.prenet.state_dict()`` plus its ``encode_positions.pe`` buffer), NOT from the checkpoint.
    Without an HF model at hand (``tts_prenet_state=None``) alpha falls back to the fairseq value
    ``text_encoder_prenet.encoder_prenet.1.alpha`` when present and ``pe`` is left out (the library generates the table)."""
    out = {"embed_tokens.weight": ckpt_model["text_encoder_prenet.encoder_prenet.0.weight"]}
    if tts_prenet_state is not None:
        for k, v in tts_prenet_state.items():
            if k != "embed_tokens.weight":
                out[k] = v
    elif "text_encoder_prenet.encoder_prenet.1.alpha" in ckpt_model:
        out["encode_positions.alpha"] = ckpt_model["text_encoder_prenet.encoder_prenet.1.alpha"]
    return out


def to_fairseq_names(prenet_sd: Dict[str, object], encoder_sd: Dict[str, object]) -> Dict[str, object]:
    """Inverse rename (HF -> fairseq), used to build test checkpoints."""
    out = {}
    for k, v in encoder_sd.items():
        k = re.sub(r"^embed_positions\.", "pos_emb.", k)
        k = re.sub(r"\.attention\.", ".self_attn.", k)
        k = re.sub(r"^(layers\.\d+)\.layer_norm\.", r"\1.self_attn_layer_norm.", k)
        k = re.sub(r"\.feed_forward\.intermediate_dense\.", ".fc1.", k)
        k = re.sub(r"\.feed_forward\.output_dense\.", ".fc2.", k)
        out["encoder." + k] = v
    for k, v in prenet_sd.items():
        k = re.sub(r"^masked_spec_embed$", "mask_emb", k)
        k = re.sub(r"^feature_projection\.layer_norm\.", "layer_norm.", k)
        k = re.sub(r"^feature_projection\.projection\.", "post_extract_proj.", k)
        k = re.sub(r"^feature_encoder\.conv_layers\.(\d+)\.conv\.weight$", r"feature_extractor.conv_layers.\1.0.weight", k)
        k = re.sub(r"^feature_encoder\.conv_layers\.(\d+)\.layer_norm\.", r"feature_extractor.conv_layers.\1.2.", k)
        k = re.sub(r"^pos_conv_embed\.conv\.parametrizations\.weight\.original0$", "pos_conv.0.weight_g", k)
        k = re.sub(r"^pos_conv_embed\.conv\.parametrizations\.weight\.original1$", "pos_conv.0.weight_v", k)
        k = re.sub(r"^pos_conv_embed\.conv\.(bias|weight_g|weight_v)$", r"pos_conv.0.\1", k)
        out["speech_encoder_prenet." + k] = v
    return out

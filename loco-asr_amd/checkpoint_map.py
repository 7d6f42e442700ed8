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


# ---- checkpoints on disk ----------------------------------------------------------------------------------------------------
# The two callers the reference has:
#   * fine-tuned: SpeechT5ForSpeechToText.from_pretrained("microsoft/speecht5_asr")       (…finetuned…py:95)
#       -> load_hf_checkpoint(dir) + SpeechT5ForSpeechToTextMI355X.from_pretrained(dir) (encoder.py)
#   * base: three pickles under extracted/speecht5/mapping/ made from a fairseq speecht5_base.pt (…base…py:40-49,
#     map_speecht5_hf.py:157-181)  ->  `python -m loco-asr_amd.checkpoint_map speecht5_base.pt --out extracted/speecht5/mapping/`
HF_PREFIXES = ("speecht5.encoder.prenet.", "speecht5.encoder.wrapped_encoder.")


def _checkpoint_files(path: str):
    """Weight files of a HuggingFace checkpoint directory (or the file itself): model.safetensors, a sharded
    model.safetensors.index.json / pytorch_model.bin.index.json, or pytorch_model.bin -- in that order, as transformers looks."""
    import json
    import os
    if os.path.isfile(path):
        return [path]
    if not os.path.isdir(path):
        # a hub NAME: only its local cache can serve it (there is no network on the build / GPU boxes, and none is attempted)
        cache = os.environ.get("HF_HUB_CACHE") or os.path.join(os.environ.get("HF_HOME", os.path.join(os.path.expanduser("~"), ".cache", "huggingface")), "hub")
        snaps = os.path.join(cache, "models--" + path.replace("/", "--"), "snapshots")
        if os.path.isdir(snaps) and os.listdir(snaps):
            return _checkpoint_files(os.path.join(snaps, sorted(os.listdir(snaps))[-1]))
        raise FileNotFoundError(f"{path!r} is neither a checkpoint directory / file nor a model in the local HuggingFace cache ({snaps}); "
                                "this loader never downloads: pass the directory that holds model.safetensors or pytorch_model.bin")
    for single in ("model.safetensors", ):
        if os.path.exists(os.path.join(path, single)):
            return [os.path.join(path, single)]
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        ip = os.path.join(path, index)
        if os.path.exists(ip):
            with open(ip) as fh:
                shards = sorted(set(json.load(fh)["weight_map"].values()))
            return [os.path.join(path, sname) for sname in shards]
    if os.path.exists(os.path.join(path, "pytorch_model.bin")):
        return [os.path.join(path, "pytorch_model.bin")]
    raise FileNotFoundError(f"{path}: no model.safetensors, pytorch_model.bin or sharded index found")


def load_hf_checkpoint(path: str, with_text_prenet: bool = False):
    """(prenet_state_dict, encoder_state_dict) -- the arguments of the two load_state_dict calls -- read from a HuggingFace
    SpeechT5 checkpoint on disk (SpeechT5ForSpeechToText / SpeechT5Model layout: keys ``speecht5.encoder.prenet.*`` and
    ``speecht5.encoder.wrapped_encoder.*``; a bare SpeechT5EncoderWithSpeechPrenet state dict, ``prenet.*`` /
    ``wrapped_encoder.*``, is accepted too).  Only those tensors are read from a .safetensors file (the ASR checkpoint's decoder
    and text post-net, 60 % of its bytes, are never touched).  Both spellings of the weight-normed positional conv are kept as they
    are (the module's load_state_dict accepts either); nothing is validated here -- ``load_state_dict(strict=True)`` names what is
    missing or unexpected."""
    import torch
    pre, enc = {}, {}

    def take(key, get):
        for prefixes, dst in (((HF_PREFIXES[0], "prenet."), pre), ((HF_PREFIXES[1], "wrapped_encoder."), enc)):
            for pf in prefixes:
                if key.startswith(pf):
                    dst[key[len(pf):]] = get()
                    return

    for f in _checkpoint_files(path):
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(f, framework="pt", device="cpu") as sf:
                for key in sf.keys():
                    take(key, lambda k_=key: sf.get_tensor(k_))
        else:
            sd = torch.load(f, map_location="cpu", weights_only=True)
            sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
            for key, v in sd.items():
                take(key, lambda v_=v: v_)
    if not pre and not enc:
        raise KeyError(f"{path}: no tensor named speecht5.encoder.prenet.* / speecht5.encoder.wrapped_encoder.* -- not a SpeechT5 speech-to-text checkpoint")
    return pre, enc


def check_hf_config(path: str):
    """config.json next to the weights, when there is one: the kernels are specialised for SpeechT5-base; say so before loading."""
    import json
    import os
    cfg = os.path.join(path, "config.json") if os.path.isdir(path) else None
    if not cfg or not os.path.exists(cfg):
        return None
    with open(cfg) as fh:
        c = json.load(fh)
    want = dict(hidden_size=768, encoder_attention_heads=12, encoder_ffn_dim=3072, num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16,
                encoder_max_relative_position=160, conv_dim=[512] * 7, conv_kernel=[10, 3, 3, 3, 3, 2, 2], conv_stride=[5, 2, 2, 2, 2, 2, 2],
                feat_extract_norm="group", hidden_act="gelu")
    bad = {k: c[k] for k, v in want.items() if k in c and c[k] != v}
    if bad:
        raise ValueError(f"{cfg}: not the SpeechT5-base geometry this library is built for: {bad} (expected {({k: want[k] for k in bad})})")
    return c


def load_fairseq_checkpoint(path: str, unsafe: bool = False):
    """``ckpt['model']`` of a fairseq SpeechT5 checkpoint (speecht5_base.pt).  Such files also pickle their training arguments
    (argparse.Namespace / omegaconf objects); the tensors-only unpickler is tried first, with Namespace allowed; ``unsafe=True``
    (the CLI's --trust-checkpoint) falls back to the full pickle for a file you trust."""
    import argparse
    import torch
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:  # noqa: BLE001
        if not unsafe:
            raise RuntimeError(f"{path}: the tensors-only unpickler refused this checkpoint ({type(e).__name__}: {str(e)[:200]}); if you trust "
                               "the file, pass --trust-checkpoint (full pickle)") from e
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    model = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    if not isinstance(model, dict) or not any(k.startswith("encoder.") for k in model):
        raise KeyError(f"{path}: no fairseq SpeechT5 'model' state dict (keys encoder.*, speech_encoder_prenet.*) found")
    return model


def write_mapping_pickles(ckpt_model, out_dir: str, tts_prenet_state=None):
    """The three files the base script opens (…base…py:40-49): encoder_state_dict.pickle, speech_prenet_state_dict.pickle,
    text_prenet_state_dict.pickle -- torch tensors under HF key names, as the reference's Mapping stores them
    (map_speecht5_hf.py:157-181; its speech-prenet dict also carries HF's sinusoid buffer, regenerated here bit for bit).  Raises when a
    speech-path key of the checkpoint matches no rule (nothing is dropped silently)."""
    import os
    import pickle
    import torch
    enc, pre, unmapped = map_fairseq_speecht5(ckpt_model)
    if unmapped:
        raise KeyError(f"fairseq keys without an HF counterpart: {unmapped[:8]}{' ...' if len(unmapped) > 8 else ''}")
    as_t = lambda d: {k: (v if torch.is_tensor(v) else torch.as_tensor(v)) for k, v in d.items()}  # noqa: E731
    enc, pre = as_t(enc), as_t(pre)
    from .encoder import MAX_SPEECH_POSITIONS, PAD_TOKEN_ID, sinusoid_table
    pre["pos_sinusoidal_embed.weights"] = sinusoid_table(MAX_SPEECH_POSITIONS + PAD_TOKEN_ID + 1 + 2)  # HF's 4 004 rows (modeling:296-303)
    os.makedirs(out_dir, exist_ok=True)
    files = {"encoder_state_dict.pickle": enc, "speech_prenet_state_dict.pickle": pre}
    if "text_encoder_prenet.encoder_prenet.0.weight" in ckpt_model:
        files["text_prenet_state_dict.pickle"] = as_t(map_text_prenet(ckpt_model, tts_prenet_state))
    for name, d in files.items():
        with open(os.path.join(out_dir, name), "wb") as fh:
            pickle.dump(d, fh, protocol=pickle.HIGHEST_PROTOCOL)
    return {name: len(d) for name, d in files.items()}


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m loco-asr_amd.checkpoint_map",
                                 description="fairseq speecht5_base.pt -> the three pickled state dicts extract.py (and the reference's base script) load")
    ap.add_argument("checkpoint", help="fairseq SpeechT5 checkpoint (.pt)")
    ap.add_argument("--out", default="extracted/speecht5/mapping/", help="directory of the three pickles (the reference's path)")
    ap.add_argument("--tts-dir", default=None,
                    help="HuggingFace SpeechT5 TTS checkpoint directory: its text prenet supplies encode_positions.alpha (the reference copies "
                         "every text-prenet entry except the embedding from the HF TTS model, map_speecht5_hf.py:168-181)")
    ap.add_argument("--trust-checkpoint", action="store_true", help="allow the full pickle when the tensors-only unpickler refuses the file")
    args = ap.parse_args(argv)
    model = load_fairseq_checkpoint(args.checkpoint, unsafe=args.trust_checkpoint)
    tts = None
    if args.tts_dir:
        pre_tts, _ = load_hf_checkpoint(args.tts_dir)
        tts = {k: v for k, v in pre_tts.items() if k.startswith(("embed_tokens.", "encode_positions."))}
    counts = write_mapping_pickles(model, args.out, tts)
    for name, n in counts.items():
        print(f"wrote {args.out.rstrip('/')}/{name}: {n} tensors")
    return counts


if __name__ == "__main__":
    main()

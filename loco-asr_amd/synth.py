"""Deterministic synthetic weights and 16 kHz audio for tests, goldens and bench.

No checkpoint or corpus is reachable from this environment, so every
parity fixture and every bench run uses

* weights from an integer-hash generator (splitmix64 finaliser over ``hash(key) + index``)
  -- reproducible bit for bit on any machine, independent of ``torch.manual_seed`` streams,
  so the 378 MB of fp32 encoder parameters never have to be committed;
* audio as SURVEY.md §8(d) prescribes: per clip ``0.1*N(0,1) + 0.05*(sin 220 + sin 440 +
  sin 1760 Hz)`` from ``numpy.random.Generator(Philox(1234 + clip))``.

Key names are the HuggingFace ``SpeechT5EncoderWithSpeechPrenet.state_dict()`` names that the
reference's callers load through ``encoder.prenet.load_state_dict`` /
``encoder.wrapped_encoder.load_state_dict``
(/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:99-100).
"""
from __future__ import annotations

import math

import numpy as np

SAMPLE_RATE = 16000

CONV_DIM = 512
CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
HIDDEN = 768
HEADS = 12
HEAD_DIM = 64
FFN = 3072
LAYERS = 12
POS_CONV_K = 128
POS_CONV_GROUPS = 16
REL_MAX = 160

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(s: str) -> int:
    h = 0xCBF29CE484222325
    for ch in s.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def hashed_uniform(key: str, shape, seed: int = 0) -> np.ndarray:
    """float32 array, uniform in [-1, 1), a pure function of (key, seed, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a(key) ^ (seed * 0xD6E8FEB86659FD93)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + base
    bits = _splitmix64(idx) >> np.uint64(40)  # 24 random bits
    u = bits.astype(np.float32) * np.float32(1.0 / (1 << 24))  # [0,1)
    return (u * np.float32(2.0) - np.float32(1.0)).reshape(shape)


def _w(key, shape, std, seed, mean=0.0):
    # uniform with the requested standard deviation (half-width = std*sqrt(3))
    a = hashed_uniform(key, shape, seed) * np.float32(std * math.sqrt(3.0))
    if mean:
        a = a + np.float32(mean)
    return a.astype(np.float32)


def encoder_state_dict(seed: int = 0, layers: int = LAYERS) -> dict[str, np.ndarray]:
    """numpy fp32 state dict with HF key names (prefix ``prenet.`` / ``wrapped_encoder.``).

    Scales are chosen so that every stage is numerically non-trivial: attention logits have
    a standard deviation of ~2 and the relative-position bias ~0.7 (a near-uniform softmax
    would hide indexing bugs), GroupNorm/LayerNorm affines are not the identity, the
    weight-norm gain differs from ||v||.
    """
    sd: dict[str, np.ndarray] = {}
    p = "prenet."
    sd[p + "masked_spec_embed"] = (_w(p + "masked_spec_embed", (HIDDEN,), 1 / math.sqrt(3), seed) + 1) / 2
    for i, k in enumerate(CONV_KERNEL):
        cin = 1 if i == 0 else CONV_DIM
        name = f"{p}feature_encoder.conv_layers.{i}.conv.weight"
        std = 0.3 if i == 0 else math.sqrt(2.0 / (cin * k))
        sd[name] = _w(name, (CONV_DIM, cin, k), std, seed)
    n = p + "feature_encoder.conv_layers.0.layer_norm."
    sd[n + "weight"] = _w(n + "weight", (CONV_DIM,), 0.1, seed, 1.0)
    sd[n + "bias"] = _w(n + "bias", (CONV_DIM,), 0.1, seed)
    n = p + "feature_projection.layer_norm."
    sd[n + "weight"] = _w(n + "weight", (CONV_DIM,), 0.1, seed, 1.0)
    sd[n + "bias"] = _w(n + "bias", (CONV_DIM,), 0.1, seed)
    n = p + "feature_projection.projection."
    sd[n + "weight"] = _w(n + "weight", (HIDDEN, CONV_DIM), 1 / math.sqrt(CONV_DIM), seed)
    sd[n + "bias"] = _w(n + "bias", (HIDDEN,), 0.02, seed)
    n = p + "pos_conv_embed.conv."
    cg = HIDDEN // POS_CONV_GROUPS
    v = _w(n + "parametrizations.weight.original1", (HIDDEN, cg, POS_CONV_K), 2 / math.sqrt(cg * POS_CONV_K), seed)
    vnorm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True)).astype(np.float32)
    g = vnorm * (1 + 0.2 * hashed_uniform(n + "parametrizations.weight.original0", (1, 1, POS_CONV_K), seed))
    sd[n + "bias"] = _w(n + "bias", (HIDDEN,), 0.02, seed)
    sd[n + "parametrizations.weight.original0"] = g.astype(np.float32)
    sd[n + "parametrizations.weight.original1"] = v

    e = "wrapped_encoder."
    sd[e + "layer_norm.weight"] = _w(e + "layer_norm.weight", (HIDDEN,), 0.1, seed, 1.0)
    sd[e + "layer_norm.bias"] = _w(e + "layer_norm.bias", (HIDDEN,), 0.1, seed)
    sd[e + "embed_positions.pe_k.weight"] = _w(e + "embed_positions.pe_k.weight", (2 * REL_MAX, HEAD_DIM), 0.5, seed)
    for l in range(layers):
        b = f"{e}layers.{l}."
        for proj, std in (("q_proj", 1.5), ("k_proj", 1.5), ("v_proj", 1.0), ("out_proj", 1.0)):
            sd[f"{b}attention.{proj}.weight"] = _w(f"{b}attention.{proj}.weight", (HIDDEN, HIDDEN), std / math.sqrt(HIDDEN), seed)
            sd[f"{b}attention.{proj}.bias"] = _w(f"{b}attention.{proj}.bias", (HIDDEN,), 0.02, seed)
        for ln in ("layer_norm", "final_layer_norm"):
            sd[f"{b}{ln}.weight"] = _w(f"{b}{ln}.weight", (HIDDEN,), 0.1, seed, 1.0)
            sd[f"{b}{ln}.bias"] = _w(f"{b}{ln}.bias", (HIDDEN,), 0.1, seed)
        sd[f"{b}feed_forward.intermediate_dense.weight"] = _w(f"{b}feed_forward.intermediate_dense.weight", (FFN, HIDDEN), 1.2 / math.sqrt(HIDDEN), seed)
        sd[f"{b}feed_forward.intermediate_dense.bias"] = _w(f"{b}feed_forward.intermediate_dense.bias", (FFN,), 0.02, seed)
        sd[f"{b}feed_forward.output_dense.weight"] = _w(f"{b}feed_forward.output_dense.weight", (HIDDEN, FFN), 1 / math.sqrt(FFN), seed)
        sd[f"{b}feed_forward.output_dense.bias"] = _w(f"{b}feed_forward.output_dense.bias", (HIDDEN,), 0.02, seed)
    return sd


def hashed_normal(key: str, shape, seed: int = 0) -> np.ndarray:
    """float32 standard-normal array, a pure function of (key, seed, flat index): Box-Muller on two hashed uniforms."""
    u1 = (hashed_uniform(key + "/u1", shape, seed).astype(np.float64) + 1.0) * 0.5
    u2 = (hashed_uniform(key + "/u2", shape, seed).astype(np.float64) + 1.0) * 0.5
    u1 = np.maximum(u1, 2.0 ** -25)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def encoder_state_dict_hf_init(seed: int = 0, layers: int = LAYERS) -> dict[str, np.ndarray]:
    """A SECOND weight family for the goldens (g10): HuggingFace's own initialisation distributions
    (SpeechT5PreTrainedModel._init_weights, modeling_speecht5.py: Linear / Embedding weights N(0, initializer_range = 0.02), conv
    layers kaiming-normal, the positional conv N(0, 2 sqrt(1 / (128 * 768))) with weight-norm g = ||v||, the feature projection
    U(+-1/sqrt(512))) drawn from the hash generator (independent of torch's RNG streams) -- plus what a TRAINED checkpoint has and
    a fresh init has not: log-normal LayerNorm / GroupNorm gains (sigma 0.5), non-zero biases, and a few x100 OUTLIER CHANNELS in
    every encoder LayerNorm gain and in the feed-forward bias (the massive-activation channels of real transformers).  The first
    family (encoder_state_dict) is uniform and tuned for sharp attention; this one is normal, near-uniform in attention and
    heavy-tailed in the residual stream."""
    sd: dict[str, np.ndarray] = {}
    N = lambda key, shape, std, mean=0.0: (hashed_normal(key, shape, seed + 1000) * np.float32(std) + np.float32(mean)).astype(np.float32)  # noqa: E731
    U = lambda key, shape, a: (hashed_uniform(key, shape, seed + 1000) * np.float32(a)).astype(np.float32)  # noqa: E731

    def gain(key, n, outliers=0):
        g = np.exp(N(key, (n,), 0.5).astype(np.float64))
        for j in range(outliers):  # deterministic outlier channels, x100
            g[(_fnv1a(f"{key}/outlier{j}") + seed) % n] *= 100.0
        return g.astype(np.float32)

    p = "prenet."
    sd[p + "masked_spec_embed"] = ((U(p + "masked_spec_embed", (HIDDEN,), 1.0) + 1) / 2).astype(np.float32)
    for i, k in enumerate(CONV_KERNEL):
        cin = 1 if i == 0 else CONV_DIM
        name = f"{p}feature_encoder.conv_layers.{i}.conv.weight"
        sd[name] = N(name, (CONV_DIM, cin, k), math.sqrt(2.0 / (cin * k)))  # kaiming_normal_, fan_in
    n = p + "feature_encoder.conv_layers.0.layer_norm."
    sd[n + "weight"] = gain(n + "weight", CONV_DIM)
    sd[n + "bias"] = N(n + "bias", (CONV_DIM,), 0.1)
    n = p + "feature_projection.layer_norm."
    sd[n + "weight"] = gain(n + "weight", CONV_DIM)
    sd[n + "bias"] = N(n + "bias", (CONV_DIM,), 0.1)
    n = p + "feature_projection.projection."
    sd[n + "weight"] = U(n + "weight", (HIDDEN, CONV_DIM), 1 / math.sqrt(CONV_DIM))
    sd[n + "bias"] = U(n + "bias", (HIDDEN,), 1 / math.sqrt(CONV_DIM))
    n = p + "pos_conv_embed.conv."
    cg = HIDDEN // POS_CONV_GROUPS
    v = N(n + "parametrizations.weight.original1", (HIDDEN, cg, POS_CONV_K), 2 * math.sqrt(1.0 / (POS_CONV_K * HIDDEN)))
    sd[n + "bias"] = N(n + "bias", (HIDDEN,), 0.01)
    sd[n + "parametrizations.weight.original0"] = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True)).astype(np.float32)
    sd[n + "parametrizations.weight.original1"] = v
    e = "wrapped_encoder."
    sd[e + "layer_norm.weight"] = gain(e + "layer_norm.weight", HIDDEN, outliers=2)
    sd[e + "layer_norm.bias"] = N(e + "layer_norm.bias", (HIDDEN,), 0.1)
    sd[e + "embed_positions.pe_k.weight"] = N(e + "embed_positions.pe_k.weight", (2 * REL_MAX, HEAD_DIM), 0.02)
    for l in range(layers):
        b = f"{e}layers.{l}."
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[f"{b}attention.{proj}.weight"] = N(f"{b}attention.{proj}.weight", (HIDDEN, HIDDEN), 0.02)
            sd[f"{b}attention.{proj}.bias"] = N(f"{b}attention.{proj}.bias", (HIDDEN,), 0.02)
        for ln in ("layer_norm", "final_layer_norm"):
            sd[f"{b}{ln}.weight"] = gain(f"{b}{ln}.weight", HIDDEN, outliers=2)
            sd[f"{b}{ln}.bias"] = N(f"{b}{ln}.bias", (HIDDEN,), 0.1)
        sd[f"{b}feed_forward.intermediate_dense.weight"] = N(f"{b}feed_forward.intermediate_dense.weight", (FFN, HIDDEN), 0.02)
        fb = N(f"{b}feed_forward.intermediate_dense.bias", (FFN,), 0.02)
        for j in range(3):
            fb[(_fnv1a(f"{b}ffn_bias/outlier{j}") + seed) % FFN] = np.float32(2.0 + j)
        sd[f"{b}feed_forward.intermediate_dense.bias"] = fb
        sd[f"{b}feed_forward.output_dense.weight"] = N(f"{b}feed_forward.output_dense.weight", (HIDDEN, FFN), 0.02)
        sd[f"{b}feed_forward.output_dense.bias"] = N(f"{b}feed_forward.output_dense.bias", (HIDDEN,), 0.02)
    return sd


TEXT_VOCAB = 81          # SpeechT5Config.vocab_size
MAX_TEXT_POSITIONS = 450  # SpeechT5Config.max_text_positions


def text_prenet_state_dict(seed: int = 0) -> dict[str, np.ndarray]:
    """Weights of SpeechT5TextEncoderPrenet under the prefix ``text_prenet.``: a non-trivial embedding table and a
    positional scale alpha != 1 (HF initialises it to 1.0; a trained checkpoint does not keep it there)."""
    p = "text_prenet."
    return {p + "embed_tokens.weight": _w(p + "embed_tokens.weight", (TEXT_VOCAB, HIDDEN), 0.6, seed),
            p + "encode_positions.alpha": np.asarray(1.37, dtype=np.float32)}


def token_ids(batch: int, tokens: int, seed: int = 7, lengths=None):
    """Deterministic synthetic token ids [batch, tokens] in [4, TEXT_VOCAB) (0..3 are the special tokens; 1 = <pad>) and the
    int32 right-padding mask for ``lengths`` (pad id 1 beyond each length), mirroring ``processor(text=..., padding="longest")``."""
    u = hashed_uniform(f"token_ids/{batch}x{tokens}", (batch, tokens), seed)
    ids = (4 + np.floor((u + 1.0) * 0.5 * (TEXT_VOCAB - 4))).astype(np.int64).clip(4, TEXT_VOCAB - 1)
    mask = np.ones((batch, tokens), dtype=np.int32)
    if lengths is not None:
        for i, n in enumerate(lengths):
            ids[i, n:] = 1
            mask[i, n:] = 0
    return ids, mask


def split_state_dict(sd: dict):
    """(prenet_sd, wrapped_encoder_sd) with the prefixes stripped -- the two dicts the
    reference loads separately (extract_speecht5_base_embeddings_slurp.py:99-100)."""
    pre = {k[len("prenet."):]: v for k, v in sd.items() if k.startswith("prenet.")}
    enc = {k[len("wrapped_encoder."):]: v for k, v in sd.items() if k.startswith("wrapped_encoder.")}
    return pre, enc


def clip(index: int, num_samples: int) -> np.ndarray:
    """One synthetic 16 kHz clip, float32 [num_samples]."""
    rng = np.random.Generator(np.random.Philox(1234 + index))
    x = 0.1 * rng.standard_normal(num_samples, dtype=np.float32)
    t = np.arange(num_samples, dtype=np.float64) / SAMPLE_RATE
    tones = sum(np.sin(2 * np.pi * f * t) for f in (220.0, 440.0, 1760.0))
    return (x + (0.05 * tones).astype(np.float32)).astype(np.float32)


def batch(lengths, first_index: int = 0):
    """Pad-to-longest batch as the HF feature extractor would hand it over:
    (input_values f32 [B, Lmax], attention_mask i32 [B, Lmax])."""
    lengths = [int(v) for v in lengths]
    lmax = max(lengths)
    x = np.zeros((len(lengths), lmax), np.float32)
    m = np.zeros((len(lengths), lmax), np.int32)
    for i, n in enumerate(lengths):
        x[i, :n] = clip(first_index + i, n)
        m[i, :n] = 1
    return x, m


def mixed_lengths(n_clips: int, max_samples: int, seed: int = 99, min_fraction: float = 0.5):
    """Lengths drawn U[min_fraction, 1]*max_samples (SURVEY.md §8d mixed-length variant: min_fraction 0.5)."""
    rng = np.random.Generator(np.random.Philox(seed))
    return [int(max_samples * (min_fraction + (1.0 - min_fraction) * u)) for u in rng.random(n_clips)]


def conv_out_length(n: int) -> int:
    """Frames produced from n samples: floor((n-k)/s)+1 chained over the 7 conv layers
    (HF modeling_speecht5.py:585-598)."""
    for k, s in zip(CONV_KERNEL, CONV_STRIDE):
        n = (n - k) // s + 1
    return n


# ---- intent head ("next" row f-1): inputs of tests/golden/make_head_goldens.py, rebuilt bit for bit by the tests -----------
def head_params(method: str):
    """Initial (q [1,768], W [101,768], b [101]) of IntentClassifier for the head goldens -- q large enough that the
    attention weights are far from uniform (the reference initialises it at 1e-3 scale, intent_classifier.py:17)."""
    q = hashed_uniform(f"head/{method}/q", (1, HIDDEN), 5) * np.float32(0.35)
    w = hashed_uniform(f"head/{method}/w", (101, HIDDEN), 5) * np.float32(1.0 / np.sqrt(float(HIDDEN)))
    b = hashed_uniform(f"head/{method}/b", (101,), 5) * np.float32(0.05)
    return q, w, b


def head_batch(B: int, T: int, tag: str):
    """Ragged zero-padded embeddings [B,T,768] as the reference's collate_fn builds them (pad_sequence,
    train_classifier.py:47-51; clip 0 has full length), one-hot int64 targets [B,101], and the lengths."""
    x = hashed_uniform(f"head/x/{tag}", (B, T, HIDDEN), 11) * np.float32(1.4)
    u = hashed_uniform(f"head/len/{tag}", (B,), 11)
    lens = np.clip((T * (0.35 + 0.65 * (u + 1) / 2)).astype(np.int64), 1, T)
    lens[0] = T
    for b in range(B):
        x[b, lens[b]:] = 0
    c = hashed_uniform(f"head/cls/{tag}", (B,), 11)
    cls = np.clip(((c + 1) / 2 * 101).astype(np.int64), 0, 100)
    return x, np.eye(101, dtype=np.int64)[cls], lens

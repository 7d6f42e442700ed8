"""CPU oracle for the SpeechT5 speech-encoder embedding path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``loco-asr_amd``) never does, and fails loudly when its HIP library is missing.

What it restates
----------------
The reference (`/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:108-109`)
calls ``model.speecht5.encoder(**audios).last_hidden_state``; the arithmetic lives in the
third-party dependency ``transformers`` (pinned ``==4.30.2`` in
`/root/reference/speech_text/requirements.txt:151`; 5.15.0 is what this image carries; same
maths for this path).  ``HF:`` line numbers below are for
``transformers/models/speecht5/modeling_speecht5.py`` 5.15.0.

The functions here are fresh code in plain torch CPU ops.  They differ from the HF module in
structure where HF cannot scale: the relative-position bias is evaluated in its compact form
``bias[i,j] = (q_i . pe_k^T)[clip(i-j,-160,159)+160]`` and attention runs over query blocks, so
10-minute inputs (T = 29 999; HF would need a 230 GB ``[T,T,64]`` table) stay feasible.

The text branch of the same scripts (`…base…py:79-93`: ``model.speecht5.encoder(texts.input_ids)``) is restated by
``text_prenet`` / ``encode_text`` (HF ``SpeechT5TextEncoderPrenet``, ``SpeechT5ScaledPositionalEncoding``,
``SpeechT5EncoderWithTextPrenet``) and pinned by fixture g6, generated from HF's own class.

Pinning: the reference has no tests, so parity is pinned by fixtures generated
in the build container from the HF implementation itself (``tests/golden/make_goldens.py``) and by
``tests/test_oracle_vs_hf.py`` which compares this file with the installed HF module directly
wherever ``transformers`` is importable.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
HEADS = 12
REL_MAX = 160
POS_CONV_K = 128
POS_CONV_GROUPS = 16
LN_EPS = 1e-5
PAD_IDX = 1  # SpeechT5Config.pad_token_id; sinusoid row used for padded frames


def _t(sd, key, dtype):
    v = sd[key]
    if not torch.is_tensor(v):
        v = torch.from_numpy(v)
    return v.to(dtype)


def _pos_conv_keys(sd, prefix):
    """weight-norm g, v under either spelling (4.30.2: weight_g/weight_v; 5.x: parametrizations)."""
    if prefix + "pos_conv_embed.conv.parametrizations.weight.original0" in sd:
        return (prefix + "pos_conv_embed.conv.parametrizations.weight.original0",
                prefix + "pos_conv_embed.conv.parametrizations.weight.original1")
    return prefix + "pos_conv_embed.conv.weight_g", prefix + "pos_conv_embed.conv.weight_v"


def gelu_erf(x):
    """exact GELU 0.5*x*(1+erf(x/sqrt(2)))  (transformers/activations.py:83-89)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def feat_extract_output_lengths(n):
    """HF:585-598 -- floor((n-k)/s)+1 over the seven conv layers; int or LongTensor."""
    for k, s in zip(CONV_KERNEL, CONV_STRIDE):
        n = torch.div(n - k, s, rounding_mode="floor") + 1 if torch.is_tensor(n) else (n - k) // s + 1
    return n


def feature_encoder(x, sd, prefix="prenet.", dtype=torch.float32):
    """HF:484-494, layer 0 = conv+GroupNorm(512 groups)+GELU (HF:260-281), layers 1-6 = conv+GELU
    (HF:210-228).  x [B, L] -> [B, T, 512] (already transposed to time-major)."""
    h = x.to(dtype)[:, None, :]
    for i, (k, s) in enumerate(zip(CONV_KERNEL, CONV_STRIDE)):
        w = _t(sd, f"{prefix}feature_encoder.conv_layers.{i}.conv.weight", dtype)
        h = F.conv1d(h, w, stride=s)
        if i == 0:
            # GroupNorm with one group per channel: statistics over the WHOLE (padded) time axis
            mean = h.mean(dim=2, keepdim=True)
            var = ((h - mean) ** 2).mean(dim=2, keepdim=True)
            gw = _t(sd, prefix + "feature_encoder.conv_layers.0.layer_norm.weight", dtype)[None, :, None]
            gb = _t(sd, prefix + "feature_encoder.conv_layers.0.layer_norm.bias", dtype)[None, :, None]
            h = (h - mean) * torch.rsqrt(var + LN_EPS) * gw + gb
        h = gelu_erf(h)
    return h.transpose(1, 2).contiguous()


def frame_counts(attention_mask, n_frames):
    """HF:569-582 -- valid frame count per clip from the sample mask's SUM; clamps are not applied
    by HF either (a count of 0 would index -1; callers never pass an all-zero mask)."""
    if attention_mask is None:
        return None
    lens = attention_mask.to(torch.long).sum(dim=-1)
    return feat_extract_output_lengths(lens)


def sinusoid_table(n_rows, dim=768, dtype=torch.float32):
    """HF:305-321 -- row p = [sin(p*w_k) | cos(p*w_k)], w_k = exp(-k*ln(1e4)/(dim/2-1)); row PAD_IDX zeroed.
    Always evaluated in fp32 like HF (the table is an fp32 buffer), then cast."""
    half = dim // 2
    w = torch.exp(torch.arange(half, dtype=torch.int64).float() * -(math.log(10000) / (half - 1)))
    ang = torch.arange(n_rows, dtype=torch.int64).float().unsqueeze(1) * w.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    tab[PAD_IDX] = 0
    return tab.to(dtype)


def pos_conv_weight(sd, prefix="prenet.", dtype=torch.float32):
    """weight_norm(dim=2): w[o,i,k] = g[k] * v[o,i,k] / ||v[:,:,k]||  (HF:358-379, torch weight_norm)."""
    kg, kv = _pos_conv_keys(sd, prefix)
    g = _t(sd, kg, dtype)
    v = _t(sd, kv, dtype)
    norm = torch.sqrt((v * v).sum(dim=(0, 1), keepdim=True))
    return v * (g / norm)


def speech_prenet(x, attention_mask, sd, prefix="prenet.", dtype=torch.float32, taps=None):
    """HF:534-566.  Returns (hidden [B,T,768], frames LongTensor[B] or None)."""
    feats = feature_encoder(x, sd, prefix, dtype)
    if taps is not None:
        taps["conv_stack"] = feats
    B, T, _ = feats.shape
    frames = frame_counts(attention_mask, T)
    # feature projection: LayerNorm(512) -> Linear(512,768)   (HF:498-510)
    h = F.layer_norm(feats, (feats.shape[-1],),
                     _t(sd, prefix + "feature_projection.layer_norm.weight", dtype),
                     _t(sd, prefix + "feature_projection.layer_norm.bias", dtype), LN_EPS)
    h = F.linear(h, _t(sd, prefix + "feature_projection.projection.weight", dtype),
                 _t(sd, prefix + "feature_projection.projection.bias", dtype))
    if taps is not None:
        taps["feature_projection"] = h
    # positional conv embedding: grouped conv k=128 pad=64, drop last frame, GELU; added  (HF:389-397,555-556)
    w = pos_conv_weight(sd, prefix, dtype)
    pc = F.conv1d(h.transpose(1, 2), w, _t(sd, prefix + "pos_conv_embed.conv.bias", dtype),
                  padding=POS_CONV_K // 2, groups=POS_CONV_GROUPS)[:, :, :-1]
    h = h + gelu_erf(pc).transpose(1, 2)
    # sinusoidal positions: valid frames 2,3,...; padded frames -> row 1 (zeros)  (HF:322-351,558-564)
    if frames is None:
        valid = torch.ones(B, T, dtype=torch.long)
    else:
        valid = (torch.arange(T)[None, :] < frames[:, None]).long()
    pos = torch.cumsum(valid, dim=1) * valid + PAD_IDX
    tab = sinusoid_table(max(int(pos.max()) + 1, PAD_IDX + 1 + T + 2), h.shape[-1], dtype)
    h = h + tab[pos]
    if taps is not None:
        taps["prenet"] = h
    return h, frames


def attention_core(q, k, v, pe_k, frames, q_block=512):
    """softmax(q k^T + rel-pos bias + key mask) v for q (already scaled), k, v of shape [B,H,T,dh]
    (HF:930-969), in query blocks, with bias[i,j] = (q_i . pe_k^T)[clip(i-j,-160,159)+160]."""
    B, H, T, dh = q.shape
    dtype = q.dtype
    out = torch.empty(B, H, T, dh, dtype=dtype)
    jj = torch.arange(T)
    neg = torch.finfo(dtype).min
    for i0 in range(0, T, q_block):
        i1 = min(T, i0 + q_block)
        qb = q[:, :, i0:i1]
        s = qb @ k.transpose(-1, -2)  # [B,H,bq,T]
        qp = qb @ pe_k.t()  # [B,H,bq,320] -- uses the already-scaled q (HF:939-945)
        rel = (torch.arange(i0, i1)[:, None] - jj[None, :]).clamp(-REL_MAX, REL_MAX - 1) + REL_MAX
        s = s + torch.gather(qp, 3, rel[None, None].expand(B, H, -1, -1))
        if frames is not None:
            masked = jj[None, :] >= frames[:, None]  # [B,T]
            s = s + masked[:, None, None, :].to(dtype) * neg  # additive finfo.min (HF:947-953)
        p = torch.softmax(s, dim=-1)
        out[:, :, i0:i1] = p @ v
    return out


def attention(x, frames, sd, lp, pe_k, dtype=torch.float32, q_block=512):
    """HF:872-986 with the compact relative-position bias and query blocking.
    x [B,T,768]; frames LongTensor[B] or None (keys >= frames[b] are masked for every query)."""
    B, T, D = x.shape
    H = HEADS
    dh = D // H
    q = F.linear(x, _t(sd, lp + "attention.q_proj.weight", dtype), _t(sd, lp + "attention.q_proj.bias", dtype)) * dh ** -0.5
    k = F.linear(x, _t(sd, lp + "attention.k_proj.weight", dtype), _t(sd, lp + "attention.k_proj.bias", dtype))
    v = F.linear(x, _t(sd, lp + "attention.v_proj.weight", dtype), _t(sd, lp + "attention.v_proj.bias", dtype))
    q = q.view(B, T, H, dh).transpose(1, 2)  # [B,H,T,dh]
    k = k.view(B, T, H, dh).transpose(1, 2)
    v = v.view(B, T, H, dh).transpose(1, 2)
    out = attention_core(q, k, v, pe_k, frames, q_block)
    o = out.transpose(1, 2).reshape(B, T, D)
    return F.linear(o, _t(sd, lp + "attention.out_proj.weight", dtype), _t(sd, lp + "attention.out_proj.bias", dtype))


def encoder_layer(x, frames, sd, lp, pe_k, dtype=torch.float32, q_block=512):
    """HF:1027-1067 post-LN: h = LN1(x + Attn(x)); y = LN2(h + FFN(h))."""
    D = x.shape[-1]
    h = x + attention(x, frames, sd, lp, pe_k, dtype, q_block)
    h = F.layer_norm(h, (D,), _t(sd, lp + "layer_norm.weight", dtype), _t(sd, lp + "layer_norm.bias", dtype), LN_EPS)
    f = F.linear(h, _t(sd, lp + "feed_forward.intermediate_dense.weight", dtype),
                 _t(sd, lp + "feed_forward.intermediate_dense.bias", dtype))
    f = gelu_erf(f)
    f = F.linear(f, _t(sd, lp + "feed_forward.output_dense.weight", dtype),
                 _t(sd, lp + "feed_forward.output_dense.bias", dtype))
    y = h + f
    return F.layer_norm(y, (D,), _t(sd, lp + "final_layer_norm.weight", dtype),
                        _t(sd, lp + "final_layer_norm.bias", dtype), LN_EPS)


def num_layers(sd, prefix="wrapped_encoder."):
    n = 0
    while f"{prefix}layers.{n}.layer_norm.weight" in sd:
        n += 1
    return n


def wrapped_encoder(h, frames, sd, prefix="wrapped_encoder.", dtype=torch.float32, q_block=512, hidden_states=None):
    """HF:1234-1322 (eval: dropout and LayerDrop are no-ops)."""
    D = h.shape[-1]
    h = F.layer_norm(h, (D,), _t(sd, prefix + "layer_norm.weight", dtype), _t(sd, prefix + "layer_norm.bias", dtype), LN_EPS)
    pe_k = _t(sd, prefix + "embed_positions.pe_k.weight", dtype)
    for l in range(num_layers(sd, prefix)):
        if hidden_states is not None:
            hidden_states.append(h)
        h = encoder_layer(h, frames, sd, f"{prefix}layers.{l}.", pe_k, dtype, q_block)
    if hidden_states is not None:
        hidden_states.append(h)
    return h


@torch.no_grad()
def encode(input_values, attention_mask, sd, dtype=torch.float32, q_block=512, taps=None, hidden_states=None):
    """SpeechT5EncoderWithSpeechPrenet.forward (HF:1339-1358): waveform [B,L] (+ int mask [B,L] or
    None) -> last_hidden_state [B,T,768].  ``sd`` holds numpy/torch tensors under the full HF key
    names (``prenet.*`` / ``wrapped_encoder.*``)."""
    x = torch.as_tensor(input_values)
    m = None if attention_mask is None else torch.as_tensor(attention_mask)
    h, frames = speech_prenet(x, m, sd, "prenet.", dtype, taps)
    return wrapped_encoder(h, frames, sd, "wrapped_encoder.", dtype, q_block, hidden_states)


def scaled_positional_table(n_rows, dim=768, dtype=torch.float32):
    """SpeechT5ScaledPositionalEncoding.__init__ (HF modeling_speecht5.py, class SpeechT5ScaledPositionalEncoding):
    pe[p, 0::2] = sin(p * w), pe[p, 1::2] = cos(p * w), w_k = exp(2k * -(ln 10000 / dim)); built in fp32 as HF does."""
    pe = torch.zeros(n_rows, dim)
    position = torch.arange(0, n_rows).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.int64).float() * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe.to(dtype)


def text_prenet(input_ids, sd, prefix="text_prenet.", dtype=torch.float32):
    """SpeechT5TextEncoderPrenet.forward (HF modeling_speecht5.py: embed_tokens, then SpeechT5ScaledPositionalEncoding.forward:
    emb + alpha * pe[:, :T]; dropout is the identity in eval).  ids [B,T] -> [B,T,768]."""
    ids = torch.as_tensor(input_ids).long()
    emb = _t(sd, prefix + "embed_tokens.weight", dtype)[ids]
    alpha = _t(sd, prefix + "encode_positions.alpha", dtype).reshape(())
    pe = scaled_positional_table(ids.shape[1], emb.shape[-1], dtype)
    return emb + alpha * pe[None]


@torch.no_grad()
def encode_text(input_ids, attention_mask, sd, dtype=torch.float32, q_block=512, hidden_states=None):
    """SpeechT5EncoderWithTextPrenet.forward (HF modeling_speecht5.py): text prenet, then the same wrapped encoder as the
    speech path.  The reference calls it WITHOUT a mask (/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:86:
    pad tokens attend like any other token); with a mask (right padding) the keys beyond each row's token count are masked."""
    h = text_prenet(input_ids, sd, "text_prenet.", dtype)
    frames = None
    if attention_mask is not None:
        m = torch.as_tensor(attention_mask)
        frames = m.long().sum(1)
        assert bool((m[:, 1:] <= m[:, :-1]).all()), "the restatement covers right padding (prefix masks)"
    return wrapped_encoder(h, frames, sd, "wrapped_encoder.", dtype, q_block, hidden_states)


# ---------------------------------------------------------------------------------------------------------
# Probe-row variants: the same maths restricted to a few output rows, so that 10-minute inputs (T = 29 999,
# BASELINE.json configs[2]) can be checked in seconds.  They take the stage INPUT (as produced by the
# implementation under test) and return the stage output for the probe rows only.

@torch.no_grad()
def encoder_layer_rows(x, rows, frames, sd, lp, pe_k, dtype=torch.float64):
    """Output rows `rows` of SpeechT5EncoderLayer (HF:1027-1067) for ONE clip x [T,768]; K/V use all T frames."""
    x = torch.as_tensor(x).to(dtype)
    T, D = x.shape
    H, dh = HEADS, D // HEADS
    rows = torch.as_tensor(rows, dtype=torch.long)
    xr = x[rows]
    q = F.linear(xr, _t(sd, lp + "attention.q_proj.weight", dtype), _t(sd, lp + "attention.q_proj.bias", dtype)) * dh ** -0.5
    k = F.linear(x, _t(sd, lp + "attention.k_proj.weight", dtype), _t(sd, lp + "attention.k_proj.bias", dtype))
    v = F.linear(x, _t(sd, lp + "attention.v_proj.weight", dtype), _t(sd, lp + "attention.v_proj.bias", dtype))
    q = q.view(-1, H, dh).transpose(0, 1)  # [H,R,dh]
    k = k.view(T, H, dh).transpose(0, 1)
    v = v.view(T, H, dh).transpose(0, 1)
    s = q @ k.transpose(-1, -2)  # [H,R,T]
    qp = q @ pe_k.to(dtype).t()  # [H,R,320]
    rel = (rows[:, None] - torch.arange(T)[None, :]).clamp(-REL_MAX, REL_MAX - 1) + REL_MAX
    s = s + torch.gather(qp, 2, rel[None].expand(H, -1, -1))
    if frames is not None:
        s = s.masked_fill((torch.arange(T) >= int(frames))[None, None, :], float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(0, 1).reshape(len(rows), D)
    a = F.linear(o, _t(sd, lp + "attention.out_proj.weight", dtype), _t(sd, lp + "attention.out_proj.bias", dtype))
    h = F.layer_norm(xr + a, (D,), _t(sd, lp + "layer_norm.weight", dtype), _t(sd, lp + "layer_norm.bias", dtype), LN_EPS)
    f = gelu_erf(F.linear(h, _t(sd, lp + "feed_forward.intermediate_dense.weight", dtype),
                          _t(sd, lp + "feed_forward.intermediate_dense.bias", dtype)))
    f = F.linear(f, _t(sd, lp + "feed_forward.output_dense.weight", dtype), _t(sd, lp + "feed_forward.output_dense.bias", dtype))
    return F.layer_norm(h + f, (D,), _t(sd, lp + "final_layer_norm.weight", dtype),
                        _t(sd, lp + "final_layer_norm.bias", dtype), LN_EPS)


@torch.no_grad()
def feature_encoder_window(x, sd, frame_lo, frame_hi, prefix="prenet.", dtype=torch.float64, stats_cache=None):
    """Frames [frame_lo, frame_hi) of the conv stack (HF:484-494) for ONE clip x [L]: GroupNorm statistics over the
    whole clip (conv0 is cheap: 1 GFLOP per 30 s), layers 1-6 only on the window's receptive field.  stats_cache: a dict the
    caller keeps per clip, so that several windows of one hour-long clip pay for the whole-clip statistics once."""
    x = torch.as_tensor(x).to(dtype)[None, None]
    w0 = _t(sd, prefix + "feature_encoder.conv_layers.0.conv.weight", dtype)
    if stats_cache is not None and "mean" in stats_cache:
        mean, var = stats_cache["mean"], stats_cache["var"]
    else:
        # statistics in chunks to bound memory
        n0 = (x.shape[-1] - 10) // 5 + 1
        s1 = torch.zeros(512, dtype=torch.float64)
        s2 = torch.zeros(512, dtype=torch.float64)
        step = 200000
        for t0 in range(0, n0, step):
            t1 = min(n0, t0 + step)
            y = F.conv1d(x[..., 5 * t0:5 * (t1 - 1) + 10], w0, stride=5)[0].double()
            s1 += y.sum(1)
            s2 += (y * y).sum(1)
        mean = (s1 / n0)
        var = s2 / n0 - mean * mean
        if stats_cache is not None:
            stats_cache["mean"], stats_cache["var"] = mean, var
    # receptive field of frames [lo, hi) back through layers 6..1
    lo, hi = frame_lo, frame_hi - 1
    for k, s in reversed(list(zip(CONV_KERNEL[1:], CONV_STRIDE[1:]))):
        lo, hi = lo * s, hi * s + k - 1
    y = F.conv1d(x[..., 5 * lo:5 * hi + 10], w0, stride=5)
    gw = _t(sd, prefix + "feature_encoder.conv_layers.0.layer_norm.weight", dtype)[None, :, None]
    gb = _t(sd, prefix + "feature_encoder.conv_layers.0.layer_norm.bias", dtype)[None, :, None]
    y = (y - mean.to(dtype)[None, :, None]) * torch.rsqrt(var.to(dtype) + LN_EPS)[None, :, None] * gw + gb
    h = gelu_erf(y)
    for i, (k, s) in enumerate(zip(CONV_KERNEL[1:], CONV_STRIDE[1:]), start=1):
        h = gelu_erf(F.conv1d(h, _t(sd, f"{prefix}feature_encoder.conv_layers.{i}.conv.weight", dtype), stride=s))
    assert h.shape[-1] == frame_hi - frame_lo, (h.shape, frame_lo, frame_hi)
    return h[0].t().contiguous()


@torch.no_grad()
def pos_conv_rows(h, rows, nvalid, sd, prefix="prenet.", dtype=torch.float64):
    """Rows `rows` of hidden + GELU(pos_conv(hidden)) + sinusoid (HF:555-564) for ONE clip h [T,768]."""
    h = torch.as_tensor(h).to(dtype)
    T = h.shape[0]
    w = pos_conv_weight(sd, prefix, dtype)
    b = _t(sd, prefix + "pos_conv_embed.conv.bias", dtype)
    tab = sinusoid_table(T + 3, h.shape[-1], dtype)
    out = []
    for r in rows:
        lo, hi = r - POS_CONV_K // 2, r + POS_CONV_K // 2  # taps read h[r-64 .. r+63]
        win = torch.zeros(POS_CONV_K, h.shape[1], dtype=dtype)
        a, bnd = max(lo, 0), min(hi, T)
        win[a - lo:bnd - lo] = h[a:bnd]
        pc = F.conv1d(win.t()[None], w, b, groups=POS_CONV_GROUPS)[0, :, 0]
        pos = r + 2 if r < nvalid else PAD_IDX
        out.append(h[r] + gelu_erf(pc) + tab[pos])
    return torch.stack(out)

"""Per-kernel parity on one MI355X: every HIP kernel, called through the C ABI (include/loco_asr.h), against
the same op of the CPU oracle / plain torch fp32 on identical seeded inputs.

Tolerances (relative L2, fp32 vs fp32): 2e-6 for byte movers and reductions, 1e-5 for the MFMA
contractions (exact-fp32 products, different summation order than the CPU's), far inside the 1e-3 the
embedding must meet end to end.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import check, dev, la, lib, ptr, rel_l2, stream


def hu(key, shape, scale=1.0):
    return torch.from_numpy(la.synth.hashed_uniform(key, shape, 11)) * scale


@pytest.mark.parametrize("rows,dim", [(1, 768), (5, 512), (1499 * 2, 768), (4097, 512)])
def test_layernorm(rows, dim):
    x = hu("ln.x", (rows, dim), 3.0) + 0.5
    g = hu("ln.g", (dim,)) + 1.0
    b = hu("ln.b", (dim,))
    xd, gd, bd, y = dev(x), dev(g), dev(b), torch.empty(rows, dim, device="cuda")
    check(lib().loco_op_layernorm(ptr(xd), ptr(gd), ptr(bd), ptr(y), rows, dim, 1e-5, stream()))
    ref = F.layer_norm(x, (dim,), g, b, 1e-5)
    assert rel_l2(y, ref) < 2e-6
    # in place
    check(lib().loco_op_layernorm(ptr(xd), ptr(gd), ptr(bd), ptr(xd), rows, dim, 1e-5, stream()))
    assert torch.equal(xd, y)


def test_layernorm_rejects_other_widths():
    x = torch.zeros(4, 640, device="cuda")
    with pytest.raises(ValueError):
        check(lib().loco_op_layernorm(ptr(x), ptr(x), ptr(x), ptr(x), 4, 640, 1e-5, stream()))


def gemm(A, W, bias=None, R=None, epi=0, lda=None, M=None, nb1=1, nb2=1, sA=(0, 0), sC=(0, 0), out=None, ldc=None):
    N, K = W.shape
    lda = lda if lda is not None else A.shape[-1]
    M = M if M is not None else A.shape[0]
    ldc = ldc or N
    C_ = out if out is not None else torch.empty(nb1 * nb2 * M, N, device="cuda")
    check(lib().loco_op_gemm(ptr(A), lda, ptr(W), W.shape[1], ptr(bias), ptr(R), ldc, ptr(C_), ldc, M, N, K, epi, nb1, nb2,
                             sA[0], sA[1], sC[0], sC[1], stream()), "gemm")
    return C_


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1, 768, 768), (300, 320, 64), (1499, 2304, 768), (257, 768, 3072),
                                    (130, 512, 1536)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_linear(M, N, K, epi):
    A = hu("g.a", (M, K))
    W = hu("g.w", (N, K), 2.0 / math.sqrt(K))
    b = hu("g.b", (N,))
    R = hu("g.r", (M, N))
    Ad, Wd, bd, Rd = dev(A), dev(W), dev(b), dev(R)
    out = gemm(Ad, Wd, bd, Rd if epi == 2 else None, epi)
    ref = A.double() @ W.double().t() + b.double()
    if epi == 1:
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))
    if epi == 2:
        ref = ref + R.double()
    assert rel_l2(out, ref) < 1e-6  # vs fp64: fp32 MFMA is an exact-product fmaf chain


def test_gemm_identity_asymmetric():
    # A = I with an asymmetric W catches a transposed C write (guide §3)
    K = 128
    A = torch.eye(K)
    W = hu("g.asym", (96, K))
    Ad, Wd = dev(A), dev(W)
    out = gemm(Ad, Wd)
    assert torch.equal(out.cpu(), W.t().contiguous())


@pytest.mark.parametrize("k,s,Tin,B", [(3, 2, 401, 2), (2, 2, 37, 3), (3, 2, 4799, 1)])
def test_gemm_as_strided_conv(k, s, Tin, B):
    """Conv1d layers 1-6 (HF modeling:216-228): channels-last rows, lda = s*C, K = k*C, tap-major weight."""
    Cc = 512
    x = hu("c.x", (B, Tin, Cc))
    w = hu("c.w", (Cc, Cc, k), math.sqrt(2.0 / (Cc * k)))
    Tout = (Tin - k) // s + 1
    wt = w.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous()
    xd, wtd = dev(x), dev(wt)
    out = gemm(xd, wtd, epi=1, lda=s * Cc, M=Tout, nb1=B, sA=(Tin * Cc, 0), sC=(Tout * Cc, 0))
    ref = F.conv1d(x.transpose(1, 2).double(), w.double(), stride=s)
    ref = (0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))).transpose(1, 2).reshape(B * Tout, Cc)
    assert rel_l2(out, ref) < 1e-6


def test_gemm_rejects_bad_k():
    a = torch.zeros(4, 48, device="cuda")
    with pytest.raises(ValueError):
        gemm(a, torch.zeros(8, 48, device="cuda"))


@pytest.mark.parametrize("lengths", [[400], [16000, 9000], [80000, 48000, 801]])
def test_conv0_groupnorm_gelu(lengths, oracle):
    x, m = la.synth.batch(lengths)
    B, L = x.shape
    sd = la.synth.encoder_state_dict(0, layers=0)
    p = "prenet.feature_encoder.conv_layers.0."
    w, gw, gb = sd[p + "conv.weight"], sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"]
    T0 = (L - 10) // 5 + 1
    out = torch.empty(B, T0, 512, device="cuda")
    scratch = torch.empty(lib().loco_conv0_scratch_bytes(B), dtype=torch.uint8, device="cuda")
    xd, wd, gwd, gbd = dev(x), dev(w.reshape(512, 10)), dev(gw), dev(gb)
    check(lib().loco_op_conv0_gn_gelu(ptr(xd), B, L, ptr(wd), ptr(gwd), ptr(gbd), ptr(out), ptr(scratch), stream()))
    h = F.conv1d(torch.from_numpy(x).double()[:, None], torch.from_numpy(w).double(), stride=5)
    mean = h.mean(2, keepdim=True)
    var = ((h - mean) ** 2).mean(2, keepdim=True)
    h = (h - mean) / torch.sqrt(var + 1e-5) * torch.from_numpy(gw).double()[None, :, None] + torch.from_numpy(gb).double()[None, :, None]
    ref = (0.5 * h * (1 + torch.erf(h / math.sqrt(2)))).transpose(1, 2)
    assert rel_l2(out, ref) < 2e-6
    # bitwise reproducible (fixed-order fp64 reduction)
    out2 = torch.empty_like(out)
    check(lib().loco_op_conv0_gn_gelu(ptr(xd), B, L, ptr(wd), ptr(gwd), ptr(gbd), ptr(out2), ptr(scratch), stream()))
    assert torch.equal(out, out2)


def test_conv0_dc_offset_is_stable():
    """GroupNorm statistics from waveform moments: a large DC offset must not lose the variance."""
    x = (la.synth.clip(3, 8000) + 5.0)[None]
    sd = la.synth.encoder_state_dict(0, layers=0)
    p = "prenet.feature_encoder.conv_layers.0."
    w, gw, gb = sd[p + "conv.weight"], sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"]
    T0 = (8000 - 10) // 5 + 1
    out = torch.empty(1, T0, 512, device="cuda")
    scratch = torch.empty(lib().loco_conv0_scratch_bytes(1), dtype=torch.uint8, device="cuda")
    xd, wd, gwd, gbd = dev(x), dev(w.reshape(512, 10)), dev(gw), dev(gb)
    check(lib().loco_op_conv0_gn_gelu(ptr(xd), 1, 8000, ptr(wd), ptr(gwd), ptr(gbd), ptr(out), ptr(scratch), stream()))
    h = F.conv1d(torch.from_numpy(x).double()[:, None], torch.from_numpy(w).double(), stride=5)
    h = (h - h.mean(2, keepdim=True)) / torch.sqrt(h.var(2, unbiased=False, keepdim=True) + 1e-5)
    h = h * torch.from_numpy(gw).double()[None, :, None] + torch.from_numpy(gb).double()[None, :, None]
    ref = (0.5 * h * (1 + torch.erf(h / math.sqrt(2)))).transpose(1, 2)
    assert rel_l2(out, ref) < 2e-5  # the fp32 conv output itself carries the offset; stats stay exact


def test_frame_counts(oracle):
    L = 100000
    lens = [100000, 99999, 400, 399 + 320, 48000, 80000, 12345]
    m = torch.zeros(len(lens), L, dtype=torch.int32)
    for i, n in enumerate(lens):
        m[i, :n] = 1
    fr = torch.empty(len(lens), dtype=torch.int32, device="cuda")
    md = m.cuda()
    check(lib().loco_op_frame_counts(ptr(md), len(lens), L, ptr(fr), stream()))
    assert fr.cpu().tolist() == [oracle.feat_extract_output_lengths(n) for n in lens]
    check(lib().loco_op_frame_counts(None, len(lens), L, ptr(fr), stream()))
    assert fr.cpu().tolist() == [oracle.feat_extract_output_lengths(L)] * len(lens)
    assert lib().loco_output_frames(480000) == 1499 and lib().loco_output_frames(9600000) == 29999


def fold_pos_conv(sd, oracle):
    w = oracle.pos_conv_weight(sd)  # [768,48,128]
    return w.view(16, 48, 48, 128).permute(0, 3, 1, 2).contiguous()  # [g][tap][o][i]


@pytest.mark.parametrize("B,T,frames", [(1, 1, None), (2, 49, [49, 30]), (1, 300, None), (3, 129, [129, 1, 77])])
def test_pos_conv_and_sinusoid(B, T, frames, oracle):
    sd = la.synth.encoder_state_dict(0, layers=0)
    h = hu("pc.h", (B, T, 768), 1.5)
    wf = fold_pos_conv(sd, oracle)
    bias = torch.from_numpy(sd["prenet.pos_conv_embed.conv.bias"])
    tab = la.sinusoid_table(T + 2)
    fr = None if frames is None else torch.tensor(frames, dtype=torch.int32)
    out = torch.empty(B, T, 768, device="cuda")
    hd, wfd, bd, td = dev(h), dev(wf), dev(bias), dev(tab)
    frd = fr.cuda() if fr is not None else None
    check(lib().loco_op_pos_conv(ptr(hd), ptr(wfd), ptr(bd), ptr(td), ptr(frd), ptr(out), B, T, stream()))
    w = oracle.pos_conv_weight(sd).double()
    pc = F.conv1d(h.double().transpose(1, 2), w, bias.double(), padding=64, groups=16)[:, :, :-1]
    ref = h.double() + (0.5 * pc * (1 + torch.erf(pc / math.sqrt(2)))).transpose(1, 2)
    valid = torch.ones(B, T, dtype=torch.long) if fr is None else (torch.arange(T)[None] < fr[:, None].long()).long()
    pos = torch.cumsum(valid, 1) * valid + 1
    ref = ref + tab.double()[pos]
    assert rel_l2(out, ref) < 2e-6


@pytest.mark.parametrize("B,T,frames", [(1, 1, None), (2, 200, [200, 131]), (1, 333, None), (2, 450, [450, 65]), (1, 700, [64])])
def test_attention_core(B, T, frames, oracle):
    """flash attention incl. the compact relative-position bias across |i-j| = 159/160/161 and key masks."""
    qkv = hu("at.qkv", (B, T, 2304), 1.5)
    qkv[..., :768] *= 0.125 * 1.5  # q arrives pre-scaled
    pe_k = hu("at.pe", (320, 64), 0.9)
    q = qkv[..., :768].view(B, T, 12, 64).transpose(1, 2)
    k = qkv[..., 768:1536].view(B, T, 12, 64).transpose(1, 2)
    v = qkv[..., 1536:].view(B, T, 12, 64).transpose(1, 2)
    qp = (q @ pe_k.t()).contiguous()  # [B,12,T,320]
    fr = None if frames is None else torch.tensor(frames, dtype=torch.int32)
    ctx = torch.empty(B, T, 768, device="cuda")
    qkvd, qpd = dev(qkv), dev(qp)
    frd = fr.cuda() if fr is not None else None
    check(lib().loco_op_attention(ptr(qkvd), ptr(qpd), ptr(frd), ptr(ctx), B, T, stream()))
    ref = oracle.attention_core(q.double(), k.double(), v.double(), pe_k.double(), None if fr is None else fr.long(), q_block=128)
    ref = ref.transpose(1, 2).reshape(B, T, 768)
    assert rel_l2(ctx, ref) < 3e-6


def test_attention_forces_online_softmax_rescale(oracle):
    """A key far down the sequence that dominates every row forces the running max to jump at a late tile
    (guide rule 26: the rescale branch needs an input that takes it)."""
    B, T = 1, 400
    qkv = hu("at2.qkv", (B, T, 2304), 1.0)
    qkv[..., :768] *= 0.125
    qkv[0, 300, 768:1536] = qkv[0, :, :768].mean(0) * 0 + 6.0  # key 300: large positive dot with most queries
    qkv[0, :, :768] = qkv[0, :, :768].abs()
    pe_k = hu("at2.pe", (320, 64), 0.3)
    q = qkv[..., :768].view(B, T, 12, 64).transpose(1, 2)
    k = qkv[..., 768:1536].view(B, T, 12, 64).transpose(1, 2)
    v = qkv[..., 1536:].view(B, T, 12, 64).transpose(1, 2)
    qp = (q @ pe_k.t()).contiguous()
    ctx = torch.empty(B, T, 768, device="cuda")
    qkvd, qpd = dev(qkv), dev(qp)
    check(lib().loco_op_attention(ptr(qkvd), ptr(qpd), None, ptr(ctx), B, T, stream()))
    ref = oracle.attention_core(q.double(), k.double(), v.double(), pe_k.double(), None).transpose(1, 2).reshape(B, T, 768)
    assert rel_l2(ctx, ref) < 3e-6


def split16(x):
    xd = dev(x)
    hi = torch.empty(xd.shape, dtype=torch.float16, device="cuda")
    lo = torch.empty_like(hi)
    check(lib().loco_op_split_f16(ptr(xd), ptr(hi), ptr(lo), xd.numel(), stream()))
    return hi, lo


def test_split_f16_reconstructs_22_bits():
    x = hu("sp.x", (4, 1000), 3.0)
    hi, lo = split16(x)
    assert torch.equal(hi.cpu(), x.half())
    rec = hi.float().cpu() + lo.float().cpu()
    assert float(((rec - x).abs() / x.abs().clamp_min(1e-3)).max()) < 2.0 ** -20


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1, 768, 768), (300, 320, 64), (1499, 2304, 768), (257, 768, 3072)])
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("out_split", [False, True])
def test_gemm_f16x3(M, N, K, epi, out_split):
    """split-precision GEMM vs fp64: fp32-class accuracy (bar 5e-6; exact fp32 MFMA gives ~2e-7, plain fp16 ~3e-4)."""
    A = hu("g3.a", (M, K), 2.0)
    W = hu("g3.w", (N, K), 2.0 / math.sqrt(K))
    b = hu("g3.b", (N,))
    R = hu("g3.r", (M, N))
    ahi, alo = split16(A)
    whi, wlo = split16(W)
    bd, Rd = dev(b), dev(R)
    C_ = torch.empty(M, N, device="cuda")
    chi = torch.empty(M, N, dtype=torch.float16, device="cuda")
    clo = torch.empty_like(chi)
    check(lib().loco_op_gemm_f16x3(ptr(ahi), ptr(alo), K, ptr(whi), ptr(wlo), K, ptr(bd), ptr(Rd) if epi == 2 else None, N,
                                   None if out_split else ptr(C_), ptr(chi) if out_split else None, ptr(clo) if out_split else None,
                                   N, M, N, K, epi, 1, 1, 0, 0, 0, 0, stream()))
    ref = A.double() @ W.double().t() + b.double()
    if epi == 1:
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))
    if epi == 2:
        ref = ref + R.double()
    out = (chi.float() + clo.float()) if out_split else C_
    assert rel_l2(out, ref) < 5e-6


def planes(x):
    hi = x.half()
    lo = (x - hi.float()).half()
    return hi.cuda().contiguous(), lo.cuda().contiguous()


@pytest.mark.parametrize("B,T,frames", [(1, 1, None), (2, 200, [200, 131]), (1, 333, None), (2, 450, [450, 65]), (1, 700, [64])])
def test_attention_core_f16x3(B, T, frames, oracle):
    """split-precision attention vs the fp64 oracle on the SAME (hi+lo) operands: fp32-class accuracy (bar 1e-5)."""
    qkv = hu("at.qkv", (B, T, 2304), 1.5)
    qkv[..., :768] *= 0.125 * 1.5
    pe_k = hu("at.pe", (320, 64), 0.9)
    q = qkv[..., :768].view(B, T, 12, 64).transpose(1, 2)
    k = qkv[..., 768:1536].view(B, T, 12, 64).transpose(1, 2)
    v = qkv[..., 1536:].view(B, T, 12, 64).transpose(1, 2)
    qp = (q @ pe_k.t()).contiguous()
    qh, ql = planes(qkv[..., :768].reshape(B * T, 768))
    kh, kl = planes(qkv[..., 768:1536].reshape(B * T, 768))
    vh, vl = planes(qkv[..., 1536:].reshape(B * T, 768))  # row-major like q and k: the kernel transposes V with its LDS read
    fr = None if frames is None else torch.tensor(frames, dtype=torch.int32)
    frd = fr.cuda() if fr is not None else None
    qpd = dev(qp)
    ctx = torch.empty(B, T, 768, device="cuda")
    check(lib().loco_op_attention_f16x3(ptr(qh), ptr(ql), ptr(kh), ptr(kl), ptr(vh), ptr(vl), ptr(qpd), ptr(frd), ptr(ctx), B, T, stream()))
    ref = oracle.attention_core(q.double(), k.double(), v.double(), pe_k.double(), None if fr is None else fr.long(), q_block=128)
    ref = ref.transpose(1, 2).reshape(B, T, 768)
    assert rel_l2(ctx, ref) < 1e-5


@pytest.mark.parametrize("B,T,frames", [(2, 249, [249, 162]), (1, 1499, None), (3, 700, [700, 333, 64])])
def test_attention_f16x3_computes_the_relative_position_table_itself(B, T, frames, oracle):
    """The form loco_forward runs (round 3): no table GEMM in front of attention -- every wave computes Qp = q . pe_k^T for its own 32
    queries (ten 32x32 MFMA blocks on the Q fragments it holds), writes it to the scratch table and reads the band back.  Against the
    fp64 oracle on the same hi+lo operands, against the external-table form of the same kernel, and the scratch must hold the table
    (pe planes scaled by 2^9 with pe_scale 2^-9, as loco_finalize_weights stores them)."""
    qkv = hu("att.qkv", (B, T, 2304), 1.5)
    qkv[..., :768] *= 0.125 * 1.5
    pe_k = hu("att.pe", (320, 64), 0.9)
    q = qkv[..., :768].view(B, T, 12, 64).transpose(1, 2)
    k = qkv[..., 768:1536].view(B, T, 12, 64).transpose(1, 2)
    v = qkv[..., 1536:].view(B, T, 12, 64).transpose(1, 2)
    qh, ql = planes(qkv[..., :768].reshape(B * T, 768))
    kh, kl = planes(qkv[..., 768:1536].reshape(B * T, 768))
    vh, vl = planes(qkv[..., 1536:].reshape(B * T, 768))
    ph, pl_ = planes(pe_k * 512.0)
    fr = None if frames is None else torch.tensor(frames, dtype=torch.int32)
    frd = fr.cuda() if fr is not None else None
    scratch = torch.full((B, 12, T, 320), float("nan"), device="cuda")
    ctx = torch.empty(B, T, 768, device="cuda")
    check(lib().loco_op_attention_f16x3_pe(ptr(qh), ptr(ql), ptr(kh), ptr(kl), ptr(vh), ptr(vl), ptr(ph), ptr(pl_), 1.0 / 512.0, ptr(scratch),
                                           ptr(frd), ptr(ctx), B, T, stream()))
    # the operands the kernel really multiplies: q and pe_k as their hi + lo planes
    qd = (qh.double() + ql.double()).view(B, T, 12, 64).transpose(1, 2).cpu()
    ped = ((ph.double() + pl_.double()) / 512.0).cpu()
    # the scratch holds, per query, the 32-column blocks some key of its band can read (include/loco_asr.h); the rest is untouched
    want = (qd @ ped.t())
    got = scratch.cpu()
    formed = torch.zeros_like(got, dtype=torch.bool)
    for b in range(B):
        nk = 64 * (((int(fr[b]) if fr is not None else T) + 63) // 64)   # keys up to the end of the last key tile
        for i0 in range(0, T, 32):
            lo, hi = max(i0 - (nk - 1), -160), min(i0 + 31, 159)
            formed[b, :, i0:i0 + 32, 32 * ((lo + 160) // 32):32 * ((hi + 160) // 32) + 32] = True
    assert bool(torch.isnan(got[~formed]).all()) and bool(torch.isfinite(got[formed]).all())
    assert rel_l2(got[formed], want[formed]) < 2e-6
    if T == 249:
        assert 0.5 < float(formed.float().mean()) < 0.9   # an utterance-length clip: a good part of the table is never formed
    scratch = torch.where(formed.cuda(), scratch, torch.zeros_like(scratch))  # the external-table form below reads columns 0 / 319 of every row
    ref = oracle.attention_core(qd, k.double(), v.double(), ped, None if fr is None else fr.long(), q_block=128).transpose(1, 2).reshape(B, T, 768)
    assert bool(torch.isfinite(ctx).all()) and rel_l2(ctx, ref) < 1e-5
    ctx2 = torch.empty_like(ctx)
    check(lib().loco_op_attention_f16x3(ptr(qh), ptr(ql), ptr(kh), ptr(kl), ptr(vh), ptr(vl), ptr(scratch), ptr(frd), ptr(ctx2), B, T, stream()))
    assert torch.equal(ctx, ctx2)  # same table, same kernel body: the two forms agree bit for bit


@pytest.mark.parametrize("key,gap", [(7, 30.0), (101, 30.0), (7, 80.0), (205, 14.0)])
def test_attention_f16x3_row_max_covers_both_lane_halves(key, gap, oracle):
    """Regression (round 3): each query's 64 keys of a tile sit in TWO lane halves (keys {0-3, 8-11, ...} / {4-7, 12-15, ...}); the
    row maximum needs both.  Written with __builtin_amdgcn_permlane32_swap(m, m), hipcc folded max(result[0], result[1]) away and
    every row used the LOWER half's maximum only.  Softmax is shift-invariant, so all goldens passed -- until the second weight
    family (g10) produced "massive activation" keys: a key of the upper half that beats the lower half's maximum by more than
    ~11 nats makes P = exp(s - m) > 65504, its fp16 hi plane inf, and the row's context NaN.  Here one key of the upper half
    (bit 2 of its index set: 7 in the prologue's tile, 101 and 205 in the loop's tiles) outscores everything by `gap` nats for
    every query; the output must be finite and match the fp64 oracle."""
    assert key & 4
    B, T = 2, 249
    qkv = hu("atx.qkv", (B, T, 2304), 1.5)
    qkv[..., :768] *= 0.125 * 1.5
    u = hu("atx.u", (64,), 1.0)
    u = u / u.norm()
    qv, kv = qkv[..., :768].view(B, T, 12, 64), qkv[..., 768:1536].view(B, T, 12, 64)
    qv -= (qv @ u)[..., None] * u           # remove, then plant: every query has component 1 along u ...
    qv += u
    kv -= (kv @ u)[..., None] * u           # ... and only the chosen key has one along it: its score leads by `gap`
    base = (qv.transpose(1, 2) @ kv.transpose(1, 2).transpose(-1, -2)).abs().max()
    kv[:, key] += (gap + float(base)) * u
    pe_k = hu("atx.pe", (320, 64), 0.9)
    q = qkv[..., :768].view(B, T, 12, 64).transpose(1, 2)
    k = qkv[..., 768:1536].view(B, T, 12, 64).transpose(1, 2)
    v = qkv[..., 1536:].view(B, T, 12, 64).transpose(1, 2)
    qp = (q @ pe_k.t()).contiguous()
    qh, ql = planes(qkv[..., :768].reshape(B * T, 768))
    kh, kl = planes(qkv[..., 768:1536].reshape(B * T, 768))
    vh, vl = planes(qkv[..., 1536:].reshape(B * T, 768))
    qpd = dev(qp)
    ctx = torch.empty(B, T, 768, device="cuda")
    check(lib().loco_op_attention_f16x3(ptr(qh), ptr(ql), ptr(kh), ptr(kl), ptr(vh), ptr(vl), ptr(qpd), None, ptr(ctx), B, T, stream()))
    assert bool(torch.isfinite(ctx).all()), f"{int((~torch.isfinite(ctx)).any(-1).sum())} rows are not finite"
    ref = oracle.attention_core(q.double(), k.double(), v.double(), pe_k.double(), None, q_block=128).transpose(1, 2).reshape(B, T, 768)
    assert rel_l2(ctx, ref) < 1e-5


@pytest.mark.parametrize("epi,out_split", [(0, False), (1, True), (2, False)])
def test_gemm_f16x3_large_tiles(epi, out_split):
    """Shapes that take the 256x256 / 16-wave LDS-DMA kernel (>= 768 tiles of 256x256) with a ragged last row tile."""
    M, N, K = 22001, 2304, 256
    A = hu("g4.a", (M, K), 2.0)
    W = hu("g4.w", (N, K), 2.0 / math.sqrt(K))
    b = hu("g4.b", (N,))
    R = hu("g4.r", (M, N))
    ahi, alo = split16(A)
    whi, wlo = split16(W)
    bd, Rd = dev(b), dev(R)
    C_ = torch.empty(M, N, device="cuda")
    chi = torch.empty(M, N, dtype=torch.float16, device="cuda")
    clo = torch.empty_like(chi)
    check(lib().loco_op_gemm_f16x3(ptr(ahi), ptr(alo), K, ptr(whi), ptr(wlo), K, ptr(bd), ptr(Rd) if epi == 2 else None, N,
                                   None if out_split else ptr(C_), ptr(chi) if out_split else None, ptr(clo) if out_split else None,
                                   N, M, N, K, epi, 1, 1, 0, 0, 0, 0, stream()))
    ref = A.double() @ W.double().t() + b.double()
    if epi == 1:
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))
    if epi == 2:
        ref = ref + R.double()
    out = (chi.float() + clo.float()) if out_split else C_
    assert rel_l2(out, ref) < 5e-6


@pytest.mark.parametrize("K", [32, 64, 96, 128, 160, 256, 1024])
def test_gemm_f16x3_every_tile_form_agrees_bit_for_bit(K):
    """The ring protocol of the LDS-DMA kernel (which slot a DMA may overwrite, which DMAs a counted vmcnt may leave in flight) is
    where a wrong count reads stale or half-landed bytes -- sporadically, and only in SOME tile form / ring depth.  Every form
    accumulates a given output element in the same order (k-tile by k-tile, three MFMAs per k-tile), so all forms must agree BIT
    FOR BIT, on every repeat, for k-loops shorter than, equal to and longer than the rings (K = 32 ... 1024 = 1 ... 32 k-tiles);
    one form is also checked against fp64.  (A prologue that waited for too few A pieces when both rings had three slots passed
    every tolerance test of this file by timing luck and failed only a determinism check of the whole encoder.)"""
    import os
    M, N = 30011, 768
    A = hu("g5.a", (M, K), 2.0)
    W = hu("g5.w", (N, K), 2.0 / math.sqrt(K))
    b = hu("g5.b", (N,))
    ahi, alo = split16(A)
    whi, wlo = split16(W)
    bd = dev(b)
    outs = {}
    try:
        for tile in (2, 1, 3, 4, 5, 6, 7, 8):
            os.environ["LOCO_GEMM_TILE"] = str(tile)
            lib().loco_debug_reload_gemm_knobs()  # the A/B knobs are read once and again on request (gemm_f16x3.hip)
            for rep in range(3):
                chi = torch.zeros(M, N, dtype=torch.float16, device="cuda")
                clo = torch.zeros_like(chi)
                check(lib().loco_op_gemm_f16x3(ptr(ahi), ptr(alo), K, ptr(whi), ptr(wlo), K, ptr(bd), None, N, None, ptr(chi), ptr(clo), N, M, N, K,
                                               1, 1, 1, 0, 0, 0, 0, stream()))
                torch.cuda.synchronize()
                if not outs:
                    ref = A.double() @ W.double().t() + b.double()
                    ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))
                    assert rel_l2(chi.float() + clo.float(), ref) < 5e-6
                    outs["hi"], outs["lo"] = chi, clo
                else:
                    nbad = int((chi.view(torch.int16) != outs["hi"].view(torch.int16)).sum()) + int((clo.view(torch.int16) != outs["lo"].view(torch.int16)).sum())
                    assert nbad == 0, (tile, rep, nbad)
    finally:
        os.environ.pop("LOCO_GEMM_TILE", None)
        lib().loco_debug_reload_gemm_knobs()


def test_gemm_f16x3_batched_strided_conv_large():
    """conv layer as a batched split GEMM on the big-tile kernel: 9 clips x 12 001 frames, k=3, stride 2."""
    Cc, k, s_, Tin, B = 512, 3, 2, 24003, 9
    x = hu("c4.x", (B, Tin, Cc))
    w = hu("c4.w", (Cc, Cc, k), math.sqrt(2.0 / (Cc * k)))
    Tout = (Tin - k) // s_ + 1
    wt = w.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous()
    xhi, xlo = split16(x)
    whi, wlo = split16(wt)
    chi = torch.empty(B * Tout, Cc, dtype=torch.float16, device="cuda")
    clo = torch.empty_like(chi)
    check(lib().loco_op_gemm_f16x3(ptr(xhi), ptr(xlo), s_ * Cc, ptr(whi), ptr(wlo), k * Cc, None, None, Cc, None, ptr(chi), ptr(clo), Cc,
                                   Tout, Cc, k * Cc, 1, B, 1, Tin * Cc, 0, Tout * Cc, 0, stream()))
    idx = torch.tensor([0, 1, 4000, Tout - 2, Tout - 1])
    ref = F.conv1d(x.transpose(1, 2).double(), w.double(), stride=s_)
    ref = (0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))).transpose(1, 2)  # [B,Tout,C]
    out = (chi.float() + clo.float()).view(B, Tout, Cc).cpu()
    assert rel_l2(out[:, idx], ref[:, idx]) < 5e-6
    assert rel_l2(out, ref) < 5e-6


@pytest.mark.parametrize("k,s_,Tin,B", [(3, 2, 24003, 9), (2, 2, 4001, 5), (3, 2, 700, 1)])
def test_conv_gemm_f16x3_channel_block_major_k(k, s_, Tin, B):
    """The form loco_forward runs conv layers 1-6 in: the k axis walked channel-block major (GemmSplitArgs::ktaps) over weights
    permuted by loco_op_permute_conv_k -- against torch's conv1d in fp64, and against the plain tap-major GEMM on the same
    operands (same products, another summation order: fp32-class agreement)."""
    Cc = 512
    x = hu("c5.x", (B, Tin, Cc))
    w = hu("c5.w", (Cc, Cc, k), math.sqrt(2.0 / (Cc * k)))
    Tout = (Tin - k) // s_ + 1
    wt = w.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous()  # [N][tap][C]
    wtd = dev(wt)
    wp = torch.empty_like(wtd)
    check(lib().loco_op_permute_conv_k(ptr(wtd), ptr(wp), Cc, k, Cc, stream()))
    order = [0, 2, 1] if k == 3 else list(range(k))  # tap slots: the two taps that share input rows (0 and 2) next to each other
    ref_perm = wt.view(Cc, k, Cc // 64, 2, 32)[:, order].permute(0, 2, 1, 3, 4).reshape(Cc, k * Cc)
    assert torch.equal(wp.cpu(), ref_perm)
    xhi, xlo = split16(x)
    phi, plo = split16(wp.cpu())
    whi, wlo = split16(wt)
    out = {}
    for name in ("blocked", "plain"):
        chi = torch.empty(B * Tout, Cc, dtype=torch.float16, device="cuda")
        clo = torch.empty_like(chi)
        if name == "blocked":
            check(lib().loco_op_conv_gemm_f16x3(ptr(xhi), ptr(xlo), s_ * Cc, ptr(phi), ptr(plo), None, ptr(chi), ptr(clo), Tout, Cc, Cc, k, 1, B,
                                                Tin * Cc, stream()))
        else:
            check(lib().loco_op_gemm_f16x3(ptr(xhi), ptr(xlo), s_ * Cc, ptr(whi), ptr(wlo), k * Cc, None, None, Cc, None, ptr(chi), ptr(clo), Cc,
                                           Tout, Cc, k * Cc, 1, B, 1, Tin * Cc, 0, Tout * Cc, 0, stream()))
        out[name] = (chi.float() + clo.float()).view(B, Tout, Cc).cpu()
    ref = F.conv1d(x.transpose(1, 2).double(), w.double(), stride=s_)
    ref = (0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))).transpose(1, 2)
    assert rel_l2(out["blocked"], ref) < 5e-6
    assert rel_l2(out["blocked"], out["plain"]) < 2e-6


def test_normalize_waveform_on_device_matches_hf_golden():
    """"next" row f-4: SpeechT5FeatureExtractor(do_normalize=True) on the device against HF's own output (fixture g7):
    values, padding, reproducibility, the mask-free form, and the feature extractor's deferred path end to end."""
    from conftest import golden
    g = golden("g7_normalize.npz")
    lengths = [int(n) for n in g["lengths"]]
    clips = [(la.synth.clip(i, n) * np.float32(0.5 + i) + np.float32(0.1 * i - 0.15)).astype(np.float32) for i, n in enumerate(lengths)]
    fe = la.SpeechT5FeatureExtractorMI355X(do_normalize=True, normalize_on_device=True)
    feats = fe(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest").to("cuda")
    x = feats["input_values"]
    assert feats._pending_normalize is None
    cols = torch.from_numpy(g["cols"]).cuda()
    assert float((x[:, cols].cpu() - torch.from_numpy(g["values"])).abs().max()) < 3e-6
    assert float((x[0].cpu() - torch.from_numpy(g["first_clip"])).abs().max()) < 3e-6
    for i, n in enumerate(lengths):
        var = float(clips[i].astype(np.float64).var())  # unit variance up to the 1e-7 epsilon under the root
        assert abs(float(x[i, :n].double().mean())) < 1e-6
        assert abs(float(x[i, :n].double().std(unbiased=False)) - (var / (var + 1e-7)) ** 0.5) < 2e-6
        assert n == 16000 or float(x[i, n:].abs().max()) == 0.0
    # bitwise reproducible, and out-of-place == in-place
    raw = fe(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest")
    xin, msk = raw["input_values"].cuda(), raw["attention_mask"].cuda()
    out = torch.empty_like(xin)
    scratch = torch.empty(int(lib().loco_normalize_scratch_bytes(4)), dtype=torch.uint8, device="cuda")
    check(lib().loco_op_normalize_waveform(ptr(xin), ptr(msk), 4, 16000, 0.0, ptr(out), ptr(scratch), scratch.numel(), stream()))
    assert torch.equal(out, x)
    # no mask: every sample counts; padding_value is honoured
    one = torch.from_numpy(clips[2]).cuda()[None].contiguous()
    o2 = torch.empty_like(one)
    check(lib().loco_op_normalize_waveform(ptr(one), None, 1, one.shape[1], -7.0, ptr(o2), ptr(scratch), scratch.numel(), stream()))
    ref = (clips[2].astype(np.float64) - clips[2].astype(np.float64).mean()) / np.sqrt(clips[2].astype(np.float64).var() + 1e-7)
    assert float((o2[0].double().cpu() - torch.from_numpy(ref)).abs().max()) < 2e-6
    m1 = torch.ones(1, one.shape[1], dtype=torch.int32, device="cuda")
    m1[0, 5000:] = 0
    check(lib().loco_op_normalize_waveform(ptr(one), ptr(m1), 1, one.shape[1], -7.0, ptr(o2), ptr(scratch), scratch.numel(), stream()))
    assert float(o2[0, 5000:].min()) == -7.0 and float(o2[0, 5000:].max()) == -7.0
    # end to end: the encoder on the device-normalised batch reproduces HF on HF's normalised batch
    from gpu_util import model
    m, _ = model()
    y = m.speecht5.encoder(**feats).last_hidden_state
    assert rel_l2(y[:, [0, 1, 20, 37, 48]], g["last_hidden_state"]) < 2e-5
    assert abs(float(y.double().norm()) / g["out_stats"][0] - 1) < 2e-5


@pytest.mark.parametrize("M,N,K", [(249, 768, 3072), (80, 2304, 768), (1, 768, 768), (512, 3072, 768), (300, 320, 64), (1992, 768, 3072),
                                   (1100, 2304, 768), (8200, 768, 768)])
@pytest.mark.parametrize("epi,out_split", [(0, False), (1, True), (2, False)])
def test_gemm_f16x3_split_k(M, N, K, epi, out_split):
    """Small problems cut K into slices (partials + fixed-order reduction with the epilogue): same accuracy bar as the
    single-pass kernel, bitwise reproducible; grids that already fill the chip, K too short and M > 8192 take the single pass."""
    A = hu("sk.a", (M, K), 2.0)
    W = hu("sk.w", (N, K), 2.0 / math.sqrt(K))
    b = hu("sk.b", (N,))
    R = hu("sk.r", (M, N))
    ahi, alo = split16(A)
    whi, wlo = split16(W)
    bd, Rd = dev(b), dev(R)
    ws = torch.empty(int(lib().loco_gemm_splitk_bytes()), dtype=torch.uint8, device="cuda")

    def run():
        C_ = torch.empty(M, N, device="cuda")
        chi = torch.empty(M, N, dtype=torch.float16, device="cuda")
        clo = torch.empty_like(chi)
        check(lib().loco_op_gemm_f16x3_splitk(ptr(ahi), ptr(alo), K, ptr(whi), ptr(wlo), K, ptr(bd), ptr(Rd) if epi == 2 else None, N,
                                              None if out_split else ptr(C_), ptr(chi) if out_split else None,
                                              ptr(clo) if out_split else None, N, M, N, K, epi, ptr(ws), ws.numel(), stream()))
        return (chi.float() + clo.float()) if out_split else C_

    out = run()
    ref = A.double() @ W.double().t() + b.double()
    if epi == 1:
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))
    if epi == 2:
        ref = ref + R.double()
    assert rel_l2(out, ref) < 5e-6
    assert torch.equal(out, run())


@pytest.mark.parametrize("B,T,frames", [(2, 700, [700, 333]), (1, 3001, None)])
def test_attention_f16x3_long_sequence_instantiation_is_bit_identical(B, T, frames):
    """Sequences of T >= 4 096 take an instantiation that skips the O rescale of a tile when alpha is exactly 1 in every lane of the
    wave (attention_f16x3.hip).  Multiplying by 1 changes nothing, so forcing either instantiation on the same input (LOCO_ATTN_LONG,
    re-read on loco_debug_reload_gemm_knobs) must give the same bits; the long tests check the long one against the oracle."""
    import os
    qkv = hu("atl.qkv", (B, T, 2304), 1.5)
    qkv[..., :768] *= 0.125 * 1.5
    pe_k = hu("atl.pe", (320, 64), 0.9)
    qh, ql = planes(qkv[..., :768].reshape(B * T, 768))
    kh, kl = planes(qkv[..., 768:1536].reshape(B * T, 768))
    vh, vl = planes(qkv[..., 1536:].reshape(B * T, 768))
    ph, pl_ = planes(pe_k * 512.0)
    frd = None if frames is None else torch.tensor(frames, dtype=torch.int32).cuda()
    outs = []
    try:
        for force in ("0", "1"):
            os.environ["LOCO_ATTN_LONG"] = force
            lib().loco_debug_reload_gemm_knobs()
            scratch = torch.empty(B, 12, T, 320, device="cuda")
            ctx = torch.empty(B, T, 768, device="cuda")
            check(lib().loco_op_attention_f16x3_pe(ptr(qh), ptr(ql), ptr(kh), ptr(kl), ptr(vh), ptr(vl), ptr(ph), ptr(pl_), 1.0 / 512.0,
                                                   ptr(scratch), ptr(frd), ptr(ctx), B, T, stream()))
            torch.cuda.synchronize()
            outs.append(ctx)
    finally:
        os.environ.pop("LOCO_ATTN_LONG", None)
        lib().loco_debug_reload_gemm_knobs()
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])

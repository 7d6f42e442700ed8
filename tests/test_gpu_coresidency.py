"""Kernels of this library must give the same bits whatever else shares their compute units: the default schedule runs two
half-batches on two streams, and a C caller may run independent forwards of one handle on several streams.

Round 1 met a kernel that did not ("the concurrency miscompare"): a conv0 variant whose taps hipcc had compiled to
`v_pk_fma_f32 ... op_sel:[0,1,0]` lost the low lane's product while workgroups of the ATTENTION kernel were resident on the
same CUs -- and only then: never alone, never beside GEMM or LayerNorm kernels.  tools/conv0_race/ holds that kernel and the
two-kernel reproducer; tests/test_isa_patterns.py bans the encoding at build time.  This file replays the co-residency that
exposed it against the SHIPPED kernels: every front-end / normalisation kernel runs beside the attention kernel on another
stream and must reproduce its solo output bit for bit."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import check, dev, la, lib, ptr


class Attention:
    """The aggressor: loco_op_attention_f16x3 on 16 x 1499 frames (72.5 KiB of LDS and 4 waves per workgroup, two per CU -- small
    workgroups of other kernels fit beside it on the same CU)."""

    def __init__(self, B=16, T=1499):
        self.B, self.T = B, T
        M = B * T
        g = torch.Generator(device="cuda").manual_seed(3)
        rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).half()
        self.q, self.k = [rn(M, 768, sc=s_) for s_ in (0.2, 2e-4)], [rn(M, 768, sc=s_) for s_ in (1.0, 1e-3)]
        self.v = [rn(B * T, 768, sc=s_) for s_ in (1.0, 1e-3)]
        self.qp = torch.randn(B, 12, T, 320, device="cuda", generator=g) * 0.5
        self.ctx = torch.empty(B, T, 768, device="cuda")

    def run(self, stream, reps):
        st = C.c_void_p(stream.cuda_stream)
        for _ in range(reps):
            check(lib().loco_op_attention_f16x3(ptr(self.q[0]), ptr(self.q[1]), ptr(self.k[0]), ptr(self.k[1]), ptr(self.v[0]), ptr(self.v[1]),
                                                ptr(self.qp), None, ptr(self.ctx), self.B, self.T, st))


def beside_attention(launch, out, trials=4, reps_attention=10, reps_victim=3):
    """launch(stream) enqueues the victim kernel writing `out`; returns the number of trials whose output differs from the solo run."""
    att = Attention()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    # the victim's inputs (and the aggressor's) were produced on the DEFAULT stream; sa / sb are not ordered behind it.  Without this
    # synchronisation the solo reference run could read a 589 MB input that torch.randn was still filling -- a sporadic 4-of-4
    # "miscompare" of this test itself (seen once in round 3, in one full-suite run of three), not of the kernels.
    torch.cuda.synchronize()
    launch(sa)
    torch.cuda.synchronize()
    ref = out.clone()
    bad = 0
    for _ in range(trials):
        out.zero_()
        torch.cuda.synchronize()
        att.run(sb, reps_attention)
        for _ in range(reps_victim):
            launch(sa)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            d = (out != ref).reshape(out.shape[0], -1) if out.dim() > 1 else (out != ref)[None]
            idx = d.nonzero()[:6].tolist()
            o2, r2 = out.reshape(d.shape), ref.reshape(d.shape)
            print(f"co-residency miscompare: {int(d.sum())} elements in {int(d.any(1).sum())} rows differ; first: "
                  + "; ".join(f"[{a},{b}] got {float(o2[a, b])!r} expected {float(r2[a, b])!r}" for a, b in idx)
                  + f"; columns hit: {d.any(0).nonzero().flatten()[:24].tolist()}")
    return bad


def test_conv0_beside_attention():
    B, L = 16, 480000
    x, _ = la.synth.batch([L] * B)
    sd = la.synth.encoder_state_dict(0, layers=0)
    p = "prenet.feature_encoder.conv_layers.0."
    xd, wd = dev(x), dev(sd[p + "conv.weight"].reshape(512, 10))
    gwd, gbd = dev(sd[p + "layer_norm.weight"]), dev(sd[p + "layer_norm.bias"])
    out = torch.empty(B, (L - 10) // 5 + 1, 512, device="cuda")
    scratch = torch.empty(lib().loco_conv0_scratch_bytes(B), dtype=torch.uint8, device="cuda")

    def launch(stream):
        check(lib().loco_op_conv0_gn_gelu(ptr(xd), B, L, ptr(wd), ptr(gwd), ptr(gbd), ptr(out), ptr(scratch), C.c_void_p(stream.cuda_stream)))

    assert beside_attention(launch, out) == 0


def test_layernorm_beside_attention():
    rows = 16 * 1499 * 8
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(rows, 768, device="cuda", generator=g) * 3.0 + 0.5
    gam, bet = torch.rand(768, device="cuda", generator=g) + 0.5, torch.randn(768, device="cuda", generator=g)
    out = torch.empty_like(x)

    def launch(stream):
        check(lib().loco_op_layernorm(ptr(x), ptr(gam), ptr(bet), ptr(out), rows, 768, 1e-5, C.c_void_p(stream.cuda_stream)))

    assert beside_attention(launch, out, reps_victim=6) == 0


def test_gemm_epilogues_beside_attention():
    """The GELU + plane-split epilogue (packed fp32 polynomial) of the FFN1-shaped GEMM."""
    M, N, K = 47968, 3072, 768
    g = torch.Generator(device="cuda").manual_seed(7)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).half()
    ahi, alo, whi, wlo = rn(M, K), rn(M, K, sc=1e-3), rn(N, K, sc=0.03), rn(N, K, sc=3e-5)
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(2, M, N, device="cuda", dtype=torch.float16)

    def launch(stream):
        check(lib().loco_op_gemm_f16x3(ptr(ahi), ptr(alo), K, ptr(whi), ptr(wlo), K, ptr(bias), None, 0, None, ptr(out[0]), ptr(out[1]), N,
                                       M, N, K, 1, 1, 1, 0, 0, 0, 0, C.c_void_p(stream.cuda_stream)))

    assert beside_attention(launch, out, reps_victim=2) == 0

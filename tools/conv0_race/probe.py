#!/usr/bin/env python3
"""Does the packed-tap conv0 kernel of commit 491647a^ miscompute BY ITSELF when other kernels share its CUs?

Round 1 saw two concurrent forwards differ from a single pass in their first clips with that kernel in place, and not after
it was rewritten -- but never located the first stage that differed.  This probe isolates the kernel: old_conv0.so
(tools/conv0_race/build.sh) exposes exactly that kernel; it runs on stream A while stream B keeps the chip busy with the
library's GEMM / LayerNorm kernels (the late-layer kernels of the other forward in the original scenario), and its fp16
planes are compared bit for bit with its own solo output.

    python tools/conv0_race/probe.py [trials]
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
la = importlib.import_module("loco-asr_amd")
L_ = importlib.import_module("loco-asr_amd._lib")
lib = L_.load()
old = C.CDLL(os.path.join(HERE, "old_conv0.so"))
old.old_conv0_planes.restype = C.c_int
old.old_conv0_planes.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
old.old_conv0_scratch_bytes.restype = C.c_size_t
old.old_conv0_scratch_bytes.argtypes = [C.c_int]

trials = int([a for a in sys.argv[1:] if not a.startswith("--")][0]) if [a for a in sys.argv[1:] if not a.startswith("--")] else 12
B, Lw = 16, 480000
T0 = (Lw - 10) // 5 + 1
sd = la.synth.encoder_state_dict(0, layers=1)
w = torch.from_numpy(sd["prenet.feature_encoder.conv_layers.0.conv.weight"].reshape(512, 10)).cuda().contiguous()
gw = torch.from_numpy(sd["prenet.feature_encoder.conv_layers.0.layer_norm.weight"]).cuda()
gb = torch.from_numpy(sd["prenet.feature_encoder.conv_layers.0.layer_norm.bias"]).cuda()
x, _ = la.synth.batch([Lw] * B)
xs = torch.from_numpy(x).cuda()
scratch = torch.empty(int(old.old_conv0_scratch_bytes(B)) + 256, dtype=torch.uint8, device="cuda")


def conv0(hi, lo, stream):
    rc = old.old_conv0_planes(xs.data_ptr(), B, Lw, w.data_ptr(), gw.data_ptr(), gb.data_ptr(), scratch.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                              stream.cuda_stream)
    assert rc == 0, rc


n = B * T0 * 512
ref_hi, ref_lo = torch.empty(n, dtype=torch.float16, device="cuda"), torch.empty(n, dtype=torch.float16, device="cuda")
hi, lo = torch.empty_like(ref_hi), torch.empty_like(ref_lo)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
conv0(ref_hi, ref_lo, sa)
torch.cuda.synchronize()

# co-runner: FFN1-shaped split GEMM (GELU epilogue, plane output) + LayerNorm, the kernels of a late encoder layer
M = 47968
ahi = torch.randn(M, 768, device="cuda").half(); alo = (torch.randn(M, 768, device="cuda") * 1e-3).half()
whi = (torch.randn(3072, 768, device="cuda") * 0.03).half(); wlo = (torch.randn(3072, 768, device="cuda") * 3e-5).half()
bias = torch.randn(3072, device="cuda")
chi = torch.empty(M, 3072, device="cuda", dtype=torch.float16); clo = torch.empty_like(chi)
lnx = torch.randn(M, 768, device="cuda"); lny = torch.empty_like(lnx); g1 = torch.ones(768, device="cuda"); b1 = torch.zeros(768, device="cuda")


def corunner(stream, reps):
    st = C.c_void_p(stream.cuda_stream)
    for _ in range(reps):
        L_.check(lib.loco_op_gemm_f16x3(ahi.data_ptr(), alo.data_ptr(), 768, whi.data_ptr(), wlo.data_ptr(), 768, bias.data_ptr(), None, 0,
                                         None, chi.data_ptr(), clo.data_ptr(), 3072, M, 3072, 768, 1, 1, 1, 0, 0, 0, 0, st))
        L_.check(lib.loco_op_layernorm(lnx.data_ptr(), g1.data_ptr(), b1.data_ptr(), lny.data_ptr(), M, 768, 1e-5, st))


def compare(tag):
    bad_hi = (hi.view(torch.int16) != ref_hi.view(torch.int16))
    bad_lo = (lo.view(torch.int16) != ref_lo.view(torch.int16))
    nb = int(bad_hi.sum()) + int(bad_lo.sum())
    if nb:
        idx = torch.nonzero(bad_hi | bad_lo).flatten()[:2000].cpu().numpy()
        clip, frame, ch = idx // (T0 * 512), (idx // 512) % T0, idx % 512
        dump = os.path.join(os.path.dirname(os.path.dirname(HERE)), "gpurun_out", "conv0_race_dump.npz")
        if not os.path.exists(dump):  # the first mismatching trial: 4000 wrong elements with what was written and what should have been
            os.makedirs(os.path.dirname(dump), exist_ok=True)
            sel = torch.nonzero(bad_hi | bad_lo).flatten()
            sel = sel[torch.linspace(0, sel.numel() - 1, min(4000, sel.numel())).long()]
            np.savez(dump, idx=sel.cpu().numpy(), got_hi=hi[sel].float().cpu().numpy(), got_lo=lo[sel].float().cpu().numpy(),
                     want_hi=ref_hi[sel].float().cpu().numpy(), want_lo=ref_lo[sel].float().cpu().numpy(), T0=T0, tag=tag)
        print(f"{tag}: {nb} differing halves; clips {sorted(set(clip.tolist()))[:8]}, frames {frame.min()}..{frame.max()} "
              f"(parity of frame: {np.bincount(frame % 2, minlength=2).tolist()}), channels mod 2: {np.bincount(ch % 2, minlength=2).tolist()}, "
              f"blocks of 64 frames: {sorted(set((frame // 64).tolist()))[:10]}", flush=True)
    return nb


# third co-runner: complete forwards of the shipped library (every kernel of the path: LDS-DMA GEMMs of all tile forms, attention,
# LayerNorm, the conv stack, memsets and copies) -- the other stream of the original scenario
pre, enc_sd = la.synth.split_state_dict(la.synth.encoder_state_dict(0))
model = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                           {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = model.speecht5.encoder
enc.range_policy = "off"
enc.streams = 1
mfull = torch.ones_like(xs, dtype=torch.int32)
enc(input_values=xs, attention_mask=mfull)
torch.cuda.synchronize()

# fourth co-runner: the attention kernel alone (72.5 KiB of LDS and 4 waves per workgroup, two per CU: conv0's small workgroups fit
# beside it on the same CU; LDS-DMA, ds_read_b128, v_exp, permlane swaps)
Ta = 1499
Tp = (Ta + 63) // 64 * 64
Ma = B * Ta
aq = [(torch.randn(Ma, 768, device="cuda") * s_).half() for s_ in (0.2, 2e-4)]
ak = [(torch.randn(Ma, 768, device="cuda") * s_).half() for s_ in (1.0, 1e-3)]
av = [(torch.randn(B * Ta, 768, device="cuda") * s_).half() for s_ in (1.0, 1e-3)]
aqp = torch.randn(B, 12, Ta, 320, device="cuda") * 0.5
actx = torch.empty(B, Ta, 768, device="cuda")


def attention_corunner(stream, reps):
    st = C.c_void_p(stream.cuda_stream)
    for _ in range(reps):
        L_.check(lib.loco_op_attention_f16x3(aq[0].data_ptr(), aq[1].data_ptr(), ak[0].data_ptr(), ak[1].data_ptr(), av[0].data_ptr(), av[1].data_ptr(),
                                              aqp.data_ptr(), None, actx.data_ptr(), B, Ta, st))


MODES = [a[2:] for a in sys.argv[1:] if a.startswith("--")] or ["solo", "busy", "attention"]
fails = {m_: 0 for m_ in ("solo", "busy", "forwards", "attention", "late_forward", "gemm8", "gemm4")}
for trial in range(trials):
    for mode in MODES:
        hi.zero_(); lo.zero_()
        torch.cuda.synchronize()
        if mode == "busy":
            corunner(sb, 6)
        if mode == "forwards":
            with torch.cuda.stream(sb):
                for _ in range(2):
                    enc(input_values=xs, attention_mask=mfull)
        if mode == "attention":
            attention_corunner(sb, 12)
        if mode in ("gemm8", "gemm4"):
            # the same GEMM forced onto a tile form whose workgroups leave room for conv0's waves on the SIMDs they occupy:
            # 256x128 / 8 waves / ~130 VGPRs (gemm8) or 128x128 / 4 waves, two workgroups per CU (gemm4).  The default 256x256 /
            # 16-wave form of the "busy" mode holds 4 x ~120 of the 512 registers of every SIMD lane: conv0 (48) never fits beside it.
            os.environ["LOCO_GEMM_TILE"] = "2" if mode == "gemm8" else "4"
            corunner(sb, 6)
            os.environ.pop("LOCO_GEMM_TILE")
        if mode == "late_forward":  # conv0 starts while the other stream's forward is deep in its encoder layers
            with torch.cuda.stream(sb):
                enc(input_values=xs, attention_mask=mfull)
            ev = torch.cuda.Event()
            with torch.cuda.stream(sb):
                pass
            torch.cuda._sleep(int(2.0e7))  # ~10 ms of spinning on the default stream: lets stream B get past its prenet
            sa.wait_stream(torch.cuda.default_stream())
        for _ in range(3):  # back to back, as in the original scenario
            conv0(hi, lo, sa)
        torch.cuda.synchronize()
        fails[mode] += 1 if compare(f"trial {trial} {mode}") else 0
print(f"old packed-tap conv0 kernel, {trials} trials each; mismatching trials by what ran on the other stream: " +
      ", ".join(f"{m_} {fails[m_]}" for m_ in MODES), flush=True)

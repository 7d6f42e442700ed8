#!/usr/bin/env python3
"""Where does the long-sequence instantiation of attention_f16x3 (O rescale skipped when alpha == 1 in every lane) start to pay?
Both instantiations on the same inputs, interleaved A B B A inside one process (LOCO_ATTN_LONG + loco_debug_reload_gemm_knobs), median
of the rounds -- a plain 'A then B' order is biased by ~2 % on this part (the second runs on a warmer, slower chip)."""
import ctypes as C, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
lib = la._lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def planes(x):
    hi = x.half()
    return hi.contiguous(), (x - hi.float()).half().contiguous()


for B, T in ((32, 1499), (12, 4000), (6, 8192), (3, 16000), (1, 29999)):
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = (torch.rand(B, T, 2304, device="cuda", generator=g) - 0.5) * 3.0
    qkv[..., :768] *= 0.125 * 1.5
    pe = (torch.rand(320, 64, device="cuda", generator=g) - 0.5) * 1.8
    qh, ql = planes(qkv[..., :768].reshape(B * T, 768)); kh, kl = planes(qkv[..., 768:1536].reshape(B * T, 768)); vh, vl = planes(qkv[..., 1536:].reshape(B * T, 768))
    ph, pl = planes(pe * 512.0)
    scratch = torch.empty(B, 12, T, 320, device="cuda"); ctx = torch.empty(B, T, 768, device="cuda")
    del qkv

    def timed(force, reps):
        os.environ["LOCO_ATTN_LONG"] = force
        lib.loco_debug_reload_gemm_knobs()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            assert lib.loco_op_attention_f16x3_pe(P(qh), P(ql), P(kh), P(kl), P(vh), P(vl), P(ph), P(pl), 1.0 / 512.0, P(scratch), None, P(ctx), B, T, st) == 0
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    reps = max(3, int(40e-3 / (1.2e-12 * B * T * T)))
    timed("0", reps); timed("1", reps)
    t = {"0": [], "1": []}
    for rnd in range(6):
        for f in (("0", "1", "1", "0") if rnd % 2 == 0 else ("1", "0", "0", "1")):
            t[f].append(timed(f, reps))
    a, b = statistics.median(t["0"]), statistics.median(t["1"])
    print(f"B={B:2d} T={T:5d}: plain {a:8.3f} ms, with the skip {b:8.3f} ms ({100 * (b / a - 1):+.2f} %)", flush=True)
os.environ.pop("LOCO_ATTN_LONG", None)

#!/usr/bin/env python3
"""A/B of the attention kernel's band-transpose scratch layout (round 4: rows at 16 r + r / 2 against rounds 1-3's rows padded to 17
floats): the shipped library against another build given on the command line, same inputs, interleaved A B B A in one process,
outputs compared bit for bit.

    python3 tools/attn_band_ab.py tools/ab/lib_band17.so"""
import ctypes as C, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
new = la._lib.load()
old = C.CDLL(os.path.abspath(sys.argv[1]))
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def planes(x):
    hi = x.half()
    return hi.contiguous(), (x - hi.float()).half().contiguous()


for B, T in ((64, 170), (64, 250), (64, 299), (32, 1499), (2, 29999)):
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = (torch.rand(B, T, 2304, device="cuda", generator=g) - 0.5) * 3.0
    qkv[..., :768] *= 0.125 * 1.5
    pe = (torch.rand(320, 64, device="cuda", generator=g) - 0.5) * 1.8
    qh, ql = planes(qkv[..., :768].reshape(B * T, 768)); kh, kl = planes(qkv[..., 768:1536].reshape(B * T, 768)); vh, vl = planes(qkv[..., 1536:].reshape(B * T, 768))
    ph, pl = planes(pe * 512.0)
    scratch = torch.empty(B, 12, T, 320, device="cuda")
    ctxs = {id(new): torch.empty(B, T, 768, device="cuda"), id(old): torch.empty(B, T, 768, device="cuda")}
    del qkv

    def timed(lib, reps):
        ctx = ctxs[id(lib)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            rc = lib.loco_op_attention_f16x3_pe(P(qh), P(ql), P(kh), P(kl), P(vh), P(vl), P(ph), P(pl), C.c_float(1.0 / 512.0), P(scratch), None, P(ctx), B, T, st)
            assert rc == 0
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    reps = max(3, int(40e-3 / (1.2e-12 * B * T * T + 2e-5)))
    timed(new, reps); timed(old, reps)
    a, b = [], []
    for _ in range(5):
        a.append(timed(new, reps)); b.append(timed(old, reps)); b.append(timed(old, reps)); a.append(timed(new, reps))
    ma, mb = statistics.median(a), statistics.median(b)
    same = torch.equal(ctxs[id(new)], ctxs[id(old)])
    print(f"B={B:3d} T={T:6d}: rows of 17 floats (old) {1e3 * mb:9.1f} us, rows at 16 r + r/2 (new) {1e3 * ma:9.1f} us ({100 * (ma / mb - 1):+.1f} %); bit-identical: {same}", flush=True)

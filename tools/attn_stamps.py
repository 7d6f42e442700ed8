#!/usr/bin/env python3
"""Where an attention workgroup's cycles go (diagnostic build: tools/ab/build_variant.sh astamps -DLOCO_ATTN_STAMPS): per loop segment,
the s_memtime cycles wave 0 of every workgroup spent in it.  Read the SHARES, not the run time of this build.

    python tools/attn_stamps.py [B T [table]]      table = the form that computes the relative-position table in its prologue
"""
import ctypes as C
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["LOCO_ASR_LIB"] = os.path.join(HERE, "ab", os.environ.get("ATTN_STAMPS_LIB", "libastamps.so"))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import torch

L_ = importlib.import_module("loco-asr_amd._lib")
lib = L_.load()
dbg = C.CDLL(os.environ["LOCO_ASR_LIB"])
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 1499)
TABLE = len(sys.argv) > 3 and sys.argv[3] == "table"
Tp = (T + 63) // 64 * 64
torch.manual_seed(0)
M = B * T
q = [(torch.randn(M, 768, device="cuda") * s_).half() for s_ in (0.2, 2e-4)]
k = [(torch.randn(M, 768, device="cuda") * s_).half() for s_ in (1.0, 1e-3)]
v = [(torch.randn(M, 768, device="cuda") * s_).half() for s_ in (1.0, 1e-3)]
qp = torch.randn(B, 12, T, 320, device="cuda") * 0.5
pe = [(torch.randn(320, 64, device="cuda") * s_).half() for s_ in (1.0, 1e-3)]
ctx = torch.empty(B, T, 768, device="cuda")
stamps = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
assert dbg.loco_debug_set_attn_stamps(C.c_void_p(stamps.data_ptr())) == 0
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run():
    if TABLE:
        L_.check(lib.loco_op_attention_f16x3_pe(q[0].data_ptr(), q[1].data_ptr(), k[0].data_ptr(), k[1].data_ptr(), v[0].data_ptr(), v[1].data_ptr(),
                                                 pe[0].data_ptr(), pe[1].data_ptr(), 0.5, qp.data_ptr(), None, ctx.data_ptr(), B, T, st))
        return
    L_.check(lib.loco_op_attention_f16x3(q[0].data_ptr(), q[1].data_ptr(), k[0].data_ptr(), k[1].data_ptr(), v[0].data_ptr(), v[1].data_ptr(),
                                          qp.data_ptr(), None, ctx.data_ptr(), B, T, st))


for _ in range(3):
    run()
torch.cuda.synchronize()
stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
s_ = stamps.cpu().numpy().reshape(-1, 8)
s_ = s_[s_[:, 6] != 0].astype(np.float64)
names = ["top: DMA issue, first K fragments, O rescale", "block 1: S(t+1) = K Q^T || exp2 + split", "blocks 2-3: O += V P || split, next max",
         "bookkeeping of tile t+1 (+ band)", "vmcnt(0) + barrier", "prologue"]
tot = s_[:, 7]
nt = s_[:, 6]
print(f"B={B} T={T}{' table form' if TABLE else ''}: {len(s_)} workgroups, event {e0.elapsed_time(e1) * 1e3:.0f} us; per workgroup: {np.median(nt):.0f} key tiles, "
      f"{np.median(tot):.0f} cycles = {np.median(tot / nt):.0f} per key tile (the MFMAs alone: 48 x 32 = 1536 per wave)")
for i, n in enumerate(names):
    print(f"   {n:52s} {np.median(s_[:, i] / nt):8.0f} cycles per key tile   {100 * np.median(s_[:, i] / tot):5.1f} %")

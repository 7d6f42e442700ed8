#!/usr/bin/env python3
"""Where a GEMM workgroup's cycles go: prologue / main loop (of which parked at the k-tile wait + barrier) / epilogue / store drain.

Needs the diagnostic build (tools/ab/build_variant.sh stamps -DLOCO_GEMM_STAMPS); wave 0 of every workgroup stamps s_memtime
at those points (gemm_f16x3.hip).  Read the SHARES, not the run time of this build.

    python tools/gemm_stamps.py [--tile=N]
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["LOCO_ASR_LIB"] = os.path.join(HERE, "ab", "libstamps.so")
sys.path.insert(0, os.path.dirname(HERE))
import importlib

import numpy as np
import torch

L = importlib.import_module("loco-asr_amd._lib")
lib = L.load()
dbg = C.CDLL(os.environ["LOCO_ASR_LIB"])
for a in sys.argv[1:]:
    if a.startswith("--tile="):
        os.environ["LOCO_GEMM_TILE"] = a.split("=")[1]
M = 47968
LDA0 = "--lda0" in sys.argv  # every A row = row 0: the A stream is served from L2 (is the k-tile time memory latency?)
LDW0 = "--ldw0" in sys.argv  # every W row = row 0 as well
shapes = [("qkv_f32out", M, 2304, 768, 0, False), ("out_proj", M, 768, 768, 2, False), ("ffn1", M, 3072, 768, 1, True), ("ffn2", M, 768, 3072, 2, False),
          ("conv2", 767968 // 32, 512, 1536, 1, True)]
torch.manual_seed(0)
stamps = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
assert dbg.loco_debug_set_gemm_stamps(C.c_void_p(stamps.data_ptr())) == 0
for name, m, n, k, epi, osplit in shapes:
    conv = name.startswith("conv")
    nb = 32 if conv else 1
    rows = nb * m * 2 + 8 if conv else m
    kk = 512 if conv else k
    ahi = torch.randn(rows, kk, device="cuda").half(); alo = (torch.randn(rows, kk, device="cuda") * 1e-3).half()
    whi = (torch.randn(n, k, device="cuda") * 0.03).half(); wlo = (torch.randn(n, k, device="cuda") * 3e-5).half()
    b = torch.randn(n, device="cuda"); R = torch.randn(nb * m, n, device="cuda")
    Cc = torch.empty(nb * m, n, device="cuda"); chi = torch.empty(nb * m, n, device="cuda", dtype=torch.float16); clo = torch.empty_like(chi)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        L.check(lib.loco_op_gemm_f16x3(ahi.data_ptr(), alo.data_ptr(), 0 if LDA0 else (2 * 512 if conv else k), whi.data_ptr(), wlo.data_ptr(), 0 if LDW0 else k,
                                       None if conv else b.data_ptr(), R.data_ptr() if epi == 2 else None, n,
                                       None if osplit else Cc.data_ptr(), chi.data_ptr() if osplit else None, clo.data_ptr() if osplit else None, n,
                                       m, n, k, epi, nb, 1, (2 * m) * 512 if conv else 0, 0, m * n if conv else 0, 0, st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    s_ = stamps.cpu().numpy().reshape(-1, 8)
    s_ = s_[s_[:, 6] != 0]
    pro, loop, stall, epi_, drain, r0, r1, xcc = [s_[:, i].astype(np.float64) for i in range(8)]
    total = pro + loop + epi_ + drain
    wall_us = (r1.max() - r0.min()) / 100.0  # s_memrealtime: 100 MHz
    life_us = (r1 - r0) / 100.0
    clk = np.median(total / np.maximum(life_us, 1e-3)) / 1e3  # cycles per ns
    nk = k // 32
    print(f"{name:11s} M={nb * m} N={n} K={k}: {len(s_)} workgroups, event {e0.elapsed_time(e1) * 1e3:.0f} us, first start -> last end {wall_us:.0f} us, "
          f"s_memtime {clk:.3f} GHz\n"
          f"   per workgroup (median cycles): prologue {np.median(pro):.0f}  main loop {np.median(loop):.0f} ({np.median(loop) / nk:.0f} per k-tile; "
          f"parked at wait+barrier {np.median(stall):.0f} = {100 * np.median(stall / loop):.0f} %)  epilogue {np.median(epi_):.0f}  drain {np.median(drain):.0f}"
          f"   -> shares {100 * np.median(pro / total):.0f} / {100 * np.median(loop / total):.0f} / {100 * np.median(epi_ / total):.0f} / {100 * np.median(drain / total):.0f} %\n"
          f"   workgroup life {np.median(life_us):.1f} us (p10 {np.percentile(life_us, 10):.1f}, p90 {np.percentile(life_us, 90):.1f}); "
          f"CU-time used {life_us.sum() / (256 * wall_us) * 100:.0f} % of 256 CUs x wall; XCDs seen {sorted(set(xcc.astype(int).tolist()))}", flush=True)
    # start-time histogram: how many rounds, how ragged the tail
    rel = (r0 - r0.min()) / 100.0
    hist, edges = np.histogram(rel, bins=12)
    print("   starts per twelfth of the launch:", hist.tolist(), flush=True)

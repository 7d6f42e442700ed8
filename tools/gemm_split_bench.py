#!/usr/bin/env python3
"""Time loco_op_gemm_f16x3 on the encoder's GEMM shapes (median over interleaved rounds)."""
import importlib, os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("loco-asr_amd._lib")
lib = L.load()
M = 47968
shapes = [("qkv", M, 2304, 768, 0, False), ("out_proj", M, 768, 768, 2, False), ("ffn1", M, 3072, 768, 1, True), ("ffn2", M, 768, 3072, 2, False),
          ("featproj", M, 768, 512, 0, False), ("conv1", 47999, 512, 1536, 1, True), ("conv4", 5999, 512, 1536, 1, True)]
torch.manual_seed(0)
bufs = {}
for name, m, n, k, epi, osplit in shapes:
    conv = name.startswith("conv")
    nb = 32 if conv else 1
    rows = nb * m * 2 + 8 if conv else m
    kk = 512 if conv else k
    ahi = (torch.randn(rows, kk, device="cuda")).half(); alo = (torch.randn(rows, kk, device="cuda") * 1e-3).half()
    whi = (torch.randn(n, k, device="cuda") * 0.03).half(); wlo = (torch.randn(n, k, device="cuda") * 3e-5).half()
    b = torch.randn(n, device="cuda"); R = torch.randn(nb * m, n, device="cuda")
    Cc = torch.empty(nb * m, n, device="cuda"); chi = torch.empty(nb * m, n, device="cuda", dtype=torch.float16); clo = torch.empty_like(chi)
    bufs[name] = (ahi, alo, whi, wlo, b, R, Cc, chi, clo, nb)
def run(name, m, n, k, epi, osplit):
    ahi, alo, whi, wlo, b, R, Cc, chi, clo, nb = bufs[name]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    conv = name.startswith("conv")
    L.check(lib.loco_op_gemm_f16x3(ahi.data_ptr(), alo.data_ptr(), 2 * 512 if conv else k, whi.data_ptr(), wlo.data_ptr(), k,
                                   None if conv else b.data_ptr(), R.data_ptr() if epi == 2 else None, n,
                                   None if osplit else Cc.data_ptr(), chi.data_ptr() if osplit else None, clo.data_ptr() if osplit else None, n,
                                   m, n, k, epi, nb, 1, (2 * m) * 512 if conv else 0, 0, m * n if conv else 0, 0, st))
res = {s[0]: [] for s in shapes}
for rnd in range(6):
    for s in shapes:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(*s); e0.record(); run(*s); run(*s); e1.record(); torch.cuda.synchronize()
        if rnd: res[s[0]].append(e0.elapsed_time(e1) / 2)
for name, m, n, k, epi, osplit in shapes:
    nb = bufs[name][9]
    t = sorted(res[name])[len(res[name]) // 2]
    print(f"{name:9s} M={nb*m:8d} N={n:5d} K={k:5d} epi={epi} split_out={int(osplit)} {t:8.3f} ms  {2.0*nb*m*n*k/t/1e9:7.1f} TFLOP/s (algorithmic)")

#!/usr/bin/env python3
"""Time loco_op_gemm on the encoder's GEMM shapes (one process, interleaved rounds, median)."""
import importlib, os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("loco-asr_amd._lib")
lib = L.load()
M = 47968
shapes = [("qkv", M, 2304, 768, 0), ("out_proj", M, 768, 768, 2), ("ffn1", M, 3072, 768, 1), ("ffn2", M, 768, 3072, 2),
          ("featproj", M, 768, 512, 0), ("conv1", 47999, 512, 1536, 1), ("conv4", 5999, 512, 1536, 1), ("conv6", 1499, 512, 1024, 1)]
torch.manual_seed(0)
bufs = {}
for name, m, n, k, epi in shapes:
    nb = 32 if name.startswith("conv") else 1
    A = torch.randn(nb * m * 2 + 8, 512 if name.startswith("conv") else k, device="cuda") if name.startswith("conv") else torch.randn(m, k, device="cuda")
    W = torch.randn(n, k, device="cuda") * 0.03
    b = torch.randn(n, device="cuda")
    R = torch.randn(nb * m, n, device="cuda")
    Cc = torch.empty(nb * m, n, device="cuda")
    bufs[name] = (A, W, b, R, Cc, nb)
def run(name, m, n, k, epi):
    A, W, b, R, Cc, nb = bufs[name]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if name.startswith("conv"):
        kk = k // 512
        tin = 2 * m + kk  # rows available per clip
        L.check(lib.loco_op_gemm(A.data_ptr(), 2 * 512, W.data_ptr(), k, None, None, n, Cc.data_ptr(), n, m, n, k, epi, nb, 1,
                                 (2 * m) * 512, 0, m * n, 0, st))
    else:
        L.check(lib.loco_op_gemm(A.data_ptr(), k, W.data_ptr(), k, b.data_ptr(), R.data_ptr() if epi == 2 else None, n, Cc.data_ptr(), n,
                                 m, n, k, epi, 1, 1, 0, 0, 0, 0, st))
res = {s[0]: [] for s in shapes}
for rnd in range(6):
    for s in shapes:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(*s); e0.record(); run(*s); run(*s); e1.record(); torch.cuda.synchronize()
        if rnd: res[s[0]].append(e0.elapsed_time(e1) / 2)
for name, m, n, k, epi in shapes:
    nb = bufs[name][5]
    t = sorted(res[name])[len(res[name]) // 2]
    print(f"{name:9s} M={nb*m:8d} N={n:5d} K={k:5d} epi={epi} {t:8.3f} ms  {2.0*nb*m*n*k/t/1e9:7.1f} TFLOP/s")

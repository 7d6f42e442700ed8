#!/usr/bin/env python3
"""Context for the f16x3 GEMM numbers: the same shapes through the vendor libraries torch dispatches to (hipBLASLt / rocBLAS)
in fp32 (the precision class this path must keep) and in plain fp16 / bf16 (which miss the 1e-3 bar: BASELINE.md)."""
import importlib, os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("loco-asr_amd._lib")
lib = L.load()
M = 47968
shapes = [("qkv", M, 2304, 768), ("ffn1", M, 3072, 768), ("ffn2", M, 768, 3072), ("out_proj", M, 768, 768)]
torch.manual_seed(0)
torch.backends.cuda.matmul.allow_tf32 = False


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, m, n, k in shapes:
    A = torch.randn(m, k, device="cuda"); W = torch.randn(n, k, device="cuda") * 0.03
    ref = (A[:256].double() @ W.double().t())
    rows = []
    y = torch.empty(m, n, device="cuda")
    t = timed(lambda: torch.matmul(A, W.t(), out=y))
    rows.append(("library fp32 (torch.matmul)", t, float((y[:256].double() - ref).norm() / ref.norm())))
    for dt, nm in ((torch.float16, "library fp16"), (torch.bfloat16, "library bf16")):
        Ah, Wh = A.to(dt), W.to(dt)
        yh = torch.empty(m, n, device="cuda", dtype=dt)
        t = timed(lambda: torch.matmul(Ah, Wh.t(), out=yh))
        rows.append((nm + " (torch.matmul)", t, float((yh[:256].double() - ref).norm() / ref.norm())))
    ahi = A.half(); alo = (A - ahi.float()).half(); whi = W.half(); wlo = (W - whi.float()).half()
    Cc = torch.empty(m, n, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: L.check(lib.loco_op_gemm_f16x3(ahi.data_ptr(), alo.data_ptr(), k, whi.data_ptr(), wlo.data_ptr(), k, None, None, n,
                                                  Cc.data_ptr(), None, None, n, m, n, k, 0, 1, 1, 0, 0, 0, 0, st))
    t = timed(call)
    rows.append(("this repo f16x3 (3 fp16 MFMAs / product)", t, float((Cc[:256].double() - ref).norm() / ref.norm())))
    for nm, t, err in rows:
        print(f"{name:9s} M={m} N={n:4d} K={k:4d}  {nm:42s} {t:7.3f} ms {2.0*m*n*k/t/1e9:7.1f} TFLOP/s  rel err {err:.1e}", flush=True)

#!/usr/bin/env python3
"""Time loco_op_gemm_f16x3 on the encoder's GEMM shapes (median over interleaved rounds)."""
import importlib, os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("loco-asr_amd._lib")
lib = L.load()
libs = {"main": lib}
tiles = {}  # variant name -> LOCO_GEMM_TILE value (the library re-reads it on every launch)
for a in sys.argv[1:]:
    if a.startswith("--tiles="):  # e.g. --tiles=0,1,2,3,4: A/B the tile forms of gemm_f16x3.hip in this process
        libs = {}
        for t in a.split("=")[1].split(","):
            libs["t" + t] = lib
            tiles["t" + t] = t
for a in sys.argv[1:]:
    if a.endswith(".so"):  # A/B other builds inside the same process (same device, same clocks); several may be given
        alt = C.CDLL(os.path.abspath(a))
        alt.loco_op_gemm_f16x3.restype = lib.loco_op_gemm_f16x3.restype
        alt.loco_op_gemm_f16x3.argtypes = lib.loco_op_gemm_f16x3.argtypes
        nm = os.path.basename(a)[3:-3] if os.path.basename(a).startswith("lib") else os.path.basename(a)[:-3]
        libs["alt" if "alt" not in libs and sum(x.endswith(".so") for x in sys.argv[1:]) == 1 else nm[:8]] = alt
envvars = {}  # variant name -> environment variable set to "1" around its calls (the library reads its knobs on every launch)
for a in sys.argv[1:]:
    if a.startswith("--envvar="):  # e.g. --envvar=LOCO_GEMM_NOCOLGROUP: A/B a run-time knob inside this process
        nm = a.split("=")[1]
        libs["+" + nm[-7:]] = lib
        envvars["+" + nm[-7:]] = nm
M = 47968
shapes = [("qkv", M, 2304, 768, 0, False), ("out_proj", M, 768, 768, 2, False), ("ffn1", M, 3072, 768, 1, True), ("ffn2", M, 768, 3072, 2, False),
          ("featproj", M, 768, 512, 0, False), ("conv1", 47999, 512, 1536, 1, True), ("conv4", 5999, 512, 1536, 1, True),
          ("qp", 1499, 320, 64, 0, False)]  # Qp[b,h] = q[b,:,h,:] pe_k^T: batched over 32 clips x 12 heads, A row stride 768
if "--pack" in sys.argv:  # the shapes of a pack of 32 pairs of ~4 s utterances: 64 clips x 250 frames (conv layers: 64 clips x 6 399 / 799 frames)
    M = 16000
    shapes = [("qkv", M, 2304, 768, 0, False), ("out_proj", M, 768, 768, 2, False), ("ffn1", M, 3072, 768, 1, True), ("ffn2", M, 768, 3072, 2, False),
              ("featproj", M, 768, 512, 0, False), ("conv1", 6399, 512, 1536, 1, True), ("conv4", 799, 512, 1536, 1, True)]
    PACK_NB = 64
if "--epi-study" in sys.argv:  # what the epilogue and the short K loop cost on the FFN1 / QKV shapes
    shapes = [("ffn1", M, 3072, 768, 1, True), ("ffn1_noepi", M, 3072, 768, 0, True), ("ffn1_f32out", M, 3072, 768, 0, False),
              ("ffn1_k1536", M, 3072, 1536, 0, False), ("ffn1_k3072", M, 3072, 3072, 0, False),
              ("ffn2", M, 768, 3072, 2, False), ("ffn2_noepi", M, 768, 3072, 0, False), ("n768_k768", M, 768, 768, 0, False),
              ("n1024_k768", M, 1024, 768, 0, False), ("n1536_k768", M, 1536, 768, 0, False)]
torch.manual_seed(0)
bufs = {}
for name, m, n, k, epi, osplit in shapes:
    if name == "qp":
        ahi = torch.randn(32 * m, 768, device="cuda").half(); alo = (torch.randn(32 * m, 768, device="cuda") * 1e-3).half()
        whi = (torch.randn(n, k, device="cuda") * 0.3).half(); wlo = (torch.randn(n, k, device="cuda") * 3e-4).half()
        bufs[name] = (ahi, alo, whi, wlo, None, None, torch.empty(32 * 12 * m, n, device="cuda"), None, None, 32 * 12)
        continue
    conv = name.startswith("conv")
    nb = (PACK_NB if "--pack" in sys.argv else 32) if conv else 1
    rows = nb * m * 2 + 8 if conv else m
    kk = 512 if conv else k
    ahi = (torch.randn(rows, kk, device="cuda")).half(); alo = (torch.randn(rows, kk, device="cuda") * 1e-3).half()
    whi = (torch.randn(n, k, device="cuda") * 0.03).half(); wlo = (torch.randn(n, k, device="cuda") * 3e-5).half()
    b = torch.randn(n, device="cuda"); R = torch.randn(nb * m, n, device="cuda")
    Cc = torch.empty(nb * m, n, device="cuda"); chi = torch.empty(nb * m, n, device="cuda", dtype=torch.float16); clo = torch.empty_like(chi)
    bufs[name] = (ahi, alo, whi, wlo, b, R, Cc, chi, clo, nb)
def run(name, m, n, k, epi, osplit, lb=lib):
    ahi, alo, whi, wlo, b, R, Cc, chi, clo, nb = bufs[name]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if name == "qp":
        L.check(lb.loco_op_gemm_f16x3(ahi.data_ptr(), alo.data_ptr(), 768, whi.data_ptr(), wlo.data_ptr(), k, None, None, 0, Cc.data_ptr(), None, None, n,
                                      m, n, k, 0, 32, 12, m * 768, 64, 12 * m * n, m * n, st))
        return
    conv = name.startswith("conv")
    L.check(lb.loco_op_gemm_f16x3(ahi.data_ptr(), alo.data_ptr(), 2 * 512 if conv else k, whi.data_ptr(), wlo.data_ptr(), k,
                                   None if conv else b.data_ptr(), R.data_ptr() if epi == 2 else None, n,
                                   None if osplit else Cc.data_ptr(), chi.data_ptr() if osplit else None, clo.data_ptr() if osplit else None, n,
                                   m, n, k, epi, nb, 1, (2 * m) * 512 if conv else 0, 0, m * n if conv else 0, 0, st))
res = {(v, s[0]): [] for s in shapes for v in libs}
for rnd in range(6):
    for s in shapes:
        for v, lb in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if v in tiles:
                os.environ["LOCO_GEMM_TILE"] = tiles[v]
            if v in envvars:
                os.environ[envvars[v]] = "1"
            lib.loco_debug_reload_gemm_knobs()  # the knobs are read once, and again on request
            run(*s, lb=lb); e0.record(); run(*s, lb=lb); run(*s, lb=lb); e1.record(); torch.cuda.synchronize()
            if rnd: res[(v, s[0])].append(e0.elapsed_time(e1) / 2)
            if v in envvars:
                os.environ.pop(envvars[v])
            if v in tiles:
                os.environ.pop("LOCO_GEMM_TILE", None)
            lib.loco_debug_reload_gemm_knobs()
for name, m, n, k, epi, osplit in shapes:
    nb = bufs[name][9]
    for v in libs:
        t = sorted(res[(v, name)])[len(res[(v, name)]) // 2]
        print(f"{v:8s} {name:11s} M={nb*m:8d} N={n:5d} K={k:5d} epi={epi} split_out={int(osplit)} {t:8.3f} ms  {2.0*nb*m*n*k/t/1e9:7.1f} TFLOP/s (algorithmic)")

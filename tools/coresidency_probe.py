#!/usr/bin/env python3
"""Details of a co-residency miscompare: LayerNorm (victim) beside the attention kernel (aggressor), per trial the number of differing
elements, which rows / columns, and the values.  Optional argument: another build of the library whose ATTENTION kernel is the aggressor
(the victim is always the shipped LayerNorm)."""
import ctypes as C, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import check, lib, ptr
L = importlib.import_module("loco-asr_amd._lib")
agg = lib()
if len(sys.argv) > 1:
    agg = C.CDLL(os.path.abspath(sys.argv[1]))
    agg.loco_op_attention_f16x3.restype = lib().loco_op_attention_f16x3.restype
    agg.loco_op_attention_f16x3.argtypes = lib().loco_op_attention_f16x3.argtypes
B, T = 16, 1499
Tp = (T + 63) // 64 * 64
M = B * T
g = torch.Generator(device="cuda").manual_seed(3)
rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).half()
q, k = [rn(M, 768, sc=s_) for s_ in (0.2, 2e-4)], [rn(M, 768, sc=s_) for s_ in (1.0, 1e-3)]
v = [rn(B * T, 768, sc=s_) for s_ in (1.0, 1e-3)]
qp = torch.randn(B, 12, T, 320, device="cuda", generator=g) * 0.5
ctx = torch.empty(B, T, 768, device="cuda")
rows = 16 * 1499 * 8
g2 = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(rows, 768, device="cuda", generator=g2) * 3.0 + 0.5
gam, bet = torch.rand(768, device="cuda", generator=g2) + 0.5, torch.randn(768, device="cuda", generator=g2)
out = torch.empty_like(x)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ln = lambda st: check(lib().loco_op_layernorm(ptr(x), ptr(gam), ptr(bet), ptr(out), rows, 768, 1e-5, C.c_void_p(st.cuda_stream)))
torch.cuda.synchronize()  # inputs were produced on the default stream
ln(sa); torch.cuda.synchronize()
ref = out.clone()
x0 = x.clone()
for trial in range(4):
    out.zero_(); torch.cuda.synchronize()
    for _ in range(10):
        assert agg.loco_op_attention_f16x3(ptr(q[0]), ptr(q[1]), ptr(k[0]), ptr(k[1]), ptr(v[0]), ptr(v[1]), ptr(qp), None, ptr(ctx), B, T, C.c_void_p(sb.cuda_stream)) == 0
    for _ in range(6):
        ln(sa)
    torch.cuda.synchronize()
    d = (out != ref)
    n = int(d.sum())
    print(f"trial {trial}: {n} differing elements in {int(d.any(1).sum())} rows; input intact: {bool(torch.equal(x, x0))}", flush=True)
    if n:
        idx = d.nonzero()[:8]
        for r_, c_ in idx.tolist():
            print(f"    row {r_} col {c_}: got {float(out[r_, c_])!r} expected {float(ref[r_, c_])!r}")
        cols = d.any(0).nonzero().flatten()
        print(f"    columns affected: {len(cols)} (first {cols[:16].tolist()}); rows first {d.any(1).nonzero().flatten()[:10].tolist()}")
# and LayerNorm alone, repeated, for run-to-run determinism
bad = 0
for _ in range(6):
    out.zero_(); torch.cuda.synchronize(); ln(sa); torch.cuda.synchronize(); bad += int(not torch.equal(out, ref))
print("LayerNorm alone, 6 runs differing from the first:", bad)

import sys, os, math, importlib, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
from gpu_util import check, dev, la, lib, ptr, stream
def hu(key, shape, scale=1.0): return torch.from_numpy(la.synth.hashed_uniform(key, shape, 11)) * scale
def split16(x):
    xd = dev(x); hi = torch.empty(xd.shape, dtype=torch.float16, device="cuda"); lo = torch.empty_like(hi)
    check(lib().loco_op_split_f16(ptr(xd), ptr(hi), ptr(lo), xd.numel(), stream())); return hi, lo
M,N,K=128,128,32
A = hu("g3.a", (M, K), 2.0); W = hu("g3.w", (N, K), 2.0 / math.sqrt(K)); b = hu("g3.b", (N,))
ahi, alo = split16(A); whi, wlo = split16(W); bd = dev(b)
outs = {}
for epi in (0,1):
  for sp in (False, True):
    C_ = torch.zeros(M, N, device="cuda"); chi = torch.zeros(M, N, dtype=torch.float16, device="cuda"); clo = torch.zeros_like(chi)
    check(lib().loco_op_gemm_f16x3(ptr(ahi), ptr(alo), K, ptr(whi), ptr(wlo), K, ptr(bd), None, N, None if sp else ptr(C_), ptr(chi) if sp else None, ptr(clo) if sp else None, N, M, N, K, epi, 1, 1, 0, 0, 0, 0, stream()))
    torch.cuda.synchronize()
    outs[(epi,sp)] = (chi.float()+clo.float()).cpu() if sp else C_.cpu()
    if sp: outs[(epi,'hi')] = chi.float().cpu(); outs[(epi,'lo')] = clo.float().cpu()
for epi in (0,1):
    d = (outs[(epi,True)] - outs[(epi,False)])
    v = outs[(epi,False)]
    print("epi", epi, "max abs diff", float(d.abs().max()), "rel", float(d.norm()/v.norm()))
    i = int(d.abs().argmax()); print("  at v=", float(v.flatten()[i]), "hi", float(outs[(epi,'hi')].flatten()[i]), "lo", float(outs[(epi,'lo')].flatten()[i]))
    small = v.abs() < 1e-3
    print("  n small", int(small.sum()), "diff on small", float(d[small].abs().max()) if small.any() else None)

#!/usr/bin/env python3
"""Diagnostic build (tools/ab/build_variant.sh attndbg -DLOCO_ATTN_DEBUG): per lane and key tile, which bookkeeping quantity of
attention_f16x3_kernel first stops being finite on the g10 layer-0 inputs.  Also runs the (q x 0.5, k x 2) / (q x 2, k x 0.5) variants:
same scores, different operand magnitudes."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import speecht5_oracle as oracle
la = importlib.import_module("loco-asr_amd")
L = importlib.import_module("loco-asr_amd._lib")
lib = L.load()
dbg = C.CDLL(os.path.join(ROOT, "tools", "ab", "libattndbg.so"))
dbg.loco_op_attention_f16x3.restype = lib.loco_op_attention_f16x3.restype
dbg.loco_op_attention_f16x3.argtypes = lib.loco_op_attention_f16x3.argtypes
dbg.loco_debug_set_attn_dbg.argtypes = [C.c_void_p]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
sd = la.synth.encoder_state_dict_hf_init(0)
x, msk = la.synth.batch([80000], first_index=40)
hs = []
oracle.encode(x, msk, sd, hidden_states=hs)
h0 = hs[0].cuda()
lp = "wrapped_encoder.layers.0."
w = lambda k: torch.from_numpy(sd[lp + k]).cuda()
B, T, _ = h0.shape
q0 = F.linear(h0, w("attention.q_proj.weight"), w("attention.q_proj.bias")) * 0.125
k0 = F.linear(h0, w("attention.k_proj.weight"), w("attention.k_proj.bias"))
v0 = F.linear(h0, w("attention.v_proj.weight"), w("attention.v_proj.bias"))
pe = torch.from_numpy(sd["wrapped_encoder.embed_positions.pe_k.weight"]).cuda()
pl = lambda t: (t.half().contiguous(), (t - t.half().float()).half().contiguous())
Tp = (T + 63) // 64 * 64
nqb = (T + 127) // 128
nblk = nqb * 12 * B


def run(q, k, v, which, label):
    qp = (q.view(B, T, 12, 64).transpose(1, 2) @ pe.t()).contiguous()
    ctx = torch.empty(B, T, 768, device="cuda")
    qh, ql = pl(q.reshape(B * T, 768)); kh, kl = pl(k.reshape(B * T, 768))
    vh, vl = pl(v.reshape(B * T, 768))
    assert which.loco_op_attention_f16x3(P(qh), P(ql), P(kh), P(kl), P(vh), P(vl), P(qp), None, P(ctx), B, T, st()) == 0
    torch.cuda.synchronize()
    bad = (~torch.isfinite(ctx)).view(B, T, 12, 64).any(-1).nonzero()
    print(f"{label:50s}: {len(bad):3d} non-finite (row, head) pairs {bad[:10, 1:].tolist()}", flush=True)
    return bad


run(q0, k0, v0, lib, "shipped library")
run(q0 * 0.5, k0 * 2, v0, lib, "shipped, q x 0.5 and k x 2 (same scores)")
run(q0 * 2, k0 * 0.5, v0, lib, "shipped, q x 2 and k x 0.5 (same scores)")
run(q0 * 0.7, k0, v0, lib, "shipped, q x 0.7")
run(q0 * 0.85, k0, v0, lib, "shipped, q x 0.85")
buf = torch.zeros(nblk * 256 * 8 * 8, device="cuda")
assert dbg.loco_debug_set_attn_dbg(C.c_void_p(buf.data_ptr())) == 0
bad = run(q0, k0, v0, dbg, "diagnostic build")
rec = buf.view(nblk, 256, 8, 8).cpu()
# the kernel's work map: w = XCD-aware permutation of blockIdx; recover (qblk, head) of every workgroup the way the kernel does
def decode(blk):
    q8, r8 = nblk >> 3, nblk & 7
    xcd, idx = blk & 7, blk >> 3
    wv = (xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8) + idx
    return wv % nqb, (wv // nqb) % 12
names = {1: "S(t+1)", 2: "O", 4: "l_run", 8: "alpha", 16: "dsh"}
for (bb, tt, hh) in bad[:10].tolist():
    for blk in range(nblk):
        qb, hd = decode(blk)
        if hd == hh and qb == tt // 128:
            wave, r = (tt % 128) // 32, tt % 32
            for half in (0, 1):
                tid = wave * 64 + half * 32 + r
                rows = rec[blk, tid]
                out = []
                for t in range(4):
                    m = int(rows[t, 0].view(torch.int32)) & 0xffff
                    out.append(f"t{t}: bad={[n for b_, n in names.items() if m & b_]} m_run={rows[t,1]:.3f} alpha={rows[t,2]:.3e} dsh={rows[t,3]:.3f} l={rows[t,4]:.3e} "
                               f"S(t+1) max={rows[t,5]:.2f} min={rows[t,6]:.2f} o0[0]={rows[t,7]:.3e}")
                print(f" frame {tt} head {hh} lane half {half}:\n    " + "\n    ".join(out))

#!/usr/bin/env python3
"""Which property of the g10 layer-0 attention inputs makes loco_op_attention_f16x3 return NaN rows?  Variants of the same call."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import speecht5_oracle as oracle
la = importlib.import_module("loco-asr_amd")
L = importlib.import_module("loco-asr_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
sd = la.synth.encoder_state_dict_hf_init(0)
x, msk = la.synth.batch([80000, 52000], first_index=40)
hs = []
oracle.encode(x[:1], msk[:1], sd, hidden_states=hs)   # CPU oracle: the layer-0 input of clip 0
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


def run(q, k, v, qp_scale=1.0, label="", f32=False, lo_zero=False):
    qp = (q.view(B, T, 12, 64).transpose(1, 2) @ pe.t()).contiguous() * qp_scale
    ctx = torch.empty(B, T, 768, device="cuda")
    if f32:
        qkv = torch.cat([q, k, v], dim=-1).contiguous()
        assert lib.loco_op_attention(P(qkv), P(qp), None, P(ctx), B, T, st()) == 0
    else:
        qh, ql = pl(q.reshape(B * T, 768)); kh, kl = pl(k.reshape(B * T, 768))
        vh, vl = pl(v.reshape(B * T, 768))
        if lo_zero:
            ql.zero_(); kl.zero_(); vl.zero_()
        assert lib.loco_op_attention_f16x3(P(qh), P(ql), P(kh), P(kl), P(vh), P(vl), P(qp), None, P(ctx), B, T, st()) == 0
    torch.cuda.synchronize()
    bad = (~torch.isfinite(ctx)).view(B, T, 12, 64).any(-1).nonzero()
    print(f"{label:58s}: {len(bad):3d} non-finite (row, head) pairs {bad[:6, 1:].tolist()}", flush=True)
    return bad


bad = run(q0, k0, v0, label="as is (clip 0, T = 249, no mask)")
run(q0, k0, v0, f32=True, label="the exact-fp32 attention kernel on the same inputs")
run(q0, k0, v0, qp_scale=0.0, label="relative-position table = 0")
run(q0, k0, torch.ones_like(v0), label="v = 1")
run(q0, k0, v0 * 0.01, label="v x 0.01")
run(q0 * 0.5, k0, v0, label="q x 0.5")
run(q0, k0 * 0.25, v0, label="k x 0.25")
run(q0, k0, v0, lo_zero=True, label="all lo planes zero")
print("max|q| %.3g max|k| %.3g max|v| %.3g" % (float(q0.abs().max()), float(k0.abs().max()), float(v0.abs().max())))
# per failing pair: the largest |v| among keys, largest k, and the p-weighted picture
qd, kd, vd = (t.view(B, T, 12, 64).transpose(1, 2).double().cpu() for t in (q0, k0, v0))
for (bb, tt, hh) in bad[:10].tolist():
    sc = qd[bb, hh, tt] @ kd[bb, hh].t()
    rel = (tt - torch.arange(T)).clamp(-160, 159) + 160
    sc = sc + (qd[bb, hh, tt] @ pe.double().cpu().t())[rel]
    p = torch.softmax(sc, 0)
    top = torch.topk(p, 3)
    print(f"  frame {tt} head {hh}: max|q row| {float(qd[bb, hh, tt].abs().max()):.3g}, score max {float(sc.max()):.2f} at key {int(sc.argmax())}, "
          f"top-3 p {[round(float(a), 4) for a in top.values]} at keys {top.indices.tolist()}, max|k| of that head {float(kd[bb, hh].abs().max()):.3g}, "
          f"max|v| of that head {float(vd[bb, hh].abs().max()):.3g}")
# how many (row, head) pairs have a one-hot softmax (p_max > 0.999999)?
sc_all = qd @ kd.transpose(-1, -2)
print("pairs with max p > 1 - 1e-7:", int((torch.softmax(sc_all, -1).max(-1).values > 1 - 1e-7).sum()), "of", B * 12 * T)

#!/usr/bin/env python3
"""Locate errors of loco_op_attention_f16x3 against an fp64 torch evaluation (debugging aid)."""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
lib = la._lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
planes = lambda x: (x.half().contiguous(), (x - x.half().float()).half().contiguous())


def run(B, T, nvalid=None, zero_bias=False):
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = (torch.rand(B, T, 2304, device="cuda", generator=g) - 0.5) * 3.0
    qkv[..., :768] *= 0.125 * 1.5
    pe = (torch.rand(320, 64, device="cuda", generator=g) - 0.5) * (0.0 if zero_bias else 1.8)
    q = qkv[..., :768].view(B, T, 12, 64).transpose(1, 2)
    qp = (q @ pe.t()).contiguous()
    Tp = (T + 63) // 64 * 64
    qh, ql = planes(qkv[..., :768].reshape(B * T, 768))
    kh, kl = planes(qkv[..., 768:1536].reshape(B * T, 768))
    vh, vl = planes(qkv[..., 1536:].reshape(B * T, 768))  # row-major like q and k
    fr = None if nvalid is None else torch.tensor(nvalid, dtype=torch.int32, device="cuda")
    ctx = torch.zeros(B, T, 768, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.loco_op_attention_f16x3(P(qh), P(ql), P(kh), P(kl), P(vh), P(vl), P(qp), P(fr), P(ctx), B, T, st) == 0
    torch.cuda.synchronize()
    qd = (qh.double() + ql.double()).view(B, T, 12, 64).transpose(1, 2)
    kd = (kh.double() + kl.double()).view(B, T, 12, 64).transpose(1, 2)
    vd = (vh.double() + vl.double()).view(B, T, 12, 64).transpose(1, 2)
    i = torch.arange(T, device="cuda")
    rel = (i[:, None] - i[None, :]).clamp(-160, 159) + 160
    s = qd @ kd.transpose(2, 3) + torch.gather(qp.double(), 3, rel[None, None].expand(B, 12, T, T))
    if fr is not None:
        s = s.masked_fill(i[None, None, None, :] >= fr[:, None, None, None].long(), float("-inf"))
    ref = (torch.softmax(s, -1) @ vd).transpose(1, 2).reshape(B, T, 768)
    err = (ctx.double() - ref)
    nan = torch.isnan(ctx)
    print(f"B={B} T={T} nvalid={nvalid} zero_bias={zero_bias}: nan count {int(nan.sum())} of {ctx.numel()}")
    e = torch.where(nan, torch.full_like(err, 1e9), err).abs()
    rows = e.view(B, T, 12, 64).amax(-1)  # [B,T,12]
    bad = (rows > 1e-4)
    print("  bad rows per head:", bad.sum(1).tolist())
    if bad.any():
        idx = bad.nonzero()[:12].tolist()
        print("  first bad (b, t, head):", idx)
        tb = bad[0, :, 0].nonzero().flatten().tolist()
        print("  head0 bad t:", tb[:40], "..." if len(tb) > 40 else "")
        b0, t0, h0 = idx[0]
        print("  got", ctx[b0, t0, h0 * 64:h0 * 64 + 8].tolist())
        print("  ref", ref[b0, t0, h0 * 64:h0 * 64 + 8].tolist())
    ok = ~bad
    print(f"  rel_l2 over good rows: {float((err.view(B,T,12,64)[ok]).norm() / ref.view(B,T,12,64)[ok].norm()):.3e}")


if __name__ == "__main__":
    run(1, 64)
    run(1, 128)
    run(1, 200)
    run(1, 200, zero_bias=True)
    run(1, 640, zero_bias=True)
    run(1, 640)
    run(1, 700, [650])

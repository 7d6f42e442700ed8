#!/usr/bin/env python3
"""Time loco_op_attention_f16x3 alone (30 s x 32 and one 10 min clip) and check it against an fp64 torch evaluation."""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
lib = la._lib.load()
if len(sys.argv) > 1:  # A/B another build of the library inside the same process (same device, same clocks)
    alt = C.CDLL(os.path.abspath(sys.argv[1]))
    for name in ("loco_op_attention_f16x3",):
        getattr(alt, name).restype = getattr(lib, name).restype
        getattr(alt, name).argtypes = getattr(lib, name).argtypes
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def planes(x):
    hi = x.half()
    return hi.contiguous(), (x - hi.float()).half().contiguous()


def run(B, T, reps, check=False, ragged=False, use_alt=False):
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = (torch.rand(B, T, 2304, device="cuda", generator=g) - 0.5) * 3.0
    qkv[..., :768] *= 0.125 * 1.5
    pe = (torch.rand(320, 64, device="cuda", generator=g) - 0.5) * 1.8
    q = qkv[..., :768].view(B, T, 12, 64).transpose(1, 2)
    qp = (q @ pe.t()).contiguous()
    Tp = (T + 63) // 64 * 64
    qh, ql = planes(qkv[..., :768].reshape(B * T, 768))
    kh, kl = planes(qkv[..., 768:1536].reshape(B * T, 768))
    vh, vl = planes(qkv[..., 1536:].reshape(B * T, 768))  # row-major like q and k
    fr = None
    if ragged:
        fr = torch.tensor([T - (i * 37) % (T // 2) for i in range(B)], dtype=torch.int32, device="cuda")
    ctx = torch.empty(B, T, 768, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = alt if use_alt else lib
    call = lambda: L.loco_op_attention_f16x3(P(qh), P(ql), P(kh), P(kl), P(vh), P(vl), P(qp), P(fr), P(ctx), B, T, st)
    assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 4.0 * B * 12 * T * T * 64
    msg = f"{'alt ' if use_alt else 'main'} B={B} T={T} ragged={ragged}: {ms:.3f} ms  {fl / ms * 1e-9:.1f} TFLOP/s algorithmic  checksum {float(ctx.double().abs().sum()):.9e}"
    if check:
        qd = (qh.double() + ql.double()).view(B, T, 12, 64).transpose(1, 2)
        kd = (kh.double() + kl.double()).view(B, T, 12, 64).transpose(1, 2)
        vd = (vh.double() + vl.double()).view(B, T, 12, 64).transpose(1, 2)
        i = torch.arange(T, device="cuda")
        rel = (i[:, None] - i[None, :]).clamp(-160, 159) + 160
        bias = torch.gather(qp.double(), 3, rel[None, None].expand(B, 12, T, T))
        s = qd @ kd.transpose(2, 3) + bias
        if fr is not None:
            s = s.masked_fill(i[None, None, None, :] >= fr[:, None, None, None].long(), float("-inf"))
        ref = (torch.softmax(s, -1) @ vd).transpose(1, 2).reshape(B, T, 768)
        msg += f"  rel_l2 vs fp64 {float((ctx.double() - ref).norm() / ref.norm()):.3e}"
    print(msg, flush=True)


if __name__ == "__main__":
    run(2, 700, 3, check=True, ragged=True)
    run(1, 1499, 3, check=True)
    variants = [False, True] if len(sys.argv) > 1 else [False]
    if len(sys.argv) > 1:
        run(2, 700, 3, check=True, ragged=True, use_alt=True)
    for rep in range(3 if len(sys.argv) > 1 else 1):
        for v in variants:
            run(32, 1499, 40, use_alt=v)
        for v in variants:
            run(1, 29999, 5, use_alt=v)
    run(32, 1499, 20, ragged=True)

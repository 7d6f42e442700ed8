#!/usr/bin/env python3
"""Is the chip power-limited on this path?  

Starts tools/power_sampler.py as a sidecar PROCESS (amdsmi: socket power, per-XCD gfx clocks, hotspot temperature, PPT / thermal
throttle residency counters at ~25 Hz) before touching the GPU, then runs timed loops of ~5 s each, with idle gaps between them:

    idle | FFN1 GEMM | the same instruction stream on all-zero operands | QKV GEMM | conv1 GEMM | attention at T = 29 999 |
    whole forward 30 s x 32 in f16x3 | in f32 | in f16x2

and joins the two clocks: per phase the achieved rate, mean / max socket power against the cap, mean gfx clock, energy per call and
per algorithmic FLOP.  Writes OUT/power_samples.jsonl (raw series), OUT/power_phases.json and OUT/power_summary.txt.

    python3 tools/power_probe.py gpurun_out/power [seconds per phase]
"""
import ctypes as C
import importlib
import json
import os
import signal
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "power")
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
os.makedirs(out_dir, exist_ok=True)
samples_path = os.path.join(out_dir, "power_samples.jsonl")
sampler = subprocess.Popen([sys.executable, os.path.join(HERE, "power_sampler.py"), samples_path, "25"])
time.sleep(1.0)

import torch  # noqa: E402  (after the sidecar is up)

sys.path.insert(0, ROOT)
la = importlib.import_module("loco-asr_amd")
L = importlib.import_module("loco-asr_amd._lib")
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
phases = []


def loop(name, call, flops_per_call, mfma_per_flop, seconds=SECONDS, note=""):
    """Run `call` back to back for ~`seconds` (batches of calls between host syncs so that the queue never drains)."""
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    per = max(1, int(0.25 / max(1e-6, e0.elapsed_time(e1) * 1e-3)))
    n = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(per):
            call()
        torch.cuda.synchronize()
        n += per
    t1 = time.time()
    phases.append(dict(name=name, t0=t0, t1=t1, calls=n, ms_per_call=(t1 - t0) / n * 1e3, flops_per_call=flops_per_call,
                       tflops=flops_per_call * n / (t1 - t0) / 1e12, mfma_per_flop=mfma_per_flop, note=note))
    print(f"{name}: {n} calls, {(t1 - t0) / n * 1e3:.3f} ms each, {flops_per_call * n / (t1 - t0) / 1e12:.1f} TFLOP/s algorithmic", flush=True)
    time.sleep(2.0)  # idle gap: the series shows the ramp down / up


def idle(name, seconds=3.0):
    t0 = time.time()
    time.sleep(seconds)
    phases.append(dict(name=name, t0=t0, t1=time.time(), calls=0, ms_per_call=None, flops_per_call=0, tflops=0, mfma_per_flop=0, note="no GPU work"))


torch.zeros(1, device="cuda")
idle("idle (context up)")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
M = 47968


def gemm_phase(name, m, n, k, epi, osplit, conv=False, zeros=False):
    nb = 32 if conv else 1
    rows = nb * m * 2 + 8 if conv else m
    kk = 512 if conv else k
    mk = (lambda *s, scale=1.0: torch.zeros(*s, device="cuda").half()) if zeros else (lambda *s, scale=1.0: (torch.randn(*s, device="cuda") * scale).half())
    ahi, alo = mk(rows, kk), mk(rows, kk, scale=1e-3)
    whi, wlo = mk(n, k, scale=0.03), mk(n, k, scale=3e-5)
    b = torch.zeros(n, device="cuda") if zeros else torch.randn(n, device="cuda")
    Cc = torch.empty(nb * m, n, device="cuda")
    chi = torch.empty(nb * m, n, device="cuda", dtype=torch.float16)
    clo = torch.empty_like(chi)

    def call():
        L.check(lib.loco_op_gemm_f16x3(P(ahi), P(alo), 2 * 512 if conv else k, P(whi), P(wlo), k, None if conv else P(b), None, n,
                                       None if osplit else P(Cc), P(chi) if osplit else None, P(clo) if osplit else None, n,
                                       m, n, k, epi, nb, 1, (2 * m) * 512 if conv else 0, 0, m * n if conv else 0, 0, st))
    loop(name, call, 2.0 * nb * m * n * k, 3, note=f"M={nb * m} N={n} K={k} epilogue={epi} plane output={osplit}" + (", all operands zero" if zeros else ""))
    del ahi, alo, whi, wlo, Cc, chi, clo
    torch.cuda.empty_cache()


gemm_phase("gemm_f16x3 FFN1", M, 3072, 768, 1, True)
gemm_phase("gemm_f16x3 FFN1, zero operands", M, 3072, 768, 1, True, zeros=True)
gemm_phase("gemm_f16x3 QKV-shaped (N=2304, plane output)", M, 2304, 768, 0, True)
gemm_phase("gemm_f16x3 conv1", 47999, 512, 1536, 1, True, conv=True)


def attention_phase(B, T):
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = (torch.rand(B, T, 2304, device="cuda", generator=g) - 0.5) * 3.0
    qkv[..., :768] *= 0.125 * 1.5
    pe = (torch.rand(320, 64, device="cuda", generator=g) - 0.5) * 1.8
    qp = (qkv[..., :768].view(B, T, 12, 64).transpose(1, 2) @ pe.t()).contiguous()
    Tp = (T + 63) // 64 * 64
    pl = lambda x: (x.half().contiguous(), (x - x.half().float()).half().contiguous())  # noqa: E731
    qh, ql = pl(qkv[..., :768].reshape(B * T, 768))
    kh, kl = pl(qkv[..., 768:1536].reshape(B * T, 768))
    vh, vl = pl(qkv[..., 1536:].reshape(B * T, 768))
    ctx = torch.empty(B, T, 768, device="cuda")
    del qkv

    def call():
        assert lib.loco_op_attention_f16x3(P(qh), P(ql), P(kh), P(kl), P(vh), P(vl), P(qp), None, P(ctx), B, T, st) == 0
    loop(f"attention_f16x3 B={B} T={T}", call, 4.0 * B * 12 * T * T * 64, 3)


attention_phase(2, 29999)
torch.cuda.empty_cache()

sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
model = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                         {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = model.speecht5.encoder
x_np, m_np = la.synth.batch([480000] * 32)
x, m = torch.from_numpy(x_np).cuda(), torch.from_numpy(m_np).cuda()
enc.range_policy = "off"  # no host synchronisation between forwards: the queue stays full
FL = 32 * (1499 * 284.2e6 + 36864.0 * 1499 * 1499)
for prec, mf in (("f16x3", 3), ("f32", 1), ("f16x2", 2)):
    enc.precision = prec
    loop(f"whole forward 30 s x 32, {prec}", lambda: enc(input_values=x, attention_mask=m), FL, mf)
idle("idle (end)")

sampler.send_signal(signal.SIGTERM)
try:
    sampler.wait(timeout=10)
except subprocess.TimeoutExpired:
    sampler.kill()

# ---- join the two clocks -----------------------------------------------------------------------------------------------------
rows = [json.loads(l) for l in open(samples_path) if l.strip()]
meta, series = rows[0], rows[1:]
summary = {"source": meta.get("source"), "meta": meta.get("meta"), "sampler_errors": meta.get("errors"), "phases": []}
lines = []
if not series:
    lines.append(f"NO TELEMETRY: {meta}")
else:
    ngpu = len(series[0]["gpus"])

    def col(g, key, t0, t1):
        v = []
        for r in series:
            if t0 <= r["t"] <= t1 and g < len(r["gpus"]):
                x_ = r["gpus"][g].get(key)
                if isinstance(x_, list):
                    xs = [a for a in x_ if isinstance(a, (int, float)) and a > 0]
                    x_ = sum(xs) / len(xs) if xs else None
                if isinstance(x_, (int, float)):
                    v.append(float(x_))
        return v
    # the busy GPU = the one whose power is highest during the first GEMM phase
    ph1 = phases[1]
    means = [(sum(col(g, "current_socket_power", ph1["t0"], ph1["t1"]) or [0]) / max(1, len(col(g, "current_socket_power", ph1["t0"], ph1["t1"]))), g) for g in range(ngpu)]
    g = max(means)[1]
    cap = None
    try:
        cap = meta["meta"][g]["power_cap"]
    except (KeyError, IndexError, TypeError):
        pass
    lines.append(f"telemetry source: {meta.get('source')}, {len(series)} samples, {ngpu} GPU(s) visible to the driver, busy GPU index {g}; power cap info: {cap}")
    lines.append(f"{'phase':58s} {'ms/call':>9s} {'TFLOP/s':>8s} {'P mean':>7s} {'P max':>6s} {'gfxclk':>7s} {'clk min':>7s} {'hot C':>6s} {'J/call':>8s} {'pJ/FLOP':>8s} {'pJ/MFMA-FLOP':>12s} {'PPT acc':>9s}")
    for ph in phases:
        a, b = ph["t0"] + 0.5, ph["t1"]  # skip the ramp of the first half second
        pw = col(g, "current_socket_power", a, b)
        ck = col(g, "current_gfxclks", a, b) or col(g, "current_gfxclk", a, b)
        tmp = col(g, "temperature_hotspot", a, b)
        ppt = col(g, "ppt_residency_acc", a, b)
        en = col(g, "energy_accumulator", a, b)
        d = dict(ph)
        d.update(samples=len(pw), power_mean_w=sum(pw) / len(pw) if pw else None, power_max_w=max(pw) if pw else None,
                 gfxclk_mean_mhz=sum(ck) / len(ck) if ck else None, gfxclk_min_mhz=min(ck) if ck else None,
                 hotspot_c=sum(tmp) / len(tmp) if tmp else None, ppt_residency_delta=(ppt[-1] - ppt[0]) if len(ppt) > 1 else None,
                 energy_accumulator_delta=(en[-1] - en[0]) if len(en) > 1 else None)
        jc = pj = pjm = None
        if pw and ph["calls"]:
            jc = d["power_mean_w"] * ph["ms_per_call"] * 1e-3
            pj = jc / ph["flops_per_call"] * 1e12
            pjm = pj / max(1, ph["mfma_per_flop"])
        d.update(joule_per_call=jc, pj_per_algorithmic_flop=pj, pj_per_issued_mfma_flop=pjm)
        summary["phases"].append(d)
        f = lambda v, w, p=1: (f"{v:{w}.{p}f}" if isinstance(v, (int, float)) else " " * (w - 1) + "-")  # noqa: E731
        lines.append(f"{ph['name']:58s} {f(ph['ms_per_call'], 9, 3)} {f(ph['tflops'], 8)} {f(d['power_mean_w'], 7)} {f(d['power_max_w'], 6, 0)} "
                     f"{f(d['gfxclk_mean_mhz'], 7, 0)} {f(d['gfxclk_min_mhz'], 7, 0)} {f(d['hotspot_c'], 6)} {f(jc, 8, 3)} {f(pj, 8, 3)} {f(pjm, 12, 3)} {f(d['ppt_residency_delta'], 9, 0)}")
json.dump(summary, open(os.path.join(out_dir, "power_phases.json"), "w"), indent=1)
open(os.path.join(out_dir, "power_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

#!/usr/bin/env python3
"""Energy per FLOP of the two fp16 MFMA shapes on a power-capped MI355X: pure register-resident matrix-pipe loops (mfma_probe.hip),
~4 s each, with the telemetry sidecar (tools/power_sampler.py).  Operand scale 0 = all-zero operands (no data toggling).

    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/mfma_probe/mfma_probe.hip -o tools/mfma_probe/libmfma_probe.so
    python3 tools/mfma_probe/run.py gpurun_out/mfma_probe
"""
import ctypes as C, json, os, signal, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_probe"
os.makedirs(out_dir, exist_ok=True)
samples = os.path.join(out_dir, "samples.jsonl")
sampler = subprocess.Popen([sys.executable, os.path.join(HERE, "..", "power_sampler.py"), samples, "25"])
time.sleep(1.0)
import torch  # noqa: E402
lib = C.CDLL(os.path.join(HERE, "libmfma_probe.so"))
lib.mfma_probe_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_uint]
out = torch.zeros(16, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
phases = []
BLOCKS, ITERS = 256 * 2, 20000   # two workgroups of 4 waves per CU: 2 waves per SIMD
flops_per_launch = BLOCKS * 4 * ITERS * 16 * (2.0 * 16 * 16 * 32)
for shape, scale, mask in ((16, 1.0, 0xffff), (32, 1.0, 0xffff), (16, 0.0, 0xffff), (32, 0.0, 0xffff), (16, 1.0, 0xfff8), (16, 1.0, 0xffe0),
                           (16, 1.0, 0xff80), (16, 1.0, 0xfc00), (16, 1.0, 0xffff)):
    assert lib.mfma_probe_run(shape, BLOCKS, 100, scale, C.c_void_p(out.data_ptr()), st, mask) == 0
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        for _ in range(4):
            assert lib.mfma_probe_run(shape, BLOCKS, ITERS, scale, C.c_void_p(out.data_ptr()), st, mask) == 0
        torch.cuda.synchronize(); n += 4
    t1 = time.time()
    phases.append(dict(shape=shape, scale=scale, mask=mask, t0=t0, t1=t1, tflops=flops_per_launch * n / (t1 - t0) / 1e12))
    print(f"shape {shape}x{shape}, operand scale {scale}: {phases[-1]['tflops']:.0f} TFLOP/s", flush=True)
    time.sleep(1.5)
sampler.send_signal(signal.SIGTERM); sampler.wait(timeout=10)
rows = [json.loads(l) for l in open(samples) if l.strip()][1:]
lines = []
for ph in phases:
    sel = [r["gpus"][0] for r in rows if ph["t0"] + 0.7 <= r["t"] <= ph["t1"]]
    pw = [g["current_socket_power"] for g in sel if isinstance(g.get("current_socket_power"), (int, float))]
    ck = [sum(c for c in g["current_gfxclks"] if c) / max(1, sum(1 for c in g["current_gfxclks"] if c)) for g in sel if g.get("current_gfxclks")]
    p = sum(pw) / len(pw) if pw else float("nan"); c = sum(ck) / len(ck) if ck else float("nan")
    lines.append(f"v_mfma_f32_{ph['shape']}x{ph['shape']}x{32 if ph['shape'] == 16 else 16}_f16, operands {'random' if ph['scale'] else 'zero  '}, B bits kept 0x{ph['mask']:04x}: {ph['tflops']:7.0f} TFLOP/s  "
                 f"{p:7.1f} W  {c:6.0f} MHz  {p / ph['tflops']:.3f} pJ/FLOP  ({ph['tflops'] * 1e12 / (256 * 4 * c * 1e6) if c == c else float('nan'):.0f} FLOP per SIMD cycle; dense peak 2048)")
open(os.path.join(out_dir, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

#!/usr/bin/env python3
"""ONE trace that tells why a hipGraph replay of the 5 s utterance was slower than the eager forward.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hg_trace -- python3 tools/hipgraph_trace.py
    python3 tools/hipgraph_trace.py --analyse gpurun_out/hg_trace          (no GPU: reads the kernel trace CSV)

The run has three phases separated by 0.6 s of sleep: warm-up + capture, N eager forwards, N graph replays -- each forward followed by
a host synchronisation, like the Python module's default policy.  The analysis splits the kernel trace at the sleeps and prints,
per phase: forwards, kernels per forward, the SUM of kernel durations per forward, the sum of the GAPS between consecutive kernels
inside a forward, and the per-kernel-name mean durations side by side -- either the kernels run slower inside the graph or the gaps
grow.
"""
import csv
import glob
import os
import sys
import time

N = 40


def run():
    import importlib
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    la = importlib.import_module("loco-asr_amd")
    sd = la.synth.encoder_state_dict(0)
    pre, encsd = la.synth.split_state_dict(sd)
    m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                         {k: torch.from_numpy(v) for k, v in encsd.items()})
    enc = m.to("cuda").speecht5.encoder
    x, msk = la.synth.batch([80000])
    xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from graph_capture import CapturedForward
    for _ in range(3):  # warm-up
        enc(input_values=xs, attention_mask=ms)
    cap = CapturedForward(enc, xs, ms)  # capture
    for _ in range(3):
        cap.replay(); torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    for flag in (False, True):
        time.sleep(0.6)
        t0 = time.perf_counter()
        for _ in range(N):
            if flag:
                cap.replay(xs, ms); torch.cuda.current_stream().synchronize(); assert cap.status_code() == 0
            else:
                enc(input_values=xs, attention_mask=ms)
        torch.cuda.synchronize()
        print(f"{'graph replay' if flag else 'eager'}: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per forward (host clock, {N} forwards)", flush=True)
    time.sleep(0.6)


def analyse(folder):
    files = glob.glob(os.path.join(folder, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {folder}")
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
    rows.sort()
    # phases: split where the GPU was idle for > 0.4 s
    phases, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - a[1] > 400_000_000:
            phases.append(cur)
            cur = []
        cur.append(b)
    phases.append(cur)
    print(f"{len(rows)} kernel records, {len(phases)} phases (split at idle gaps > 0.4 s): {[len(p) for p in phases]} kernels")
    if len(phases) < 3:
        raise SystemExit("expected >= 3 phases (warm-up + capture, eager, replay)")
    names = ("eager", "graph replay")
    stats = {}
    for name, ph in zip(names, phases[-2:]):
        per = len(ph) / N
        busy = sum(e - s for s, e, _ in ph)
        span = ph[-1][1] - ph[0][0]
        # gaps between consecutive kernels, excluding the host-side gap between forwards (the largest N-1 gaps)
        gaps = sorted((b[0] - a[1] for a, b in zip(ph, ph[1:])), reverse=True)
        between, inside = gaps[:N - 1], gaps[N - 1:]
        by = {}
        for s, e, k in ph:
            by.setdefault(k, []).append(e - s)
        stats[name] = by
        print(f"{name:13s}: {len(ph)} kernels = {per:.1f} per forward; kernel time {busy / N / 1e3:8.1f} us per forward; gaps inside a forward "
              f"{sum(max(0, g) for g in inside) / N / 1e3:8.1f} us per forward (overlapped launches count as 0; {sum(1 for g in inside if g < 0)} overlaps); "
              f"gap between forwards {sum(between) / max(1, len(between)) / 1e3:8.1f} us; whole phase {span / N / 1e3:8.1f} us per forward")
    print(f"\n{'kernel':62s} {'eager n':>8s} {'us':>8s} {'replay n':>8s} {'us':>8s} {'ratio':>6s}")
    for k in sorted(set(stats["eager"]) | set(stats["graph replay"]), key=lambda k: -sum(stats["graph replay"].get(k, [0]))):
        a, b = stats["eager"].get(k, []), stats["graph replay"].get(k, [])
        ma, mb = (sum(a) / len(a) / 1e3 if a else 0.0), (sum(b) / len(b) / 1e3 if b else 0.0)
        print(f"{k:62s} {len(a) / N:8.1f} {ma:8.2f} {len(b) / N:8.1f} {mb:8.2f} {(mb / ma if ma else 0):6.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2])
    else:
        run()

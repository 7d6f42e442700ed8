#!/usr/bin/env python3
"""How do kernels of K forwards in flight actually overlap on the GPU?

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/inflight_trace.py run [K]
    python3 tools/inflight_trace.py analyse OUT

run: 60 reference pairs (2-6 s clips) one at a time, a 0.5 s pause, the same 60 with K in flight.  analyse: per phase the wall span, the
union of kernel intervals (time with at least one kernel resident), the sum of kernel durations, their ratio (average number of
kernels resident while any is) and the same weighted by grid size in workgroups (how much of 512 workgroup slots is asked for)."""
import csv, glob, importlib, os, sys, time


def run(k):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    la = importlib.import_module("loco-asr_amd")
    sd = la.synth.encoder_state_dict(0)
    pre, enc_sd = la.synth.split_state_dict(sd)
    m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({a: torch.from_numpy(v) for a, v in pre.items()}, {a: torch.from_numpy(v) for a, v in enc_sd.items()}).cuda()
    enc = m.speecht5.encoder
    NP = 60
    lens = la.synth.mixed_lengths(2 * NP, 6 * 16000, min_fraction=2.0 / 6.0)
    fe = la.SpeechT5FeatureExtractorMI355X()
    batches = []
    for p in range(NP):
        b = fe(audio=[la.synth.clip(2 * p + j, lens[2 * p + j]) for j in (0, 1)], sampling_rate=16000, return_tensors="pt")
        batches.append((b["input_values"].cuda(), b["attention_mask"].cuda()))
    enc.set_inflight(k)
    for x, a in batches[:2 * k]:
        enc.forward_async(input_values=x, attention_mask=a)
    enc.drain()
    for x, a in batches[:4]:
        enc(input_values=x, attention_mask=a)
    torch.cuda.synchronize(); time.sleep(0.5)
    t0 = time.perf_counter()
    for x, a in batches:
        enc(input_values=x, attention_mask=a)
    torch.cuda.synchronize()
    print(f"one at a time: {(time.perf_counter() - t0) / NP * 1e3:.3f} ms per pair", flush=True)
    time.sleep(0.5)
    t0 = time.perf_counter()
    tk = [enc.forward_async(input_values=x, attention_mask=a) for x, a in batches]
    [t.result() for t in tk]
    torch.cuda.synchronize()
    print(f"{k} in flight: {(time.perf_counter() - t0) / NP * 1e3:.3f} ms per pair", flush=True)
    time.sleep(0.5)


def analyse(folder):
    rows = []
    for f in glob.glob(os.path.join(folder, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            wg = 1
            try:
                wg = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1))))
            except (KeyError, ValueError):
                pass
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), wg, r["Kernel_Name"][:50]))
    rows.sort()
    phases, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - max(x[1] for x in cur[-32:]) > 300_000_000:
            phases.append(cur); cur = []
        cur.append(b)
    phases.append(cur)
    print(f"{len(rows)} kernels, phases: {[len(p) for p in phases]}")
    for name, ph in zip(("one at a time", "in flight"), phases[-2:]):
        span = max(e for _, e, _, _ in ph) - ph[0][0]
        ev = sorted([(s, 1) for s, _, _, _ in ph] + [(e, -1) for _, e, _, _ in ph])
        busy, depth, last = 0, 0, ev[0][0]
        hist = {}
        for t, d in ev:
            if depth > 0:
                busy += t - last
                hist[depth] = hist.get(depth, 0) + (t - last)
            depth += d; last = t
        total = sum(e - s for s, e, _, _ in ph)
        wgt = sum((e - s) * min(w, 512) for s, e, w, _ in ph)
        print(f"{name:14s}: {len(ph)} kernels, span {span / 1e6:8.2f} ms, some kernel resident {busy / 1e6:8.2f} ms ({busy / span:.2f} of the span), "
              f"sum of durations {total / 1e6:8.2f} ms -> {total / busy:.2f} kernels resident on average; workgroup-slot demand {wgt / busy / 512:.2f} of 512 slots")
        print("                 time share by number of kernels resident: " + ", ".join(f"{d}: {hist[d] / busy:.2f}" for d in sorted(hist)[:10]))
        by = {}
        for s, e, w, k in ph:
            d = by.setdefault(k, [0, 0]); d[0] += 1; d[1] += e - s
        top = sorted(by.items(), key=lambda kv: -kv[1][1])[:6]
        print("                 " + "; ".join(f"{k[:34]} n={v[0]} avg {v[1] / v[0] / 1e3:.1f} us" for k, v in top))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    else:
        analyse(sys.argv[2])

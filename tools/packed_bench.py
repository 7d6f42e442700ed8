#!/usr/bin/env python3
"""GPU-side capability of the PACKED forward at the reference's operating point (batch_size = 2 in corpus order, …base…py:67-68),
host pipeline taken out: N reference pairs of a SLURP-like ragged corpus (2-6 s), packed G pairs at a time (encoder.forward_packed:
one launch sequence per pack, every clip keeps its own pair's padded length), packs pre-staged in HBM, K packs in flight.

    python3 tools/packed_bench.py [pairs, default 512] [--sorted] [--reps R]

Frames are the frames the reference pickles: B_i x T_i of every pair (its own padded frames included, the pack's padding not).
--sorted: pairs are sorted by length inside a window of 8 packs before packing (what extract.py --pack does)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
NP = int(args[0]) if args else 512
SORTED = "--sorted" in sys.argv
REPS = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
STREAMS = int(sys.argv[sys.argv.index("--streams") + 1]) if "--streams" in sys.argv else 2  # the library's two half-batch schedule inside one forward
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                     {k: torch.from_numpy(v) for k, v in enc_sd.items()}).cuda()
enc = m.speecht5.encoder
enc.streams = STREAMS
lens = la.synth.mixed_lengths(2 * NP, 6 * 16000, min_fraction=2.0 / 6.0)
fe = la.SpeechT5FeatureExtractorMI355X()
batches, frames = [], 0
for p in range(NP):
    b = fe(audio=[la.synth.clip(2 * p + j, lens[2 * p + j]) for j in (0, 1)], sampling_rate=16000, return_tensors="pt")
    batches.append(dict(input_values=b["input_values"], attention_mask=b["attention_mask"]))
    frames += 2 * la.synth.conv_out_length(b["input_values"].shape[1])
print(f"{NP} reference pairs, {frames} padded frames, mean clip {sum(lens) / len(lens) / 16000:.2f} s, sorted inside windows: {SORTED}, streams inside a forward: {STREAMS}", flush=True)
for G in (4, 8, 16, 32, 64, 128):
    if G > NP:
        break
    order = list(range(NP))
    if SORTED:
        win = 8 * G
        order = [i for w0 in range(0, NP, win) for i in sorted(range(w0, min(NP, w0 + win)), key=lambda j: batches[j]["input_values"].shape[1])]
    packs = [enc.pack_batches([batches[i] for i in order[g0:g0 + G]]) for g0 in range(0, NP, G)]
    torch.cuda.synchronize()
    rows = sum(p.wav.shape[0] * la.synth.conv_out_length(p.wav.shape[1]) for p in packs)
    for K in (1, 2, 3):
        enc.set_inflight(K)
        for p in packs[:2 * K]:
            enc.forward_packed_async(packed=p)
        enc.drain(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(REPS):
            t0 = time.perf_counter()
            tickets = [enc.forward_packed_async(packed=p) for p in packs]
            for t in tickets:
                t.result()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            assert not any(t.used_fp32 for t in tickets)
            del tickets
        print(f"pack of {G:3d} pairs, {K} in flight: {best / len(packs) * 1e3:8.3f} ms per pack, {frames / best:10,.0f} frames/s "
              f"(rows computed / frames kept = {rows / frames:.3f})", flush=True)

#!/usr/bin/env python3
"""Per-kernel time of ONE packed forward at the reference's operating point (32 pairs of 2-6 s utterances, sorted): where a pack's
16 ms go.  The library's own per-launch events (single stream, every launch bracketed)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).cuda()
enc = m.speecht5.encoder
npairs = 8 * G
lens = la.synth.mixed_lengths(2 * npairs, 6 * 16000, min_fraction=2.0 / 6.0)
fe = la.SpeechT5FeatureExtractorMI355X()
order = sorted(range(npairs), key=lambda p_: max(lens[2 * p_], lens[2 * p_ + 1]))
packs = []
for g0 in range(0, npairs, G):
    groups = [[la.synth.clip(2 * p_ + j, lens[2 * p_ + j]) for j in (0, 1)] for p_ in order[g0:g0 + G]]
    packs.append(fe.pack_clips(groups, torch.device("cuda"), la.synth.conv_out_length))
for pk in packs[:2]:
    enc.forward_packed(packed=pk)
enc.set_profiling(True)
enc.profile_reset()
kept = 0
for pk in packs:
    enc.forward_packed(packed=pk)
    kept += sum(nb * t for (_, nb, t) in pk.spans)
torch.cuda.synchronize()
st = enc.profile_read()
enc.set_profiling(False)
tot = sum(s["ms"] for s in st)
print(f"{len(packs)} packs of {G} pairs (T = {[la.synth.conv_out_length(p.wav.shape[1]) for p in packs]}), {kept} frames kept; kernel time {tot / len(packs):.3f} ms per pack")
for s in sorted(st, key=lambda s_: -s_["ms"]):
    print(f"  {s['name']:22s} {s['launches'] / len(packs):6.1f} launches  {s['ms'] / len(packs):7.3f} ms  {100 * s['ms'] / tot:5.1f} %  "
          f"{(s['flops'] / (s['ms'] * 1e-3) / 1e12) if s['flops'] else 0:7.1f} TFLOP/s  {(s['bytes'] / (s['ms'] * 1e-3) / 1e9) if s['bytes'] else 0:7.1f} GB/s")

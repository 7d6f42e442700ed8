#!/usr/bin/env python3
"""One shape, one stream mode, a few forwards: run under `rocprofv3 --kernel-trace --stats` to compare the per-kernel times of
loco_set_streams 1 and 2.   python tools/stream_mode_trace.py B seconds streams [reps]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
B, secs, n = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = m.speecht5.encoder
enc.streams = n
x, msk = la.synth.batch([int(secs * 16000)] * B)
xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
for _ in range(3):
    enc(input_values=xs, attention_mask=ms)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    enc(input_values=xs, attention_mask=ms)
torch.cuda.synchronize()
print(f"batch {B} x {secs:.0f} s, streams={n}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms/step", flush=True)

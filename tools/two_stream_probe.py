#!/usr/bin/env python3
"""A/B in one process (same device, same clocks): loco_set_streams 1 vs 2 through the product path."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = m.speecht5.encoder
SHAPES = ((32, 30.0, 8), (4, 600.0, 3), (16, 30.0, 8)) if len(sys.argv) < 2 else ((2, 5.0, 40), (4, 5.0, 40), (8, 5.0, 30), (8, 2.5, 30), (6, 10.0, 30))
for B, secs, reps in SHAPES:
    x, msk = la.synth.batch([int(secs * 16000)] * B)
    xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
    for rnd in range(2):
        for n in (1, 2):
            enc.streams = n
            for _ in range(2):
                y = enc(input_values=xs, attention_mask=ms).last_hidden_state
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                y = enc(input_values=xs, attention_mask=ms).last_hidden_state
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e3
            print(f"batch {B} x {secs:.0f} s, streams={n}: {dt:.2f} ms/step  {B * y.shape[1] / dt * 1e3:.0f} frames/s", flush=True)
    del xs, ms, y
    enc._workspace = None
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Per-kernel time for the reference's own batch shape (batch_size = 2, ~5 s utterances; ...base...py:67)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
if len(sys.argv) > 1:  # profile another build of the library (A/B runs inside one gpurun call)
    la._lib.LIB_PATH = os.path.abspath(sys.argv[1])
    print("library:", la._lib.LIB_PATH)
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = m.speecht5.encoder
for lens in ([80000], [80000, 48000], [80000] * 8, [80000] * 32):
    x, msk = la.synth.batch(lens)
    xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
    for _ in range(3): enc(input_values=xs, attention_mask=ms)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): enc(input_values=xs, attention_mask=ms)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    enc.set_profiling(True); enc.profile_reset()
    for _ in range(5): enc(input_values=xs, attention_mask=ms)
    st = enc.profile_read(); enc.set_profiling(False)
    print(f"batch {len(lens)} x 5 s: {dt:.3f} ms per forward;", ", ".join(f"{s['name']} {s['ms']/5:.3f} ms/{s['launches']//5}" for s in st))

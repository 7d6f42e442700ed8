#!/usr/bin/env python3
"""Mid-size batches through the product path under two builds of the library (one child process each, interleaved A B A B): does the
GEMM's tile choice (round 4: one cost model over all four forms) hold up between the reference's pair and the 30 s x 32 headline?

    python3 tools/midsize_ab.py loco-asr_amd/libloco_asr.so tools/ab/libold_r4_tiles.so"""
import importlib, os, subprocess, sys, time
SHAPES = ((4, 30.0, 10), (8, 30.0, 8), (16, 30.0, 6), (8, 10.0, 12), (16, 10.0, 10), (32, 5.0, 12), (64, 4.0, 10), (12, 60.0, 4))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    la = importlib.import_module("loco-asr_amd")
    sd = la.synth.encoder_state_dict(0)
    pre, enc_sd = la.synth.split_state_dict(sd)
    m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
    enc = m.speecht5.encoder
    for B, secs, reps in SHAPES:
        x, msk = la.synth.batch([int(secs * 16000)] * B)
        xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
        out = []
        for n in (1, 2):
            enc.streams = n
            for _ in range(2):
                y = enc(input_values=xs, attention_mask=ms).last_hidden_state
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                y = enc(input_values=xs, attention_mask=ms).last_hidden_state
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e3
            out.append(f"streams={n}: {dt:7.2f} ms {B * y.shape[1] / dt * 1e3:9.0f} frames/s")
        print(f"  {B:2d} x {secs:4.0f} s (M = {B * y.shape[1]:6d}): " + "   ".join(out), flush=True)
        del xs, ms, y
        enc._workspace = None
        torch.cuda.empty_cache()
    sys.exit(0)
libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, LOCO_ASR_LIB=os.path.abspath(lib), LOCO_ALLOW_BANNED_ISA="1")
        print(f"== {lib}", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print(r.stdout + (r.stderr[-600:] if r.returncode else ""), flush=True)

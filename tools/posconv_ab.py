#!/usr/bin/env python3
"""The grouped positional conv on the resident-A kernel against the generic GEMM kernel (LOCO_POSCONV_GENERIC=1), whole forwards of
30 s x 32 and 10 min x 4, interleaved A B B A in one process; the per-kernel bucket times come from the library's own events."""
import importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).cuda()
enc = m.speecht5.encoder
lib = enc._lib if hasattr(enc, "_lib") else la._lib.load()
for B, L, reps in ((32, 480000, 6), (4, 9600000, 2)):
    x, _ = la.synth.batch([L] * B)
    x = torch.from_numpy(x).cuda()
    enc(input_values=x)
    enc.set_profiling(True)
    res = {"0": [], "1": []}
    outs = {}
    for rnd in range(4):
        for f in (("0", "1", "1", "0") if rnd % 2 == 0 else ("1", "0", "0", "1")):
            if f == "1":
                os.environ["LOCO_POSCONV_GENERIC"] = "1"
            else:
                os.environ.pop("LOCO_POSCONV_GENERIC", None)
            lib.loco_debug_reload_gemm_knobs()
            enc.profile_reset()
            for _ in range(reps):
                y = enc(input_values=x).last_hidden_state
            torch.cuda.synchronize()
            st = {s["name"]: s for s in enc.profile_read()}
            res[f].append(st["pos_conv_f16x3_gemm"]["ms"] / reps)
            outs[f] = y.clone()
    enc.set_profiling(False)
    os.environ.pop("LOCO_POSCONV_GENERIC", None)
    lib.loco_debug_reload_gemm_knobs()
    a, b = statistics.median(res["0"]), statistics.median(res["1"])
    print(f"B={B} x {L / 16000:.0f} s: positional conv resident-A {a:.3f} ms, generic GEMM kernel {b:.3f} ms ({100 * (a / b - 1):+.1f} %); outputs bit-identical: {torch.equal(outs['0'], outs['1'])}", flush=True)
    del x

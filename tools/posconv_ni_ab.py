#!/usr/bin/env python3
"""pos_conv_resident_kernel with 512-frame tiles (NI = 4) against 256-frame tiles (NI = 2) at several clip lengths: whole forwards,
interleaved A B B A in one process (LOCO_POSCONV_NI + loco_debug_reload_gemm_knobs), the library's own per-bucket events; outputs compared bit for bit."""
import importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
sd = la.synth.encoder_state_dict(0, layers=1)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}, layers=1).cuda()
enc = m.speecht5.encoder
lib = enc._lib
for B, secs, reps in ((64, 3.4, 10), (64, 4.0, 10), (64, 5.0, 10), (64, 5.2, 10), (64, 6.0, 10), (32, 12.0, 8), (32, 15.5, 8), (32, 30.0, 6)):
    x, _ = la.synth.batch([int(secs * 16000)] * B)
    x = torch.from_numpy(x).cuda()
    enc(input_values=x)
    enc.set_profiling(True)
    res, outs = {"2": [], "4": []}, {}
    for rnd in range(4):
        for f in (("4", "2", "2", "4") if rnd % 2 == 0 else ("2", "4", "4", "2")):
            os.environ["LOCO_POSCONV_NI"] = f
            lib.loco_debug_reload_gemm_knobs()
            enc.profile_reset()
            for _ in range(reps):
                y = enc(input_values=x).last_hidden_state
            torch.cuda.synchronize()
            st = {s["name"]: s for s in enc.profile_read()}
            res[f].append(st["pos_conv_f16x3_gemm"]["ms"] / reps)
            outs[f] = y.clone()
    enc.set_profiling(False)
    os.environ.pop("LOCO_POSCONV_NI", None)
    lib.loco_debug_reload_gemm_knobs()
    a, b = statistics.median(res["4"]), statistics.median(res["2"])
    print(f"B={B} T={y.shape[1]:5d}: 512-frame tiles {a:.3f} ms, 256-frame tiles {b:.3f} ms (ratio per tile {b * ((y.shape[1] + 511) // 512) / (a * ((y.shape[1] + 255) // 256)):.2f}; "
          f"{100 * (b / a - 1):+.1f} %); bit-identical: {torch.equal(outs['2'], outs['4'])}", flush=True)
    del x

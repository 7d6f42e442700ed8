#!/usr/bin/env python3
"""Do two builds of the library produce the same BITS?  Runs the encoder of each given .so in a child process on the same inputs (a ragged
reference pair -- the split-K path -- and 4 x 30 s -- the big-tile path) and prints a sha256 per hidden state, so that the first stage
that differs is named.

    python3 tools/bit_compare.py loco-asr_amd/libloco_asr.so tools/ab/libold_<commit>.so ..."""
import hashlib, importlib, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    la = importlib.import_module("loco-asr_amd")
    sd = la.synth.encoder_state_dict(0)
    pre, enc_sd = la.synth.split_state_dict(sd)
    m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                         {k: torch.from_numpy(v) for k, v in enc_sd.items()}).cuda()
    enc = m.speecht5.encoder
    enc.precision = os.environ.get("LOCO_BITCMP_PRECISION", enc.precision)  # --precision f32: the exact-fp32 kernel set
    for name, lens in (("pair 5 s + 3.7 s", [80000, 59200]), ("4 x 30 s", [480000] * 4)):
        x, msk = la.synth.batch(lens, first_index=7)
        st = {}
        out = enc(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda(), output_hidden_states=True, stage_taps=st)
        torch.cuda.synchronize()
        sh = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]
        print(f"{name}: conv {sh(st['conv_stack'])} proj {sh(st['feature_projection'])} prenet {sh(st['prenet'])} | "
              + " ".join(sh(h) for h in out.hidden_states), flush=True)
    sys.exit(0)

for lib in [a for a in sys.argv[1:] if a.endswith('.so')]:
    env = dict(os.environ, LOCO_ASR_LIB=os.path.abspath(lib), LOCO_ALLOW_BANNED_ISA="1")
    if "--precision" in sys.argv:
        env["LOCO_BITCMP_PRECISION"] = sys.argv[sys.argv.index("--precision") + 1]
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    print(f"== {lib}\n{r.stdout}{r.stderr[-400:] if r.returncode else ''}", flush=True)

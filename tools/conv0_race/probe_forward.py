#!/usr/bin/env python3
"""Round 1's failing scenario, replayed with per-layer outputs: two half-batch forwards of ONE handle on two user streams, three
steps back to back, against a single in-order pass -- with the library named by LOCO_ASR_LIB (libloco_oldconv0.so = today's
library with only the packed-tap conv0 kernel of commit 491647a^ swapped back in; unset = the shipped library).  Every
forward writes its 13 hidden states (a per-call argument), so a mismatch is located: which clip, and which is the FIRST
hidden state that differs -- index 0 is the encoder's input, i.e. the prenet (conv stack, projection, positional conv).

    LOCO_ALLOW_BANNED_ISA=1 LOCO_ASR_LIB=tools/conv0_race/libloco_oldconv0.so python tools/conv0_race/probe_forward.py [trials]
(_lib.load() refuses a foreign build that carries the banned encoding unless LOCO_ALLOW_BANNED_ISA=1 says this is the reproducer)
"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
la = importlib.import_module("loco-asr_amd")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
trials = int(args[0]) if args else 12
HIDDEN = "--hidden" in sys.argv  # also collect the 13 hidden states of every forward (adds 13 copy kernels per forward: other timing)
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = m.speecht5.encoder
enc.range_policy = "off"
lib = enc._lib
B, L = 32, 480000
x, msk = la.synth.batch([L] * B)
xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda().int()
enc.streams = 1
ref = enc(input_values=xs, attention_mask=ms, output_hidden_states=True)
ref_h = [h.clone() for h in ref.hidden_states]
torch.cuda.synchronize()
lib.loco_set_streams(enc._handle, 1)
h = enc._handle
T = int(lib.loco_output_frames(L))
half = B // 2
need = int(lib.loco_workspace_bytes(h, half, L))
wss = [torch.empty(need, dtype=torch.uint8, device="cuda") for _ in range(2)]
out = torch.empty(B, T, 768, device="cuda")
hid = [[torch.empty(half, T, 768, device="cuda") for _ in range(13)] for _ in range(2)]
hptrs = [(C.c_void_p * 13)(*[t.data_ptr() for t in hid[i]]) for i in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
fails = 0
first_stage = {}
for trial in range(trials):
    out.zero_()
    torch.cuda.synchronize()
    for _ in range(3):
        for i in range(2):
            a, b = i * half, (i + 1) * half
            rc = lib.loco_forward(h, C.c_void_p(xs[a:b].data_ptr()), C.c_void_p(ms[a:b].data_ptr()), half, L, C.c_void_p(out[a:b].data_ptr()), None,
                                  hptrs[i] if HIDDEN else None, C.c_void_p(wss[i].data_ptr()), need, C.c_void_p(streams[i].cuda_stream))
            assert rc == 0, lib.loco_last_error()
    torch.cuda.synchronize()
    bad = []
    if not HIDDEN:
        bad = [(g, 12, float((out[g] - ref_h[12][g]).abs().max())) for g in range(B) if not torch.equal(out[g], ref_h[12][g])]
    for i in range(2 if HIDDEN else 0):
        for c in range(half):
            g = i * half + c
            for k in range(13):
                if not torch.equal(hid[i][k][c], ref_h[k][g]):
                    d = (hid[i][k][c] - ref_h[k][g]).abs()
                    rows = torch.nonzero(d.amax(1) > 0).flatten()
                    bad.append((g, k, float(d.max()), int(rows.numel()), int(rows.min()), int(rows.max())))
                    first_stage[k] = first_stage.get(k, 0) + 1
                    break
    if bad:
        fails += 1
        print(f"trial {trial}: {len(bad)} clips differ; (clip, first differing hidden state, max |diff|, frames touched, first, last): {bad[:6]}", flush=True)
print(f"library {os.environ.get('LOCO_ASR_LIB', '(shipped)')}, hidden states {'on' if HIDDEN else 'off'}: {fails} of {trials} trials differ from the single pass; first differing hidden state -> clips: {first_stage}", flush=True)

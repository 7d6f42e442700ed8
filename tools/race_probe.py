#!/usr/bin/env python3
"""Race detector: two half-batch forwards on two streams, several steps back to back, compared bit for bit with one pass."""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = m.speecht5.encoder
lib = enc._lib
B, secs = (32, 30.0) if len(sys.argv) < 3 else (int(sys.argv[1]), float(sys.argv[2]))
x, msk = la.synth.batch([int(secs * 16000)] * B)
xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda().int()
enc.streams = 1
ref = enc(input_values=xs, attention_mask=ms).last_hidden_state
if hasattr(lib, "loco_set_streams"):
    lib.loco_set_streams(enc._handle, 1)
L = xs.shape[1]; T = int(lib.loco_output_frames(L)); h = enc._handle
cuts = [0, B // 2, B]
wss = [torch.empty(int(lib.loco_workspace_bytes(h, B // 2, L)) + (1 << 20), dtype=torch.uint8, device="cuda") for i in range(2)]
guard = torch.zeros(1 << 22, dtype=torch.uint8, device="cuda")
out = torch.empty(B, T, 768, device="cuda")
streams = [torch.cuda.Stream() for _ in range(2)]
fails = 0
for trial in range(40):
    out.zero_()
    torch.cuda.synchronize()
    for _ in range(3):
        for i in range(2):
            a, b = cuts[i], cuts[i + 1]
            rc = lib.loco_forward(h, C.c_void_p(xs[a:b].data_ptr()), C.c_void_p(ms[a:b].data_ptr()), b - a, L, C.c_void_p(out[a:b].data_ptr()), None, None,
                                  C.c_void_p(wss[i].data_ptr()), wss[i].numel() - (1 << 20), C.c_void_p(streams[i].cuda_stream))
            assert rc == 0
    torch.cuda.synchronize()
    bad = [(i, float((out[i] - ref[i]).abs().max())) for i in range(B) if not torch.equal(out[i], ref[i])]
    if bad:
        fails += 1
        print(f"trial {trial}: MISMATCH clips {bad[:6]}", flush=True)
print(f"{fails} of 40 trials differ from the single pass; guard intact: {int(guard.sum()) == 0}; ws tails intact: {[int(w[-(1<<20):].sum()) for w in wss]}", flush=True)

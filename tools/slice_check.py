#!/usr/bin/env python3
"""Is a batch slice's result independent of (a) how the batch is sliced and (b) other slices running concurrently?"""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = m.speecht5.encoder
lib = enc._lib
B, secs = 32, 30.0
x, msk = la.synth.batch([int(secs * 16000)] * B)
xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda().int()
enc.streams = 1
ref = enc(input_values=xs, attention_mask=ms, output_hidden_states=False, stage_taps=None).last_hidden_state
taps = {}
enc(input_values=xs, attention_mask=ms, stage_taps=taps)
lib.loco_set_streams(enc._handle, 1)
L = xs.shape[1]; T = int(lib.loco_output_frames(L)); h = enc._handle

def run(cuts, concurrent):
    n = len(cuts) - 1
    wss = [torch.empty(int(lib.loco_workspace_bytes(h, cuts[i + 1] - cuts[i], L)), dtype=torch.uint8, device="cuda") for i in range(n)]
    streams = [torch.cuda.Stream() if concurrent else torch.cuda.current_stream() for _ in range(n)]
    out = torch.empty(B, T, 768, device="cuda")
    torch.cuda.synchronize()
    for i in range(n):
        a, b = cuts[i], cuts[i + 1]
        rc = lib.loco_forward(h, C.c_void_p(xs[a:b].data_ptr()), C.c_void_p(ms[a:b].data_ptr()), b - a, L, C.c_void_p(out[a:b].data_ptr()), None, None,
                              C.c_void_p(wss[i].data_ptr()), wss[i].numel(), C.c_void_p(streams[i].cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    return out

for cuts in ([0, 16, 32], [0, 8, 16, 24, 32], [0, 12, 24, 32], [0, 24, 32]):
    seq = run(cuts, False)
    con = run(cuts, True)
    con2 = run(cuts, True)
    bad = [(i, float((seq[i] - ref[i]).abs().max())) for i in range(B) if not torch.equal(seq[i], ref[i])]
    print(f"cuts {cuts}: sequential slices == full batch: {torch.equal(seq, ref)}; concurrent == sequential: {torch.equal(con, seq)}; concurrent repeat equal: {torch.equal(con, con2)}; clips differing from full: {bad[:6]}", flush=True)

#!/usr/bin/env python3
"""How many independent slices of the batch are worth running side by side?  Calls loco_forward directly on n slices of the
batch on n torch streams (library-internal splitting off), same process."""
import ctypes as C, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = m.speecht5.encoder
lib = enc._lib
for B, secs, reps in ((32, 30.0, 6), (32, 10.0, 10), (64, 5.0, 10), (48, 30.0, 4)):
    x, msk = la.synth.batch([int(secs * 16000)] * B)
    xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda().int()
    enc.streams = 1
    ref = enc(input_values=xs, attention_mask=ms).last_hidden_state
    lib.loco_set_streams(enc._handle, 1)
    L = xs.shape[1]; T = int(lib.loco_output_frames(L)); h = enc._handle
    for n in (1, 2, 3, 4):
        cuts = [round(i * B / n) for i in range(n + 1)]
        sizes = [cuts[i + 1] - cuts[i] for i in range(n)]
        wss = [torch.empty(int(lib.loco_workspace_bytes(h, b, L)), dtype=torch.uint8, device="cuda") for b in sizes]
        streams = [torch.cuda.Stream() for _ in range(n)]
        out = torch.empty(B, T, 768, device="cuda"); frames = torch.empty(B, dtype=torch.int32, device="cuda")
        def step():
            for i in range(n):
                a, b = cuts[i], cuts[i + 1]
                rc = lib.loco_forward(h, C.c_void_p(xs[a:b].data_ptr()), C.c_void_p(ms[a:b].data_ptr()), b - a, L, C.c_void_p(out[a:b].data_ptr()),
                                      C.c_void_p(frames[a:b].data_ptr()), None, C.c_void_p(wss[i].data_ptr()), wss[i].numel(), C.c_void_p(streams[i].cuda_stream))
                assert rc == 0, lib.loco_last_error()
        step(); step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e3
        print(f"batch {B} x {secs:.0f} s in {n} slice(s) {sizes}: {dt:.2f} ms/step {B*T/dt*1e3:.0f} frames/s  equal={bool(torch.equal(out, ref))}", flush=True)
    del xs, ms, ref, out, wss
    enc._workspace = None
    torch.cuda.empty_cache()

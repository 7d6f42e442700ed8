#!/usr/bin/env python3
"""Measure the two "next" rows in front of the path: the text branch (tokens/s) and the device-side
waveform normaliser (GB/s against its 2 x 4 bytes/sample of compulsory traffic + the mask)."""
import ctypes as C, importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
lib = la._lib.load()

sd = la.synth.encoder_state_dict(0)
_, enc_sd = la.synth.split_state_dict(sd)
tpre = {k[len("text_prenet."):]: torch.from_numpy(np.asarray(v)) for k, v in la.synth.text_prenet_state_dict(0).items()}
m = la.SpeechT5ForTextToSpeechMI355X.from_state_dicts(tpre, {k: torch.from_numpy(v) for k, v in enc_sd.items()}).to("cuda")
enc = m.speecht5.encoder
for B, T in ((2, 40), (16, 40), (256, 40), (1024, 40), (512, 128), (128, 450)):
    ids = torch.from_numpy(la.synth.token_ids(B, T, seed=1)[0]).cuda()
    for _ in range(3):
        enc(ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        enc(ids)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    fl = B * (T * 284.2e6 * (12 * 14.155e6 + 12 * 12 * 320 * 64 * 2) / 284.2e6 + 36864.0 * T * T)  # encoder layers + rel-pos table + attention
    print(f"text encoder  batch {B:5d} x {T:3d} tokens: {dt*1e3:8.3f} ms  {B*T/dt:12.0f} tokens/s  {fl/dt/1e12:6.1f} TFLOP/s algorithmic", flush=True)

for B, L in ((32, 480000), (4, 9600000)):
    x = torch.randn(B, L, device="cuda") * 0.1 + 0.05
    msk = torch.ones(B, L, dtype=torch.int32, device="cuda")
    msk[1, L // 2:] = 0
    out = torch.empty_like(x)
    scratch = torch.empty(int(lib.loco_normalize_scratch_bytes(B)), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.loco_op_normalize_waveform(C.c_void_p(x.data_ptr()), C.c_void_p(msk.data_ptr()), B, L, 0.0, C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(scratch.data_ptr()), scratch.numel(), st)
    for _ in range(3):
        assert call() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = B * L * (4 + 4 + 4 + 4)  # mask read, waveform read twice (moments, apply), output write
    print(f"normaliser    batch {B:3d} x {L:8d} samples: {ms:7.3f} ms  {by/ms/1e6:7.0f} GB/s of {by/1e6:.0f} MB (mask + 2 reads + 1 write)", flush=True)

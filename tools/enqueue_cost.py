#!/usr/bin/env python3
"""Host cost of ONE enqueue of the forward (a reference pair of ~4 s clips): the C call alone, and the Python wrapper around it."""
import ctypes as C, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
sd = la.synth.encoder_state_dict(0)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                     {k: torch.from_numpy(v) for k, v in enc_sd.items()}).cuda()
enc = m.speecht5.encoder
x, a = la.synth.batch([64000, 52000])
x, a = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()
enc(input_values=x, attention_mask=a)
lib = enc._lib
B, L = x.shape
T = int(lib.loco_output_frames(L))
NS = 32
nst = int(lib.loco_status_bytes())
slots = [dict(st=torch.cuda.Stream(), ws=torch.empty(int(lib.loco_workspace_bytes(enc._handle, B, L)), dtype=torch.uint8, device="cuda"),
              status=torch.zeros(nst, dtype=torch.uint8).pin_memory(), out=torch.empty((B, T, 768), device="cuda")) for _ in range(NS)]
torch.cuda.synchronize()
for rep in range(3):
    t_call = []
    t0 = time.perf_counter()
    for s in slots:
        t1 = time.perf_counter()
        rc = lib.loco_forward_async(enc._handle, 1, C.c_void_p(x.data_ptr()), C.c_void_p(a.data_ptr()), B, L, C.c_void_p(s["out"].data_ptr()), None, None,
                                    C.c_void_p(s["ws"].data_ptr()), s["ws"].numel(), C.c_void_p(s["st"].cuda_stream), C.c_void_p(s["status"].data_ptr()))
        t_call.append(time.perf_counter() - t1)
        assert rc == 0
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    t_call.sort()
    print(f"{NS} forwards on {NS} streams: C call median {t_call[NS // 2] * 1e3:.3f} ms (min {t_call[0] * 1e3:.3f}, max {t_call[-1] * 1e3:.3f}); "
          f"all enqueued after {t_enq * 1e3:.2f} ms, all complete after {t_all * 1e3:.2f} ms = {t_all / NS * 1e3:.3f} ms per forward", flush=True)
enc.set_inflight(NS)
for rep in range(2):
    t0 = time.perf_counter()
    tk = [enc.forward_async(input_values=x, attention_mask=a) for _ in range(NS)]
    t_enq = time.perf_counter() - t0
    [t.result() for t in tk]
    t_all = time.perf_counter() - t0
    print(f"Python forward_async x {NS}: enqueued after {t_enq * 1e3:.2f} ms ({t_enq / NS * 1e3:.3f} ms each), resolved after {t_all * 1e3:.2f} ms = {t_all / NS * 1e3:.3f} ms per forward", flush=True)

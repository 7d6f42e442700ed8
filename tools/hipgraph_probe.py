import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
la = importlib.import_module("loco-asr_amd")
sys.path.insert(0, "/root/repo/tests")
m = la.SpeechT5ForSpeechToTextMI355X()
sd = la.synth.encoder_state_dict(0)
pre, encsd = la.synth.split_state_dict(sd)
m.speecht5.encoder.wrapped_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in encsd.items()})
m.speecht5.encoder.prenet.load_state_dict({k: torch.from_numpy(v) for k, v in pre.items()})
enc = m.to("cuda").speecht5.encoder
x, msk = la.synth.batch([80000])
xs, ms = torch.from_numpy(x).cuda(), torch.from_numpy(msk).cuda()
enc(input_values=xs, attention_mask=ms)
def timed(flag, n=30):
    enc.use_graphs = flag
    enc(input_values=xs, attention_mask=ms)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        enc(input_values=xs, attention_mask=ms)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    te = timed(False); fb_e = enc.last_range_fallback
    tg = timed(True); fb_g = enc.last_range_fallback
    print(f"eager {te:.3f} ms (fallback {fb_e}), graph {tg:.3f} ms (fallback {fb_g})", flush=True)
# GPU time of one replay by events
enc.use_graphs = True
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): enc(input_values=xs, attention_mask=ms)
e1.record(); torch.cuda.synchronize()
print("graph, events: %.3f ms per forward" % (e0.elapsed_time(e1) / 10))
enc.use_graphs = False
e0.record()
for _ in range(10): enc(input_values=xs, attention_mask=ms)
e1.record(); torch.cuda.synchronize()
print("eager, events: %.3f ms per forward" % (e0.elapsed_time(e1) / 10))

#!/usr/bin/env python3
"""Random PACKS against the one-batch forwards they replace (loco_forward_packed): 1-12 reference batches of 1-4 clips each, lengths
from one encoder frame (400 samples) to 20 s on and off every tile boundary, with masks, without, mixed; host-packed and device-packed;
both fp32-class modes.  Every clip's rows up to the end of ITS OWN batch (padded frames included) must agree with the forward of that
batch alone to 5e-6 relative L2 (the GEMM summation order is the only difference), valid-frame counts exactly.

    python3 tools/pack_fuzz.py [cases, default 40] [seed]"""
import importlib, os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LAYERS = 2
sd = la.synth.encoder_state_dict(0, LAYERS)
pre, enc_sd = la.synth.split_state_dict(sd)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}, layers=LAYERS).cuda()
enc = m.speecht5.encoder
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
LENS = [400, 401, 719, 720, 1039, 1040, 8000, 16000, 20479, 20480, 40000, 80000, 81234, 163840, 200001, 320000]
worst = 0.0
for case in range(N):
    nb = rng.randint(1, 12)
    mode = rng.choice(["mask", "nomask", "mixed"])
    on_device = rng.random() < 0.4
    prec = rng.choice(["f16x3", "f16x3", "f32"])
    enc.precision = prec
    batches = []
    for i in range(nb):
        longest = rng.choice(LENS)
        k = rng.choice([1, 2, 2, 2, 3, 4])
        lens = [longest] + [rng.randint(400, longest) for _ in range(k - 1)]
        rng.shuffle(lens)
        x, msk = la.synth.batch(lens, first_index=5000 + 97 * case + 7 * i)
        b = dict(input_values=torch.from_numpy(x))
        if mode == "mask" or (mode == "mixed" and rng.random() < 0.5):
            b["attention_mask"] = torch.from_numpy(msk)
        if on_device:
            b = {k_: v.cuda() for k_, v in b.items()}
        batches.append(b)
    ref, ref_frames = [], []
    for b in batches:
        ref.append(enc(**{k_: v.cuda() for k_, v in b.items()}).last_hidden_state.clone())
        ref_frames.append(enc.last_frames.cpu().tolist())
    assert not enc.last_range_fallback
    t = enc.forward_packed_async(batches)
    outs = t.result()
    assert not t.used_fp32
    fr = enc.last_frames.cpu().tolist()
    errs, b0 = [], 0
    for o, r, rf in zip(outs, ref, ref_frames):
        y = o.last_hidden_state
        assert tuple(y.shape) == tuple(r.shape), (y.shape, r.shape)
        assert fr[b0:b0 + len(rf)] == rf, (fr[b0:b0 + len(rf)], rf)
        errs += [rel(y[c], r[c]) for c in range(y.shape[0])]
        b0 += len(rf)
    assert bool(torch.isfinite(t.packed_output()[0]).all())  # the unspecified rows beyond a clip's own batch are finite
    worst = max(worst, max(errs))
    flag = "" if max(errs) < 5e-6 else "   <-- ABOVE 5e-6"
    print(f"case {case:2d}: {nb:2d} batches, {b0:2d} clips, {mode:6s} {'device' if on_device else 'host  '} {prec:5s} pack {tuple(t.packed_output()[0].shape)}: "
          f"worst clip {max(errs):.2e}{flag}", flush=True)
enc.precision = "f16x3"
print(f"worst over {N} cases: {worst:.2e}")
sys.exit(0 if worst < 5e-6 else 1)

#!/usr/bin/env python3
"""Where does the second weight family (g10: HF init distributions + x100 outlier channels) lose accuracy, and which stage trips the
f16x3 range guard?  Per encoder layer: the LOCAL error of each arithmetic mode -- the layer applied to the mode's own input hidden
state, against an fp64 evaluation of that layer on the same input (oracle.encoder_layer_rows) -- next to a plain torch fp32 layer."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import speecht5_oracle as oracle
la = importlib.import_module("loco-asr_amd")
sd = la.synth.encoder_state_dict_hf_init(0)
pre, enc_sd = la.synth.split_state_dict(sd)
x, msk = la.synth.batch([80000, 52000], first_index=40)
rows = [0, 1, 74, 148, 149, 161, 162, 248]
pe_k = torch.from_numpy(sd["wrapped_encoder.embed_positions.pe_k.weight"])
rel = lambda a, b: float((torch.as_tensor(a).double().cpu() - b.double()).norm() / b.double().norm())
for prec in ("f16x3", "f32"):
    m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                         {k: torch.from_numpy(v) for k, v in enc_sd.items()}, precision=prec).cuda()
    enc = m.speecht5.encoder
    enc.range_policy = "off"
    out = enc(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda(), output_hidden_states=True)
    torch.cuda.synchronize()
    if prec == "f16x3":
        buf = __import__("ctypes").create_string_buffer(400)
        rc = enc._lib.loco_forward_status(enc._handle, buf, 400)
        print("f16x3 range status:", rc, buf.value.decode())
        for n, l, a in enc.range_report():
            flag = " <-- outside [2^-6, 65504)" if not (2 ** -6 <= a < 65504) else ""
            print(f"   {n:55s} layer {l:3d} max|x| = {a:.4g}{flag}")
    hs = [h.cpu() for h in out.hidden_states]
    print(f"{prec}: finite = {bool(torch.isfinite(hs[-1]).all())}; local error of each layer vs fp64 on the same input (clip 0 rows), torch fp32 beside it")
    for l in range(12):
        ref = oracle.encoder_layer_rows(hs[l][0], rows, 249, sd, f"wrapped_encoder.layers.{l}.", pe_k)
        t32 = oracle.encoder_layer_rows(hs[l][0], rows, 249, sd, f"wrapped_encoder.layers.{l}.", pe_k, dtype=torch.float32)
        print(f"   layer {l:2d}: {prec} {rel(hs[l + 1][0, rows], ref):.2e}   torch fp32 {rel(t32, ref):.2e}   max|in| {float(hs[l].abs().max()):.1f}")

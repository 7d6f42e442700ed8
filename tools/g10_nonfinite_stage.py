import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
la = importlib.import_module("loco-asr_amd")
sd = la.synth.encoder_state_dict_hf_init(0, layers=2)
pre, enc_sd = la.synth.split_state_dict(sd)
x, msk = la.synth.batch([80000], first_index=40)
m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()}, {k: torch.from_numpy(v) for k, v in enc_sd.items()}, layers=2).cuda()
enc = m.speecht5.encoder
enc.range_policy = "off"
out = enc(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda())
torch.cuda.synchronize()
print("finite:", bool(torch.isfinite(out.last_hidden_state).all()))

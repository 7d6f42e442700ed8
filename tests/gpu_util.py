"""Helpers shared by the -m gpu tests: raw C-ABI calls on torch-owned device buffers."""
import ctypes as C
import importlib

import numpy as np
import torch

la = importlib.import_module("loco-asr_amd")
_libmod = importlib.import_module("loco-asr_amd._lib")


def lib():
    return _libmod.load()


def check(rc, what=""):
    return _libmod.check(rc, what)


def dev(a, dtype=torch.float32):
    t = torch.as_tensor(a)
    return t.to(device="cuda", dtype=dtype).contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel_l2(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm())


_model_cache = {}


def model(layers=12, seed=0, precision=None):
    """Encoder on cuda:0 loaded through the same two load_state_dict calls the reference makes.
    precision None = the package default ("f16x3"); the cached module's precision is reset on every call."""
    key = (layers, seed)
    if key not in _model_cache:
        sd = la.synth.encoder_state_dict(seed, layers)
        pre, enc = la.synth.split_state_dict(sd)
        m = la.SpeechT5ForSpeechToTextMI355X(layers)
        m.speecht5.encoder.wrapped_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in enc.items()})
        m.speecht5.encoder.prenet.load_state_dict({k: torch.from_numpy(v) for k, v in pre.items()})
        _model_cache[key] = (m.to("cuda"), sd)
    m, sd = _model_cache[key]
    m.speecht5.encoder.precision = precision or "f16x3"
    return m, sd

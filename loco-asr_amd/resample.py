"""Device sample-rate conversion to the model's 16 kHz ("next" row f-4): what ``librosa.load(path, sr=16000)`` does on the
host in the reference (/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:56), as one HBM-bound HIP kernel
(``loco_op_resample``, csrc/resample.hip).  Parity with librosa/soxr itself is unpinned -- see include/loco_asr.h.

    y = resample_to_16k(x, sr)          # x: 1-D numpy / torch (host or device), returns a CUDA float32 tensor

The tap table of a rate is designed once (fp64, in the library) and cached on the device.  No CPU path: without a GPU it raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

TARGET_SR = 16000
_tables = {}  # (sr_in, device index) -> (up, down, K, taps_dev)


def _table(sr_in: int, device: torch.device):
    key = (int(sr_in), device.index)
    if key not in _tables:
        lib = _lib.load()
        up, down, K = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(lib.loco_resample_design(int(sr_in), TARGET_SR, C.byref(up), C.byref(down), C.byref(K), None), "resample_design")
        taps = np.empty((up.value, K.value), np.float32)
        _lib.check(lib.loco_resample_design(int(sr_in), TARGET_SR, C.byref(up), C.byref(down), C.byref(K), taps.ctypes.data_as(C.c_void_p)),
                   "resample_design")
        _tables[key] = (up.value, down.value, K.value, torch.from_numpy(taps).to(device))
    return _tables[key]


def resampled_length(n_in: int, sr_in: int) -> int:
    from math import gcd
    g = gcd(int(sr_in), TARGET_SR)
    return int(_lib.load().loco_resample_length(int(n_in), TARGET_SR // g, int(sr_in) // g))


def resample_to_16k(x, sr_in: int, device=None) -> torch.Tensor:
    """x [n] or [B, n] (all clips of the batch the same length) at sr_in Hz -> float32 CUDA tensor at 16 kHz."""
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
    if device is None:
        device = t.device if t.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("resample_to_16k runs on an AMD GPU only (there is no CPU path)")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = t.to(device=device, dtype=torch.float32).contiguous()
    squeeze = t.dim() == 1
    if squeeze:
        t = t[None]
    if t.dim() != 2 or t.shape[1] < 1:
        raise ValueError(f"expected [samples] or [batch, samples], got {tuple(t.shape)}")
    if int(sr_in) == TARGET_SR:
        return t[0] if squeeze else t
    up, down, K, taps = _table(sr_in, device)
    B, n_in = t.shape
    lib = _lib.load()
    n_out = int(lib.loco_resample_length(n_in, up, down))
    y = torch.empty((B, n_out), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.loco_op_resample(C.c_void_p(t.data_ptr()), B, n_in, n_in, C.c_void_p(taps.data_ptr()), up, down, K,
                                        C.c_void_p(y.data_ptr()), n_out, n_out, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                   "loco_op_resample")
    return y[0] if squeeze else y

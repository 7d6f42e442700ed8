"""The C-ABI library loads on a machine without a GPU and exports every symbol include/loco_asr.h declares
(no compute call is made here)."""
import ctypes
import importlib
import os
import re

import pytest

from conftest import ROOT

_libmod = importlib.import_module("loco-asr_amd._lib")


def header_functions():
    text = open(os.path.join(ROOT, "include", "loco_asr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(loco_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built():
    assert os.path.exists(_libmod.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported_and_bound():
    names = header_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(_libmod.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in loco_asr.h but missing from libloco_asr.so"
    assert set(names) == set(_libmod.SIGNATURES), set(names) ^ set(_libmod.SIGNATURES)


def test_host_only_entry_points():
    lib = _libmod.load()
    assert lib.loco_abi_version() == 1
    # floor((n-k)/s)+1 chained over the conv stack (HF modeling:585-598): 5 s -> 249, 30 s -> 1499, 10 min -> 29 999
    assert lib.loco_output_frames(80000) == 249
    assert lib.loco_output_frames(480000) == 1499
    assert lib.loco_output_frames(9600000) == 29999
    assert lib.loco_output_frames(400) == 1 and lib.loco_output_frames(399) == 0
    cfg = _libmod.LocoConfig()
    lib.loco_default_config(ctypes.byref(cfg))
    assert (cfg.hidden, cfg.heads, cfg.ffn, cfg.layers, cfg.conv_dim, cfg.rel_max) == (768, 12, 3072, 12, 512, 160)
    assert cfg.struct_size == ctypes.sizeof(_libmod.LocoConfig)
    assert lib.loco_conv0_scratch_bytes(32) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_libmod, "_lib", None)
    monkeypatch.setattr(_libmod, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _libmod.load()


def test_encoder_refuses_cpu_tensors():
    import torch
    la = importlib.import_module("loco-asr_amd")
    enc = la.SpeechT5EncoderWithSpeechPrenetMI355X(layers=1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc(torch.zeros(1, 1000))
    # state-dict surface matches HF's names (checked against the synthetic generator's key set)
    keys = {"prenet." + k for k in enc.prenet.state_dict()} | {"wrapped_encoder." + k for k in enc.wrapped_encoder.state_dict()}
    assert keys == set(la.synth.encoder_state_dict(0, layers=1))

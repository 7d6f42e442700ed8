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


def test_documented_precision_modes_are_the_accepted_ones():
    """include/loco_asr.h documents the modes in LOCO_PRECISION_MODES; loco_set_precision accepts exactly the modes
    loco_precision_name knows; the Python module offers the same set."""
    text = open(os.path.join(ROOT, "include", "loco_asr.h")).read()
    doc = re.search(r'#define LOCO_PRECISION_MODES "([^"]+)"', text).group(1)
    documented = {int(a): b for a, b in (item.split() for item in doc.split(", "))}
    lib = _libmod.load()
    accepted = {m: lib.loco_precision_name(m).decode() for m in range(-2, 16) if lib.loco_precision_name(m)}
    assert documented == accepted == {0: "f32", 1: "f16x3", 2: "f16x2"}
    la = importlib.import_module("loco-asr_amd")
    assert la.SpeechT5EncoderWithSpeechPrenetMI355X.PRECISIONS == {v: k for k, v in accepted.items()}
    for mode, name in accepted.items():  # each documented mode has its paragraph in the header comment
        assert re.search(rf'\*\s+{mode}\s+"{name}"', text), (mode, name)


def test_status_block_is_a_plain_host_buffer():
    lib = _libmod.load()
    n = lib.loco_status_bytes()
    assert 3000 < n < 16384 and n % 8 == 0
    buf = ctypes.create_string_buffer(n)  # zeros: not a block filled by loco_forward_async
    assert lib.loco_status_check(buf, None, 0) == -1 and b"status block" in lib.loco_last_error()


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

"""Pins the resampler oracle (oracle/resample_oracle.py) WITHOUT librosa / soxr (not installed; their source is not in the
reference): against the library's own fp64 design (loco_resample_design is host code, so this runs without a GPU), against
scipy's polyphase engine driven by the same prototype, and through design-independent properties that any faithful
'soxr_hq'-class converter has -- unity gain in the pass band, > 110 dB rejection of what would alias, unity DC gain, linear
phase (no delay), librosa's output length."""
import ctypes as C
import importlib

import numpy as np
import pytest

import resample_oracle as ro

_lib = importlib.import_module("loco-asr_amd._lib")
RATES = [8000, 11025, 22050, 32000, 44100, 48000]


@pytest.mark.parametrize("sr", RATES)
def test_library_design_matches_the_fp64_restatement(sr):
    lib = _lib.load()
    up, down, K = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.loco_resample_design(sr, 16000, C.byref(up), C.byref(down), C.byref(K), None) == 0
    L, M, Ko, h = ro.design(sr)
    assert (up.value, down.value, K.value) == (L, M, Ko)
    taps = np.empty((L, Ko), np.float32)
    assert lib.loco_resample_design(sr, 16000, C.byref(up), C.byref(down), C.byref(K), taps.ctypes.data_as(C.c_void_p)) == 0
    assert np.abs(taps - h).max() < 1e-7  # fp32 rounding of an fp64 design (numpy's i0 vs the library's series)
    assert np.abs(h.sum(1) - 1.0).max() < 1e-6  # every phase has unity DC gain
    assert lib.loco_resample_length(44100, L, M) == ro.out_length(44100, sr)


def test_output_length_is_librosas():
    for sr, n in ((8000, 12345), (44100, 44100), (44100, 1), (22050, 99999), (48000, 7)):
        assert ro.out_length(n, sr) == int(np.ceil(n * 16000 / sr))


@pytest.mark.parametrize("sr", [8000, 44100])
def test_oracle_agrees_with_scipy_polyphase_on_the_same_prototype(sr):
    """scipy.signal.resample_poly(x, up, down, window=<FIR>) is an independent evaluation of sum_m x[m] h(n M - m L)."""
    from scipy.signal import resample_poly
    L, M, K, h = ro.design(sr)
    # the full prototype in time order: h(t) for t = -(K/2) L ... (K/2) L - 1 is taps[p, j] at t = (j - K/2) L + p
    proto = h.T.reshape(-1)  # index j * L + p
    rng = np.random.default_rng(0)
    x = rng.standard_normal(3000)
    y = ro.resample(x, sr)
    # resample_poly takes the filter's centre at index (len - 1) // 2 and multiplies it by `up` itself; ours has K L taps with
    # the centre at index (K/2) L = K L / 2 and already carries the factor L: append one zero, divide by L
    z = resample_poly(x, L, M, window=np.concatenate([proto, [0.0]]) / L)
    n = min(len(y), len(z))
    assert np.abs(y[:n] - z[:n]).max() < 1e-9 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("sr", RATES)
def test_tone_gain_alias_rejection_and_phase(sr):
    n = sr  # one second
    t = np.arange(n) / sr
    nyq = 0.5 * min(sr, 16000)
    f_pass = 0.5 * nyq
    y = ro.resample(np.sin(2 * np.pi * f_pass * t), sr)
    tt = np.arange(len(y)) / 16000.0
    mid = slice(2000, len(y) - 2000)
    assert np.abs(y[mid] - np.sin(2 * np.pi * f_pass * tt[mid])).max() < 2e-6  # unity gain AND zero delay
    if sr > 16000:  # a tone between the new Nyquist frequency and the old one must vanish instead of aliasing
        f_alias = 0.5 * (8000 + sr / 2)
        ya = ro.resample(np.sin(2 * np.pi * f_alias * t), sr)
        assert 20 * np.log10(np.abs(ya[mid]).max() + 1e-300) < -110.0
    dc = ro.resample(np.ones(n), sr)
    assert np.abs(dc[mid] - 1.0).max() < 1e-6

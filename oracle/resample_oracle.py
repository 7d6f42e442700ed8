"""CPU oracle of the device resampler ("next" row f-4).  TEST INFRASTRUCTURE ONLY (see speecht5_oracle.py header).

The reference resamples on the host with ``librosa.load(path, sr=16000)``
(/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:56); with its pins (librosa 0.10.0.post2, soxr 0.3.5,
requirements.txt:63,138) that is ``soxr.resample(y, sr_native, 16000, quality="soxr_hq")`` followed by
``librosa.util.fix_length(..., size=ceil(n * 16000 / sr_native))``.  Neither library is installed and soxr's source is not in
/root/reference: PARITY UNPINNED against librosa itself.  This file restates, in fp64 numpy, the SPECIFICATION the HIP kernel
implements -- soxr.h's 'HQ' recipe: linear phase, pass band to 0.913 of the lower Nyquist frequency, stop band from 1.0,
20-bit precision -- as one Kaiser-windowed-sinc polyphase filter:

    y[n] = sum_m x[m] h(n M - m L),   L / M = 16000 / sr_in in lowest terms,   h(t) = L fc sinc(fc t) kaiser_beta(t / half)

``design`` mirrors loco_resample_design (csrc/resample.hip) with numpy's own np.i0; ``resample`` evaluates the sum directly.
tests/test_resample_oracle.py pins it with design-independent properties (tone gain, alias rejection, DC gain, agreement with
scipy.signal.resample_poly driven by the same prototype)."""
from math import ceil, gcd, pi

import numpy as np

ATT_DB = 125.0
PASS, STOP = 0.913, 1.0


def design(sr_in: int, sr_out: int = 16000):
    """(L, M, K, taps[L, K] float64).  taps[p, j] weights x[base + K/2 - j], base = floor(n M / L), p = (n M) mod L."""
    g = gcd(sr_in, sr_out)
    L, M = sr_out // g, sr_in // g
    beta = 0.1102 * (ATT_DB - 8.7)
    nyq = 0.5 * min(sr_in, sr_out)
    rate = float(L) * sr_in
    dw = 2 * pi * (STOP - PASS) * nyq / rate
    N = ceil((ATT_DB - 7.95) / (2.285 * dw)) + 1
    K = max(8, (((N + L - 1) // L) + 3) & ~3)
    half = 0.5 * K * L
    fc = (PASS + STOP) * nyq / rate
    j = np.arange(K, dtype=np.float64)[None, :]
    p = np.arange(L, dtype=np.float64)[:, None]
    t = (j - K // 2) * L + p
    r = t / half
    w = np.where(np.abs(r) < 1.0, np.i0(beta * np.sqrt(np.clip(1.0 - r * r, 0.0, None))) / np.i0(beta), 0.0)
    taps = fc * np.sinc(fc * t) * w * L
    return L, M, K, taps


def out_length(n_in: int, sr_in: int, sr_out: int = 16000) -> int:
    """librosa.resample: n_samples = ceil(n * ratio), enforced by fix_length."""
    g = gcd(sr_in, sr_out)
    return (n_in * (sr_out // g) + (sr_in // g) - 1) // (sr_in // g)


def resample(x, sr_in: int, sr_out: int = 16000, taps=None):
    """fp64 evaluation for one clip x [n_in] -> [out_length]; zero extension beyond both ends."""
    x = np.asarray(x, dtype=np.float64)
    L, M, K, h = design(sr_in, sr_out)
    if taps is not None:
        h = np.asarray(taps, dtype=np.float64)
    n_out = out_length(len(x), sr_in, sr_out)
    n = np.arange(n_out, dtype=np.int64)
    base, ph = (n * M) // L, (n * M) % L
    pad = K
    xp = np.concatenate([np.zeros(pad), x, np.zeros(pad + 2)])
    y = np.zeros(n_out)
    for j in range(K):
        idx = base + K // 2 - j
        ok = (idx >= -pad) & (idx < len(x) + pad)
        y += np.where(ok, h[ph, j] * xp[np.clip(idx, -pad, len(x) + pad) + pad], 0.0)
    return y

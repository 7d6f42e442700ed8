"""Device resampler ("next" row f-4, loco_op_resample) against the fp64 oracle of the same specification
(oracle/resample_oracle.py, itself pinned by tests/test_resample_oracle.py).  fp32 FMAs over <= 564 taps: bar 2e-6 of the
signal's peak.  Parity with librosa / soxr themselves is unpinned (neither is installed; include/loco_asr.h says so)."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import la
    import resample_oracle as ro
    rs = importlib.import_module("loco-asr_amd.resample")


def signal(n, sr, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    return (0.3 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 0.45 * min(sr, 16000) * t)).astype(np.float32)


@pytest.mark.parametrize("sr,n", [(8000, 1), (8000, 7), (8000, 4001), (8000, 80000), (11025, 30001), (22050, 66150), (32000, 50000),
                                  (44100, 3), (44100, 441), (44100, 132301), (48000, 96000)])
def test_matches_the_oracle(sr, n):
    x = signal(n, sr, n)
    y = rs.resample_to_16k(x, sr).cpu().numpy()
    ref = ro.resample(x, sr)
    assert y.shape == ref.shape == (ro.out_length(n, sr),) and y.dtype == np.float32
    assert np.abs(y - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())


def test_batch_and_identity_and_errors():
    x = np.stack([signal(50000, 44100, s) for s in range(3)])
    y = rs.resample_to_16k(torch.from_numpy(x).cuda(), 44100)
    assert tuple(y.shape) == (3, ro.out_length(50000, 44100))
    for b in range(3):
        assert np.abs(y[b].cpu().numpy() - ro.resample(x[b], 44100)).max() < 2e-6 * 2
    z = rs.resample_to_16k(x[0], 16000)  # already 16 kHz: librosa.load leaves such files untouched
    assert torch.equal(z.cpu(), torch.from_numpy(x[0]))
    with pytest.raises(RuntimeError):
        rs.resample_to_16k(x[0], 8000, device="cpu")
    with pytest.raises(ValueError):
        rs.resample_to_16k(x[0], 0)


def test_fisher_and_podcast_lengths():
    """configs[2] / configs[3] on real audio start here: ten minutes of 8 kHz telephone speech -> 9 600 000 samples, ten minutes of
    44.1 kHz podcast audio -> the same.  A full fp64 pass over 26 M samples would take minutes, so three windows of 2000 output
    samples (start, middle, end) are recomputed by the oracle from the input slice around them: a slice that starts at a
    multiple of `down` input samples maps onto a whole number of output samples, and 4000 samples of margin keep the slice's
    own zero-extended ends out of the filter's reach."""
    from math import gcd
    for sr in (8000, 44100):
        g = gcd(sr, 16000)
        L, M = 16000 // g, sr // g
        n = 600 * sr
        x = signal(n, sr, 5)
        y = rs.resample_to_16k(x, sr)
        assert y.shape == (9_600_000,) and bool(torch.isfinite(y).all())
        for a in (0, 4_800_000, 9_600_000 - 2000):
            lo = max(0, ((a * M // L - 4000) // M) * M)
            hi = min(n, (a + 2000) * M // L + 4000)
            o0 = lo * L // M  # output index of the slice's first input sample
            ref = ro.resample(x[lo:hi], sr)
            got = y[a:a + 2000].cpu().numpy()
            assert np.abs(got - ref[a - o0:a - o0 + 2000]).max() < 4e-6, (sr, a)


def test_feature_extractor_takes_device_clips_after_resampling(oracle):
    """The extraction loop on 8 kHz input: resample on the device, pad on the device, encode -- equals encoding the oracle's
    16 kHz rendition of the same clips."""
    from gpu_util import model, rel_l2
    clips8 = [signal(12000, 8000, 1), signal(9000, 8000, 2)]
    dev = [rs.resample_to_16k(c, 8000) for c in clips8]
    fe = la.SpeechT5FeatureExtractorMI355X()
    b = fe(audio=dev, sampling_rate=16000)
    assert b["input_values"].is_cuda and tuple(b["input_values"].shape) == (2, 24000) and b["attention_mask"].sum(1).tolist() == [24000, 18000]
    m, sd = model(layers=2)
    y = m.speecht5.encoder(**b).last_hidden_state
    ref_in = fe(audio=[ro.resample(c, 8000).astype(np.float32) for c in clips8], sampling_rate=16000)
    ref = oracle.encode(ref_in["input_values"], ref_in["attention_mask"], la.synth.encoder_state_dict(0, 2))
    assert rel_l2(y, ref) < 1e-4

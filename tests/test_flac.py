"""The FLAC decoder of the corpus side (include/loco_asr.h, loco_flac_decode; row a2: librosa.load of SLURP's .flac recordings,
/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:56).  Bit-exact integer work: every construct of the format
(RFC 9639) is put on the wire by the independent writer in tests/flac_writer.py and must come back sample for sample, with the frame
CRCs and the STREAMINFO MD5 verified by the decoder; damaged streams must be refused, not decoded.  Host code: runs without a GPU."""
import ctypes as C
import importlib
import random

import numpy as np
import pytest

import flac_writer as fw

la = importlib.import_module("loco-asr_amd")
_lib = importlib.import_module("loco-asr_amd._lib")


def decode(data, capacity=None, verify=1):
    lib = _lib.load()
    sr, ch, bits, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    rc = lib.loco_flac_info(data, len(data), C.byref(sr), C.byref(ch), C.byref(bits), C.byref(total))
    if rc:
        return rc, lib.loco_flac_last_error().decode(), None, None
    cap = capacity if capacity is not None else max(1, total.value)
    mono = np.zeros(cap, np.float32)
    pcm = np.zeros((cap, ch.value), np.int32)
    n = C.c_int64()
    rc = lib.loco_flac_decode(data, len(data), mono.ctypes.data_as(C.c_void_p), pcm.ctypes.data_as(C.c_void_p), cap, C.byref(n), verify)
    if rc:
        return rc, lib.loco_flac_last_error().decode(), None, None
    return 0, (sr.value, ch.value, bits.value, total.value), mono[:n.value], pcm[:n.value]


def speech_like(n, bps, channels=1, seed=0):
    x = np.stack([la.synth.clip(seed + c, n) for c in range(channels)], axis=1).astype(np.float64)
    return np.clip(np.round(x * (1 << (bps - 1)) * 0.8), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)


def frames_for(n, size, spec_fn, nch=1, assignment=None):
    out, at, k = [], 0, 0
    while at < n:
        s = min(size, n - at)
        out.append(dict(size=s, assignment=(assignment(k) if assignment else nch - 1), specs=[spec_fn(k, c) for c in range(nch)]))
        at += s
        k += 1
    return out


def test_mono_16_bit_corpus_like_stream_round_trips_and_scales_like_soundfile():
    pcm = speech_like(16000 * 2 + 777, 16)                       # 2.05 s at 16 kHz: 7 blocks of 4096 + a last one of 4105... 
    frames = frames_for(len(pcm), 4096, lambda k, c: dict(kind="fixed", order=k % 5, porder=2 if k % 2 else 0))
    frames[-1]["specs"] = [dict(kind="fixed", order=2, porder=0)]  # the last block's size is not a power-of-two multiple
    data = fw.write_stream(pcm, 16, 16000, frames)
    rc, info, mono, got = decode(data)
    assert rc == 0, info
    assert info == (16000, 1, 16, len(pcm))
    assert np.array_equal(got[:, 0], pcm[:, 0])
    assert np.array_equal(mono, (pcm[:, 0].astype(np.float32) * np.float32(1 / 32768)))  # soundfile's float32 read: exact scaling


def test_every_subframe_kind_stereo_mode_and_residual_coding():
    rng = random.Random(5)
    n = 6 * 1152 + 200 + 37
    pcm = speech_like(n, 16, channels=2, seed=11)
    pcm[2 * 1152:3 * 1152] = (pcm[2 * 1152:3 * 1152] >> 3) << 3    # a block with three wasted bits
    pcm[3 * 1152:4 * 1152, 0] = 321                                 # a constant channel
    pcm[3 * 1152:4 * 1152, 1] = -7

    def lpc(order, prec=12, shift=9):
        return dict(kind="lpc", order=order, precision=prec, shift=shift, coefs=[rng.randint(-(1 << (prec - 3)), (1 << (prec - 3))) for _ in range(order)],
                    porder=1, method=1)

    specs = [
        [dict(kind="fixed", order=4, porder=3), dict(kind="fixed", order=1, porder=0, method=1)],          # left/side
        [lpc(8), lpc(1)],                                                                                   # side/right
        [dict(kind="fixed", order=2, porder=1, wasted=3), dict(kind="verbatim", wasted=3)],                 # independent, wasted bits
        [dict(kind="constant"), dict(kind="constant")],                                                     # independent
        [lpc(32, prec=15, shift=14), dict(kind="fixed", order=3, porder=2, escape_part=1)],                 # mid/side, an escaped partition
        [dict(kind="verbatim"), dict(kind="fixed", order=0, porder=0, escape_part=0)],                      # left/side
        [dict(kind="fixed", order=2, porder=0), lpc(4, prec=5, shift=0)],                                   # block of 200: 8-bit size field
        [dict(kind="fixed", order=1, porder=0), dict(kind="verbatim")],                                     # block of 37
    ]
    sizes = [1152] * 6 + [200, 37]
    assign = [8, 9, 1, 1, 10, 8, 10, 9]
    frames = [dict(size=s, assignment=a, specs=sp) for s, a, sp in zip(sizes, assign, specs)]
    data = fw.write_stream(pcm, 16, 44100, frames, id3=True)
    rc, info, mono, got = decode(data)
    assert rc == 0, info
    assert info == (44100, 2, 16, n)
    assert np.array_equal(got, pcm)
    want = (pcm[:, 0].astype(np.float32) * np.float32(1 / 32768) + pcm[:, 1].astype(np.float32) * np.float32(1 / 32768)) / np.float32(2)
    assert np.array_equal(mono, want)                               # soundfile(..., always_2d=True).mean(axis=1)


@pytest.mark.parametrize("bps", [8, 12, 20, 24, 32])
def test_sample_sizes(bps):
    n = 3 * 576 + 100
    pcm = speech_like(n, bps, channels=2 if bps < 32 else 1, seed=bps)
    nch = pcm.shape[1]
    frames = frames_for(n, 576, lambda k, c: dict(kind="fixed", order=(k + c) % 3 + 1, porder=k % 3, method=1 if bps > 16 else 0), nch=nch,
                        assignment=(lambda k: [10, 8, 9, 1][k % 4]) if nch == 2 else None)
    if bps == 32:
        # a residual must fit 32 bits signed (RFC 9639 section 9.2.7.3; the decoder refuses wider ones): the full range incl. both
        # extremes travels in a VERBATIM block, the predicted blocks carry half-scale samples under orders 0 / 1
        pcm[0, 0], pcm[1, 0] = -(1 << 31), (1 << 31) - 1
        pcm[576:] //= 2
        for k, f in enumerate(frames):
            f["specs"] = [dict(kind="verbatim")] if k == 0 else [dict(kind="fixed", order=k % 2, porder=k % 3, method=1)]
    for f in frames:
        f["bps_from_streaminfo"] = bps not in fw.BPS_CODES or f["size"] == 100
        if f["size"] % 8:
            for sp in f["specs"]:
                sp["porder"] = 0
    data = fw.write_stream(pcm, bps, 8000, frames, extra_metadata=False)
    rc, info, mono, got = decode(data)
    assert rc == 0, info
    assert info == (8000, nch, bps, n)
    assert np.array_equal(got.astype(np.int64), pcm)
    scale = np.float32(1.0 / (1 << (bps - 1)))
    want = pcm[:, 0].astype(np.float32) * scale if nch == 1 else (pcm[:, 0].astype(np.float32) * scale + pcm[:, 1].astype(np.float32) * scale) / np.float32(2)
    assert np.array_equal(mono, want)


def test_a_residual_wider_than_32_bits_is_refused():
    """RFC 9639 section 9.2.7.3: residuals fit 32 bits signed.  A third-order predictor on full-scale 32-bit noise produces wider ones;
    the stream carries valid CRCs and MD5, and the decoder says no instead of computing with them."""
    pcm = speech_like(576, 32, seed=5)
    pcm[::2] = (1 << 31) - 1
    pcm[1::2] = -(1 << 31)
    data = fw.write_stream(pcm, 32, 16000, [dict(size=576, specs=[dict(kind="fixed", order=3, porder=0, method=1)])])
    rc, msg, _, _ = decode(data)
    assert rc != 0 and "malformed subframe" in msg, (rc, msg)


def test_unknown_length_variable_block_sizes_and_long_coded_numbers():
    pcm = speech_like(5000, 16)
    sizes = [256, 1000, 4096 - 1256 - 200, 200, 5000 - 4096]
    frames = [dict(size=s, specs=[dict(kind="fixed", order=2, porder=0)]) for s in sizes]
    for off in (0, 1 << 20, (1 << 31) + 5, (1 << 35) + 9):            # 1- to 7-byte coded sample numbers
        data = fw.write_stream(pcm, 16, 16000, frames, total_known=False, variable=True, number_offset=off)
        rc, info, mono, got = decode(data, capacity=6000)
        assert rc == 0, info
        assert info[3] == 0 and np.array_equal(got[:, 0], pcm[:, 0])
    rc, msg, _, _ = decode(data, capacity=4999)
    assert rc == -3 and "capacity" in msg
    # no MD5 signature stored (all zero): nothing to verify, still decodes
    data = fw.write_stream(pcm, 16, 16000, frames, md5=False)
    rc, info, _, got = decode(data)
    assert rc == 0 and np.array_equal(got[:, 0], pcm[:, 0])


def test_damaged_streams_are_refused():
    pcm = speech_like(3000, 16)
    frames = frames_for(len(pcm), 1024, lambda k, c: dict(kind="fixed", order=2, porder=1))
    data = bytearray(fw.write_stream(pcm, 16, 16000, frames))
    audio = data.index(b"\xff\xf8")
    for where, what in ((audio + 40, "CRC-16"), (audio + 2, "CRC-8"), (len(data) - 200, "CRC-16")):
        bad = bytearray(data)
        bad[where] ^= 0x10
        rc, msg, _, _ = decode(bytes(bad))
        assert rc == -1 and what in msg, (where, msg)
    bad = bytearray(data)
    bad[4 + 4 + 18] ^= 0xFF                                            # the stored MD5
    rc, msg, _, _ = decode(bytes(bad))
    assert rc == -1 and "MD5" in msg
    rc, info, _, _ = decode(bytes(bad), verify=0)                      # not asked to verify: decodes
    assert rc == 0
    rc, msg, _, _ = decode(bytes(data[:len(data) - 300]))
    assert rc == -1 and ("truncated" in msg or "ends after" in msg or "CRC" in msg or "synchronisation" in msg), msg
    rc, msg, _, _ = decode(b"RIFF" + bytes(100))
    assert rc == -1 and "fLaC" in msg


def test_the_cli_loader_reads_flac_without_soundfile(tmp_path):
    extract = importlib.import_module("loco-asr_amd.extract")
    pcm = speech_like(16000, 16, channels=2, seed=3)
    frames = frames_for(len(pcm), 4096, lambda k, c: dict(kind="fixed", order=2, porder=2), nch=2, assignment=lambda k: 10)
    for f in frames:
        if f["size"] % 4:
            for sp in f["specs"]:
                sp["porder"] = 0
    path = tmp_path / "audio-1-headset.flac"
    path.write_bytes(fw.write_stream(pcm, 16, 16000, frames))
    x = extract.load_audio_16k(str(path))
    assert isinstance(x, np.ndarray) and x.dtype == np.float32 and x.shape == (16000,)
    want = (pcm[:, 0].astype(np.float32) / np.float32(32768) + pcm[:, 1].astype(np.float32) / np.float32(32768)) / np.float32(2)
    assert np.array_equal(x, want)


def test_random_streams():
    """40 seeded random streams: channel count, sample size, block sizes, channel assignment, subframe kind, predictor order, LPC
    precision / shift / coefficients, Rice method, partition order and escaped partitions all drawn at random."""
    rng = random.Random(2024)
    for case in range(40):
        bps = rng.choice([8, 12, 16, 16, 16, 20, 24])
        nch = rng.choice([1, 2, 2])
        rate = rng.choice([8000, 16000, 22050, 44100, 12345, 48000])
        n = rng.randint(300, 5000)
        pcm = speech_like(n, bps, channels=nch, seed=100 + case)
        frames, at = [], 0
        while at < n:
            size = min(n - at, rng.choice([192, 256, 576, 1000, 1024, 1152, 77, 2048]))
            specs = []
            for c in range(nch):
                kind = rng.choice(["fixed", "fixed", "lpc", "verbatim"])
                sp = dict(kind=kind)
                if kind == "fixed":
                    sp["order"] = min(rng.randint(0, 4), size)
                elif kind == "lpc":
                    sp["order"] = min(rng.choice([1, 2, 3, 6, 12, 32]), size)
                    sp["precision"] = rng.randint(3, 15)
                    # random (unfitted) coefficients predict up to 2^(precision - 1 - shift) times the signal: the shift keeps the residual
                    # inside 32 bits signed, as the format demands of an encoder (the decoder refuses wider residuals)
                    sp["shift"] = rng.randint(max(0, bps + sp["precision"] - 31), sp["precision"])
                    lim = (1 << (sp["precision"] - 1)) // max(1, sp["order"])
                    sp["coefs"] = [rng.randint(-lim, max(0, lim - 1)) for _ in range(sp["order"])]
                if kind != "verbatim":
                    po = rng.randint(0, 3)
                    while po and (size % (1 << po) or (size >> po) < sp["order"]):
                        po -= 1
                    sp["porder"] = po
                    sp["method"] = rng.randint(0, 1)
                    if rng.random() < 0.2:
                        sp["escape_part"] = rng.randrange(1 << po)
                specs.append(sp)
            frames.append(dict(size=size, assignment=rng.choice([1, 8, 9, 10]) if nch == 2 else 0, specs=specs,
                               bps_from_streaminfo=rng.random() < 0.3))
            at += size
        data = fw.write_stream(pcm, bps, rate, frames, id3=rng.random() < 0.3, extra_metadata=rng.random() < 0.5, variable=rng.random() < 0.5)
        rc, info, mono, got = decode(data)
        assert rc == 0, (case, info)
        assert info == (rate, nch, bps, n), (case, info)
        assert np.array_equal(got.astype(np.int64), pcm), case

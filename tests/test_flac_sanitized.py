"""The FLAC decoder parses files: bytes it does not control.  This test builds loco-asr_amd/csrc/flac_decode.hip (host-only C++) with
g++ -fsanitize=address,undefined -fno-sanitize-recover=all next to a small driver (tests/native/flac_fuzz_driver.cc) and feeds it a
few thousand damaged streams: random byte damage anywhere (a frame is parsed BEFORE its CRC-16 can be checked), damage inside frame
bodies and headers with both CRCs recomputed (what a crafted file would carry: the stereo / MD5 / output code then runs on garbage),
damaged STREAMINFO fields, every truncation of a small stream.  Any out-of-bounds access, signed overflow or invalid shift aborts
the driver; every refusal must come with a message; a good stream must still decode.  CPU only (sanitizers run on the CPU build)."""
import os
import random
import shutil
import struct
import subprocess

import numpy as np
import pytest

import flac_writer as fw
from test_flac import frames_for, speech_like

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def frame_spans(data):
    """(offset, header bytes incl. CRC-8, length) of every frame of a stream written by flac_writer: the sync code plus a matching
    CRC-8 finds the header, the matching CRC-16 the end (quadratic, fine for test-sized streams)."""
    spans, o = [], data.index(b"fLaC")
    while True:  # skip the metadata blocks
        o_hdr = o + 4 if not spans and data[o:o + 4] == b"fLaC" else o
        last, ln = data[o_hdr] & 0x80, int.from_bytes(data[o_hdr + 1:o_hdr + 4], "big")
        o = o_hdr + 4 + ln
        if last:
            break
    while o + 2 <= len(data):
        assert data[o] == 0xFF and data[o + 1] & 0xFE == 0xF8, o
        hdr = next(h for h in range(5, 17) if fw.crc8(data[o:o + h]) == data[o + h]) + 1
        c, end = 0, None
        for e in range(o, len(data) - 1):  # running CRC-16 of data[o:e]
            if e >= o + hdr + 1 and c == int.from_bytes(data[e:e + 2], "big") and (e + 2 == len(data) or (data[e + 2] == 0xFF and data[e + 3] & 0xFE == 0xF8)):
                end = e
                break
            c ^= data[e] << 8
            for _ in range(8):
                c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
        assert end is not None, o
        spans.append((o, hdr, end + 2 - o))
        o = end + 2
    return spans


def seed_streams():
    rng = random.Random(99)
    out = []
    for case in range(6):
        bps = [16, 8, 24, 16, 12, 32][case]
        nch = [1, 2, 2, 1, 2, 1][case]
        n = [700, 900, 400, 1500, 600, 300][case]
        pcm = speech_like(n, bps, channels=nch, seed=40 + case)
        if bps == 32:
            pcm = pcm // 2  # LPC on full-scale 32-bit samples would need more than 64 bits in the WRITER's plain arithmetic

        def spec(k, c, case=case, bps=bps):
            kind = ["fixed", "lpc", "verbatim", "fixed", "lpc", "lpc"][(k + c + case) % 6]
            sp = dict(kind=kind, porder=(k + case) % 3, method=1 if bps > 20 else (k + c) % 2)  # 4-bit Rice parameters stop at 14: unary runs of 2^16 for 32-bit residuals
            if kind == "fixed":
                sp["order"] = (k + c) % 5
            if kind == "lpc":
                sp.update(order=[1, 2, 8, 12][(k + c) % 4], precision=12, shift=9)
                sp["coefs"] = [rng.randint(-200, 199) for _ in range(sp["order"])]
            if kind != "verbatim" and (k + c) % 4 == 1:
                sp["escape_part"] = 0
            return sp
        size = [192, 256, 100, 576, 128, 64][case]
        frames = frames_for(n, size, spec, nch=nch, assignment=(lambda k: [1, 8, 9, 10][k % 4]) if nch == 2 else None)
        for f in frames:  # partition orders must divide the block and leave the warm-up samples in partition 0
            for sp in f["specs"]:
                while sp.get("porder", 0) and (f["size"] % (1 << sp["porder"]) or (f["size"] >> sp["porder"]) < sp.get("order", 0)):
                    sp["porder"] -= 1
                if "escape_part" in sp:
                    sp["escape_part"] = 0
                if sp.get("order", 0) > f["size"]:
                    sp["order"] = min(sp["order"], f["size"])
                    if "coefs" in sp:
                        sp["coefs"] = sp["coefs"][:sp["order"]]
        out.append(fw.write_stream(pcm, bps, 16000, frames, id3=case == 3, variable=case % 2 == 1, total_known=case != 4))
    return out


def damaged(seeds, rng):
    for s in seeds:
        yield s  # the good stream itself: must decode
        spans = frame_spans(s)
        audio0 = spans[0][0]
        for _ in range(150):  # raw damage anywhere: 1-3 bytes
            b = bytearray(s)
            for _ in range(rng.randint(1, 3)):
                b[rng.randrange(len(b))] = rng.randrange(256)
            yield bytes(b)
        for _ in range(250):  # damage inside ONE frame with both CRCs put right again
            o, hdr, ln = spans[rng.randrange(len(spans))]
            b = bytearray(s)
            for _ in range(rng.randint(1, 4)):
                where = rng.random()
                if where < 0.25:   # header fields behind the sync code (block size / rate / channel / sample size codes, coded number)
                    at = o + rng.randrange(2, hdr - 1)
                elif where < 0.6:  # the first bytes of the body: subframe headers, warm-up samples, LPC precision / shift, Rice parameters
                    at = o + hdr + rng.randrange(0, min(24, ln - hdr - 2))
                else:
                    at = o + hdr + rng.randrange(0, ln - hdr - 2)
                b[at] = rng.randrange(256) if rng.random() < 0.7 else b[at] ^ (1 << rng.randrange(8))
            b[o + hdr - 1] = fw.crc8(bytes(b[o:o + hdr - 1]))
            b[o + ln - 2:o + ln] = struct.pack(">H", fw.crc16(bytes(b[o:o + ln - 2])))
            yield bytes(b)
        for _ in range(60):  # STREAMINFO fields (block sizes, rate, channels, bits per sample, total, MD5) and metadata lengths
            b = bytearray(s)
            at = rng.randrange(s.index(b"fLaC"), audio0)
            b[at] = rng.randrange(256)
            yield bytes(b)
    small = seeds[1]
    for cut in range(len(small)):  # every truncation
        yield small[:cut]
    for fill in (0x00, 0xFF):  # long runs of zeros / ones where a unary code is expected
        o, hdr, ln = frame_spans(seeds[0])[0]
        b = bytearray(seeds[0])
        b[o + hdr + 2:o + ln - 2] = bytes([fill]) * (ln - hdr - 4)
        b[o + ln - 2:o + ln] = struct.pack(">H", fw.crc16(bytes(b[o:o + ln - 2])))
        yield bytes(b)
        yield bytes(b[:o + hdr + 2]) + bytes([fill]) * 300000


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_damaged_streams_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    exe = tmp_path / "flac_fuzz"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-x", "c++", os.path.join(ROOT, "loco-asr_amd/csrc/flac_decode.hip"), os.path.join(ROOT, "tests/native/flac_fuzz_driver.cc"),
           "-o", str(exe)]
    subprocess.run(cmd, check=True, cwd=ROOT)
    seeds = seed_streams()
    rng = random.Random(4242)
    corpus = tmp_path / "corpus.bin"
    n = 0
    with open(corpus, "wb") as fh:
        for s in damaged(seeds, rng):
            fh.write(struct.pack("<I", len(s)))
            fh.write(s)
            n += 1
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe), str(corpus)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    fields = dict(kv.split("=") for kv in r.stdout.split())
    assert int(fields["streams"]) == n
    # the good streams decode; most damaged ones are refused, some damage is harmless (padding, unused metadata) or -- with the CRCs
    # put right and no MD5 check -- decodes to other samples: what matters is that nothing above tripped a sanitizer
    assert int(fields["decoded"]) >= len(seeds)
    assert int(fields["refused"]) > n // 2
    print(r.stdout.strip())

"""TEST INFRASTRUCTURE: an independent, deliberately plain Python WRITER of FLAC streams (RFC 9639) for tests/test_flac.py.

No FLAC encoder or file exists in the build image, so the streams the decoder in loco-asr_amd/csrc/flac_decode.hip is checked with are
written here, from the same format specification but sharing no code with it (bit WRITER vs bit reader, hashlib's MD5 vs the
decoder's own, table-free CRCs).  It is not an efficient encoder -- it is a way to put every construct of the format on the wire
under the test's control: CONSTANT / VERBATIM / FIXED orders 0-4 / LPC subframes, Rice partitions of order 0-3 with 4- and 5-bit
parameters and escaped partitions, wasted bits, the four channel assignments, fixed and variable block sizes, 8-32 bits per sample,
block sizes coded by table entry, by 8 and by 16 extra bits, an ID3v2 tag in front, PADDING / VORBIS_COMMENT metadata blocks."""
import hashlib
import struct

import numpy as np


class BitWriter:
    def __init__(self):
        self.bytes = bytearray()
        self.acc = 0
        self.n = 0

    def write(self, value, bits):
        if bits == 0:
            return
        value &= (1 << bits) - 1
        self.acc = (self.acc << bits) | value
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.bytes.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, q):  # q zeros, then a one
        while q >= 32:
            self.write(0, 32)
            q -= 32
        self.write(1, q + 1)

    def align(self):
        if self.n:
            self.write(0, 8 - self.n)

    def getvalue(self):
        assert self.n == 0
        return bytes(self.bytes)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v):
    """the 'UTF-8-like' coding of the frame / sample number (up to 36 bits)"""
    if v < 0x80:
        return bytes([v])
    out = []
    n = 1
    while True:
        n += 1
        lead_bits = 7 - n
        if v < (1 << (6 * (n - 1) + lead_bits)):
            break
    for i in range(n - 1):
        out.append(0x80 | (v & 0x3F))
        v >>= 6
    lead = ((0xFF << (8 - n)) & 0xFF) | v
    return bytes([lead] + out[::-1])


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def residual_of(s, order, coefs, shift):
    res = []
    for i in range(order, len(s)):
        pred = sum(c * s[i - 1 - j] for j, c in enumerate(coefs))
        res.append(s[i] - (pred >> shift))
    return res


def write_residual(bw, res, blocksize, order, porder, method, escape_part=None):
    bw.write(method, 2)
    bw.write(porder, 4)
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    i = 0
    for part in range(1 << porder):
        count = (blocksize >> porder) - (order if part == 0 else 0)
        chunk = res[i:i + count]
        i += count
        if escape_part is not None and part == escape_part:
            nb = max([1] + [(v.bit_length() if v >= 0 else (~v).bit_length()) + 1 for v in chunk])
            bw.write(esc, pbits)
            bw.write(nb, 5)
            for v in chunk:
                bw.write(v, nb)
            continue
        mean = sum(abs(v) for v in chunk) / max(1, len(chunk))
        k = min(esc - 1, max(0, int(mean).bit_length()))
        bw.write(k, pbits)
        for v in chunk:
            u = (v << 1) if v >= 0 else ((-v) << 1) - 1  # zigzag
            bw.unary(u >> k)
            bw.write(u & ((1 << k) - 1), k)
    assert i == len(res)


def write_subframe(bw, samples, bps, spec):
    """spec: dict(kind='constant'|'verbatim'|'fixed'|'lpc', order=, coefs=, shift=, precision=, porder=, method=, escape_part=, wasted=)"""
    s = [int(v) for v in samples]
    wasted = spec.get("wasted", 0)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in s)
        s = [v >> wasted for v in s]
    kind = spec["kind"]
    order = spec.get("order", 0)
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + order
    elif kind == "lpc":
        code = 32 + order - 1
    bw.write(0, 1)
    bw.write(code, 6)
    if wasted:
        bw.write(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.write(0, 1)
    b = bps - wasted
    if kind == "constant":
        assert len(set(s)) == 1
        bw.write(s[0], b)
    elif kind == "verbatim":
        for v in s:
            bw.write(v, b)
    else:
        coefs, shift = (FIXED[order], 0) if kind == "fixed" else (spec["coefs"], spec["shift"])
        for v in s[:order]:
            bw.write(v, b)
        if kind == "lpc":
            prec = spec["precision"]
            bw.write(prec - 1, 4)
            bw.write(shift, 5)
            for c in coefs:
                bw.write(c, prec)
        res = residual_of(s, order, coefs, shift)
        write_residual(bw, res, len(s), order, spec.get("porder", 0), spec.get("method", 0), spec.get("escape_part"))


BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
BPS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def write_frame(channels, bps, rate, number, assignment, specs, variable=False, bps_from_streaminfo=False):
    """channels: list of per-channel int lists (already decorrelated by the caller for assignments 8-10)"""
    blocksize = len(channels[0])
    bw = BitWriter()
    bw.write(0b11111111111110, 14)
    bw.write(0, 1)
    bw.write(1 if variable else 0, 1)
    if blocksize in BLOCK_CODES:
        bcode = BLOCK_CODES[blocksize]
    else:
        bcode = 6 if blocksize <= 256 else 7
    bw.write(bcode, 4)
    rates = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
    rcode = rates.get(rate, 0 if rate % 10 else 14)
    if rcode == 14 and rate // 10 > 0xFFFF:
        rcode = 0
    bw.write(rcode, 4)
    bw.write(assignment, 4)
    bw.write(0 if bps_from_streaminfo else BPS_CODES[bps], 3)
    bw.write(0, 1)
    for byte in utf8_number(number):
        bw.write(byte, 8)
    if bcode == 6:
        bw.write(blocksize - 1, 8)
    elif bcode == 7:
        bw.write(blocksize - 1, 16)
    if rcode == 14:
        bw.write(rate // 10, 16)
    head = bw.getvalue()
    bw.write(crc8(head), 8)
    for c, (ch, spec) in enumerate(zip(channels, specs)):
        side = (assignment == 8 and c == 1) or (assignment == 9 and c == 0) or (assignment == 10 and c == 1)
        write_subframe(bw, ch, bps + (1 if side else 0), spec)
    bw.align()
    body = bw.getvalue()
    return body + struct.pack(">H", crc16(body))


def decorrelate(left, right, assignment):
    left, right = [int(v) for v in left], [int(v) for v in right]
    if assignment == 8:
        return [left, [a - b for a, b in zip(left, right)]]
    if assignment == 9:
        return [[a - b for a, b in zip(left, right)], right]
    if assignment == 10:
        return [[(a + b) >> 1 for a, b in zip(left, right)], [a - b for a, b in zip(left, right)]]
    return [left, right]


def write_stream(pcm, bps, rate, frames, id3=False, extra_metadata=True, total_known=True, md5=True, variable=False, number_offset=0):
    """pcm: int array [n, channels]; frames: list of dict(size=, assignment=, specs=[per-channel spec]) covering the n samples in order."""
    pcm = np.asarray(pcm, dtype=np.int64)
    n, nch = pcm.shape
    out = bytearray()
    if id3:
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, 10]) + b"\x00" * 10
    out += b"fLaC"
    nbytes = (bps + 7) // 8
    raw = b"".join(int(v).to_bytes(nbytes, "little", signed=True) for v in pcm.reshape(-1))
    sizes = [f["size"] for f in frames]
    bw = BitWriter()
    bw.write(min(sizes[:-1] or sizes), 16)
    bw.write(max(sizes), 16)
    bw.write(0, 24)
    bw.write(0, 24)
    bw.write(rate, 20)
    bw.write(nch - 1, 3)
    bw.write(bps - 1, 5)
    bw.write(n if total_known else 0, 36)
    info = bw.getvalue() + (hashlib.md5(raw).digest() if md5 else b"\x00" * 16)
    blocks = [(0, info)]
    if extra_metadata:
        vendor = b"loco-asr test writer"
        blocks.append((4, struct.pack("<I", len(vendor)) + vendor + struct.pack("<I", 0)))
        blocks.append((1, b"\x00" * 37))
    for i, (t, body) in enumerate(blocks):
        out += bytes([(0x80 if i == len(blocks) - 1 else 0) | t]) + len(body).to_bytes(3, "big") + body
    at = 0
    for k, f in enumerate(frames):
        size, assignment = f["size"], f.get("assignment", nch - 1)
        chunk = pcm[at:at + size]
        pad = size - len(chunk)  # a last frame may be declared longer than the samples left only if total is known (decoder truncates)
        if pad:
            chunk = np.concatenate([chunk, np.zeros((pad, nch), np.int64)])
        if nch == 2:
            chans = decorrelate(chunk[:, 0], chunk[:, 1], assignment)
        else:
            chans = [[int(v) for v in chunk[:, c]] for c in range(nch)]
        out += write_frame(chans, bps, rate, number_offset + (at if variable else k), assignment, f["specs"], variable=variable,
                           bps_from_streaminfo=f.get("bps_from_streaminfo", False))
        at += size
    return bytes(out)

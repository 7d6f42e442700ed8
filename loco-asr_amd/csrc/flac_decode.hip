// FLAC decoder for the corpus side of the path (host code only: no kernel in this file).
//
// SLURP's recordings (audio/slurp_real/*.flac) reach the reference through librosa.load(path, sr=16000)
// (/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:56) -> soundfile -> libsndfile -> libFLAC; none of those is
// part of this image, and scipy reads WAV only.  FLAC is a bit-exact integer format (RFC 9639, "Free Lossless Audio Codec"), so the
// decoder is restated here from the format specification: metadata (STREAMINFO), frame headers (CRC-8), the four subframe kinds
// (CONSTANT, VERBATIM, FIXED orders 0-4, LPC orders 1-32), partitioned Rice residuals incl. the escape code, wasted bits, the three
// stereo decorrelations, the frame CRC-16 and the MD5 signature of the decoded samples that every encoder stores in STREAMINFO --
// which makes each real file its own known-answer test (loco_flac_decode verifies it on request).  The output is what
// soundfile.read(dtype="float32", always_2d=True).mean(axis=1) hands the reference: sample / 2^(bits-1), channels averaged.
// PARITY NOTE: no libFLAC-encoded file is available offline; tests/test_flac.py pins the decoder with streams written by an
// independent Python encoder of the same specification (tests/flac_writer.py) that exercises every subframe kind, and with the
// CRC / MD5 structure of the format itself.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/loco_asr.h"

namespace {

thread_local std::string g_flac_err;
int ffail(int code, const std::string& msg) {
    g_flac_err = msg;
    return code;
}

// MSB-first bit reader over a 64-bit window (the top `cnt` bits of `acc` are the next bits of the stream, the rest are zero): a Rice
// code is one count-leading-zeros and one shift, not a loop over bits.
struct BitReader {
    const uint8_t* p;
    size_t n, byte = 0;
    uint64_t acc = 0;
    int cnt = 0;
    bool bad = false;
    BitReader(const uint8_t* d, size_t bytes) : p(d), n(bytes) {}
    void refill() {
        if (cnt > 56) return;
        if (byte + 8 <= n) {
            // one unaligned big-endian load tops the window up to 56 .. 63 bits: whole bytes only are taken from p, the fraction of the
            // following byte that the OR brings in below them is masked off again (unary() relies on zeros below `cnt`)
            uint64_t w;
            memcpy(&w, p + byte, 8);
            acc |= __builtin_bswap64(w) >> cnt;
            byte += (size_t)((63 - cnt) >> 3);
            cnt |= 56;
            acc &= ~(~0ull >> cnt);
            return;
        }
        while (cnt <= 56 && byte < n) {
            acc |= (uint64_t)p[byte++] << (56 - cnt);
            cnt += 8;
        }
    }
    size_t pos() const { return byte * 8 - (size_t)cnt; }  // bits consumed
    uint64_t bits(int k) {                                  // k <= 64
        if (k == 0) return 0;
        if (k > 32) {
            const uint64_t hi = bits(k - 32);
            return (hi << 32) | bits(32);
        }
        if (cnt < k) {
            refill();
            if (cnt < k) { bad = true; return 0; }
        }
        const uint64_t v = acc >> (64 - k);
        acc <<= k;
        cnt -= k;
        return v;
    }
    uint32_t bit() { return (uint32_t)bits(1); }
    int64_t sbits(int k) {  // two's complement, 1 <= k <= 33
        const uint64_t v = bits(k);
        const uint64_t sign = 1ull << (k - 1);
        return (int64_t)((v ^ sign) - sign);
    }
    uint32_t unary() {  // number of 0 bits before the next 1
        uint32_t q = 0;
        for (;;) {
            refill();
            if (cnt == 0) { bad = true; return q; }
            if (acc == 0) { q += (uint32_t)cnt; cnt = 0; continue; }
            const int z = __builtin_clzll(acc);  // < cnt: the bits below the window are zero
            q += (uint32_t)z;
            acc <<= z;
            acc <<= 1;
            cnt -= z + 1;
            return q;
        }
    }
    void align() {
        const int r = (int)(pos() & 7);
        if (r) bits(8 - r);
    }
};

struct CrcTables {
    uint8_t t8[256];
    uint16_t t16[8][256];  // t16[j][v] = CRC of byte v followed by j zero bytes: eight bytes per step (every byte of a stream passes the CRC-16)
    CrcTables() {
        for (int v = 0; v < 256; ++v) {
            uint8_t c = (uint8_t)v;
            for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));  // x^8 + x^2 + x + 1
            t8[v] = c;
            uint16_t w = (uint16_t)(v << 8);
            for (int b = 0; b < 8; ++b) w = (uint16_t)((w & 0x8000) ? (w << 1) ^ 0x8005 : (w << 1));  // x^16 + x^15 + x^2 + 1
            t16[0][v] = w;
        }
        for (int j = 1; j < 8; ++j)
            for (int v = 0; v < 256; ++v) t16[j][v] = (uint16_t)((t16[j - 1][v] << 8) ^ t16[0][t16[j - 1][v] >> 8]);
    }
};
const CrcTables& crc_tables() {
    static const CrcTables t;
    return t;
}
uint8_t crc8(const uint8_t* d, size_t n) {  // init 0, MSB first
    const CrcTables& t = crc_tables();
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) c = t.t8[c ^ d[i]];
    return c;
}
uint16_t crc16(const uint8_t* d, size_t n) {  // init 0, MSB first
    const CrcTables& t = crc_tables();
    uint16_t c = 0;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {  // the running CRC enters the first two bytes, then eight independent table reads
        const uint8_t b0 = (uint8_t)(d[i] ^ (c >> 8)), b1 = (uint8_t)(d[i + 1] ^ (c & 0xff));
        c = (uint16_t)(t.t16[7][b0] ^ t.t16[6][b1] ^ t.t16[5][d[i + 2]] ^ t.t16[4][d[i + 3]] ^ t.t16[3][d[i + 4]] ^ t.t16[2][d[i + 5]] ^
                       t.t16[1][d[i + 6]] ^ t.t16[0][d[i + 7]]);
    }
    for (; i < n; ++i) c = (uint16_t)((c << 8) ^ t.t16[0][(c >> 8) ^ d[i]]);
    return c;
}

// MD5 (RFC 1321), streaming
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t len = 0;
    uint8_t buf[64];
    size_t fill = 0;
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* m) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
            0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
            0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
            0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
            0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
            0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)m[4 * i] | ((uint32_t)m[4 * i + 1] << 8) | ((uint32_t)m[4 * i + 2] << 16) | ((uint32_t)m[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f;
            int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D;
            D = C;
            C = B;
            B = B + rol(A + f + K[i] + w[g], S[i]);
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        while (n) {
            const size_t take = (64 - fill) < n ? (64 - fill) : n;
            memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void finish(uint8_t out[16]) {
        const uint64_t bitlen = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t l[8];
        for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bitlen >> (8 * i));
        update(l, 8);
        const uint32_t v[4] = {a, b, c, d};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(v[i] >> (8 * j));
    }
};

struct StreamInfo {
    int min_block = 0, max_block = 0, rate = 0, channels = 0, bps = 0;
    int64_t total = 0;
    uint8_t md5[16] = {0};
    size_t audio_offset = 0;  // first frame
};

int parse_header(const uint8_t* d, size_t n, StreamInfo& si) {
    size_t o = 0;
    if (n >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3') {  // an ID3v2 tag in front of the stream: skip it
        const size_t sz = ((size_t)(d[6] & 0x7f) << 21) | ((size_t)(d[7] & 0x7f) << 14) | ((size_t)(d[8] & 0x7f) << 7) | (size_t)(d[9] & 0x7f);
        o = 10 + sz;
    }
    if (o + 4 > n || memcmp(d + o, "fLaC", 4) != 0) return ffail(LOCO_E_INVALID, "not a FLAC stream (no fLaC marker)");
    o += 4;
    bool have_info = false;
    for (;;) {
        if (o + 4 > n) return ffail(LOCO_E_INVALID, "FLAC: truncated metadata");
        const bool last = (d[o] & 0x80) != 0;
        const int type = d[o] & 0x7f;
        const size_t len = ((size_t)d[o + 1] << 16) | ((size_t)d[o + 2] << 8) | d[o + 3];
        o += 4;
        if (o + len > n) return ffail(LOCO_E_INVALID, "FLAC: truncated metadata block");
        if (type == 0) {
            if (len < 34) return ffail(LOCO_E_INVALID, "FLAC: short STREAMINFO");
            BitReader br(d + o, len);
            si.min_block = (int)br.bits(16);
            si.max_block = (int)br.bits(16);
            br.bits(24); br.bits(24);
            si.rate = (int)br.bits(20);
            si.channels = (int)br.bits(3) + 1;
            si.bps = (int)br.bits(5) + 1;
            si.total = (int64_t)br.bits(36);
            memcpy(si.md5, d + o + 18, 16);
            have_info = true;
        } else if (type == 127) {
            return ffail(LOCO_E_INVALID, "FLAC: invalid metadata block type 127");
        }
        o += len;
        if (last) break;
    }
    if (!have_info) return ffail(LOCO_E_INVALID, "FLAC: no STREAMINFO block");
    if (si.rate <= 0 || si.bps < 4 || si.bps > 32) return ffail(LOCO_E_INVALID, "FLAC: unsupported STREAMINFO (sample rate / bits per sample)");
    si.audio_offset = o;
    return LOCO_OK;
}

// Sample arithmetic wraps (two's complement in uint64_t) instead of overflowing: a well-formed stream never comes near 2^63 (<= 33-bit
// samples, 32-bit residuals, 15-bit coefficients), but a frame is PARSED before its CRC-16 can be checked, and a crafted file carries
// valid CRCs anyway -- predictor feedback over 65 536 samples of garbage must stay defined behaviour (tests/test_flac_sanitized.py
// runs mutated streams through an ASan + UBSan build of this file).
inline int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
inline int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
inline int64_t wmul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

__attribute__((always_inline)) inline bool read_residual_body(BitReader& br, int64_t* s, int blocksize, int order) {
    const int method = (int)br.bits(2);
    if (method > 1) return false;
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = (int)br.bits(4);
    const int parts = 1 << porder;
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
    int i = order;
    for (int part = 0; part < parts; ++part) {
        int count = blocksize >> porder;
        if (part == 0) count -= order;
        if (count < 0) return false;
        const int k = (int)br.bits(pbits);
        if (k == esc) {
            const int nb = (int)br.bits(5);
            for (int j = 0; j < count; ++j) s[i++] = nb ? br.sbits(nb) : 0;
        } else {
            for (int j = 0; j < count; ++j) {
                br.refill();
                uint64_t u;
                const int z = br.acc ? __builtin_clzll(br.acc) : 64;
                if (z + 1 + k <= br.cnt) {  // the whole code -- z zeros, the one, k low bits -- lies in the window: one count, two shifts
                    const uint64_t rest = br.acc << z << 1;
                    u = ((uint64_t)z << k) | (k ? rest >> (64 - k) : 0);
                    br.acc = k ? rest << k : rest;
                    br.cnt -= z + 1 + k;
                } else {  // a long run of zeros, or the end of the stream
                    const uint64_t q = br.unary();
                    u = (q << k) | (k ? br.bits(k) : 0);
                }
                if (u >> 32) return false;  // a residual must fit 32 bits signed (RFC 9639 section 9.2.7.3)
                s[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
            }
        }
        if (br.bad) return false;
    }
    return i == blocksize;
}
bool read_residual(BitReader& outer, int64_t* s, int blocksize, int order) {
    BitReader br = outer;  // a local whose address never escapes: the window, its fill and the byte position live in registers over the
    const bool ok = read_residual_body(br, s, blocksize, order);  // Rice loop (through the reference every store to s[] forces a reload)
    outer = br;
    return ok;
}

// s[i] += (sum_j coef[j] * s[i - 1 - j]) >> shift for i = order .. n - 1, the order a compile-time constant for the common ones
template <int ORDER>
void lpc_restore_fixed_order(int64_t* s, int n, const int64_t* coef, int shift) {
    int64_t c[ORDER];
    for (int j = 0; j < ORDER; ++j) c[j] = coef[j];
    for (int i = ORDER; i < n; ++i) {
        int64_t acc = 0;
#pragma GCC unroll 32
        for (int j = 0; j < ORDER; ++j) acc = wadd(acc, wmul(c[j], s[i - 1 - j]));
        s[i] = wadd(s[i], acc >> shift);  // arithmetic shift (floor), as the format prescribes
    }
}
void lpc_restore(int64_t* s, int n, const int64_t* coef, int order, int shift) {
    switch (order) {
#define LOCO_LPC_CASE(o_) case o_: lpc_restore_fixed_order<o_>(s, n, coef, shift); return;
        LOCO_LPC_CASE(1) LOCO_LPC_CASE(2) LOCO_LPC_CASE(3) LOCO_LPC_CASE(4) LOCO_LPC_CASE(5) LOCO_LPC_CASE(6) LOCO_LPC_CASE(7) LOCO_LPC_CASE(8)
        LOCO_LPC_CASE(9) LOCO_LPC_CASE(10) LOCO_LPC_CASE(11) LOCO_LPC_CASE(12)
#undef LOCO_LPC_CASE
        default: break;
    }
    for (int i = order; i < n; ++i) {
        int64_t acc = 0;
        for (int j = 0; j < order; ++j) acc = wadd(acc, wmul(coef[j], s[i - 1 - j]));
        s[i] = wadd(s[i], acc >> shift);
    }
}

bool read_subframe(BitReader& br, int64_t* s, int blocksize, int bps) {
    if (br.bit()) return false;  // padding bit must be 0
    const int type = (int)br.bits(6);
    int wasted = 0;
    if (br.bit()) {
        const uint32_t w = br.unary();
        if (w >= (uint32_t)bps - 1) return false;  // at least one bit of the sample must remain
        wasted = (int)w + 1;
    }
    bps -= wasted;
    if (bps < 1 || br.bad) return false;
    if (type == 0) {  // CONSTANT
        const int64_t v = br.sbits(bps);
        for (int i = 0; i < blocksize; ++i) s[i] = v;
    } else if (type == 1) {  // VERBATIM
        for (int i = 0; i < blocksize; ++i) s[i] = br.sbits(bps);
    } else if (type >= 8 && type <= 12) {  // FIXED, order type - 8
        const int order = type - 8;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.sbits(bps);
        if (!read_residual(br, s, blocksize, order)) return false;
        for (int i = order; i < blocksize; ++i) {
            int64_t pred = 0;
            switch (order) {
                case 1: pred = s[i - 1]; break;
                case 2: pred = wsub(wmul(2, s[i - 1]), s[i - 2]); break;
                case 3: pred = wadd(wmul(3, wsub(s[i - 1], s[i - 2])), s[i - 3]); break;
                case 4: pred = wsub(wadd(wmul(4, wadd(s[i - 1], s[i - 3])), wmul(-6, s[i - 2])), s[i - 4]); break;
                default: break;
            }
            s[i] = wadd(s[i], pred);
        }
    } else if (type >= 32) {  // LPC, order (type & 31) + 1
        const int order = (type & 31) + 1;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.sbits(bps);
        const int prec = (int)br.bits(4) + 1;
        if (prec == 16) return false;
        const int shift = (int)br.sbits(5);
        if (shift < 0) return false;
        int64_t coef[32];
        for (int i = 0; i < order; ++i) coef[i] = br.sbits(prec);
        if (!read_residual(br, s, blocksize, order)) return false;
        lpc_restore(s, blocksize, coef, order, shift);
    } else {
        return false;  // reserved subframe type
    }
    if (wasted)
        for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
    return !br.bad;
}

// Decodes the whole stream; `sink(frame_samples, blocksize, channels)` receives every block's channel-major samples.
template <class Sink>
int decode_stream(const uint8_t* d, size_t n, const StreamInfo& si, bool verify_md5, Sink&& sink, int64_t* n_out) {
    static const int kBlockSizes[16] = {0, 192, 576, 1152, 2304, 4608, -8, -16, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
    static const int kSampleBits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    std::vector<int64_t> buf;
    std::vector<uint8_t> pcm;
    Md5 md5;
    int64_t decoded = 0;
    size_t o = si.audio_offset;
    const int bytes_ps = (si.bps + 7) / 8;
    while (o + 2 <= n) {
        if (si.total > 0 && decoded >= si.total) break;
        if (si.total == 0 && o + 3 <= n && memcmp(d + o, "TAG", 3) == 0) break;  // an ID3v1 tag behind a stream of unknown length
        if (!(d[o] == 0xff && (d[o + 1] & 0xfe) == 0xf8)) return ffail(LOCO_E_INVALID, "FLAC: lost frame synchronisation at byte " + std::to_string(o));
        BitReader br(d + o, n - o);
        br.bits(15);
        br.bit();  // blocking strategy: only tells whether the coded number counts frames or samples
        const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4), ch_code = (int)br.bits(4), ss_code = (int)br.bits(3);
        if (br.bit()) return ffail(LOCO_E_INVALID, "FLAC: reserved bit set in a frame header");
        {   // UTF-8-like coded frame / sample number: leading byte tells the length
            const uint32_t b0 = (uint32_t)br.bits(8);
            int extra = 0;
            if (b0 & 0x80) {
                uint32_t m = 0x40;
                while (b0 & m) { ++extra; m >>= 1; }
                if (extra < 1 || extra > 6) return ffail(LOCO_E_INVALID, "FLAC: bad coded number in a frame header");
            }
            for (int i = 0; i < extra; ++i)
                if (((uint32_t)br.bits(8) & 0xc0) != 0x80) return ffail(LOCO_E_INVALID, "FLAC: bad coded number in a frame header");
        }
        int blocksize = kBlockSizes[bs_code];
        if (bs_code == 0) return ffail(LOCO_E_INVALID, "FLAC: reserved block size code");
        if (blocksize == -8) blocksize = (int)br.bits(8) + 1;
        else if (blocksize == -16) blocksize = (int)br.bits(16) + 1;
        if (sr_code == 12) br.bits(8);
        else if (sr_code == 13 || sr_code == 14) br.bits(16);
        else if (sr_code == 15) return ffail(LOCO_E_INVALID, "FLAC: invalid sample rate code");
        if (br.bad) return ffail(LOCO_E_INVALID, "FLAC: truncated frame header");
        const size_t hdr_bytes = br.pos() / 8;
        const uint8_t want8 = (uint8_t)br.bits(8);
        if (crc8(d + o, hdr_bytes) != want8) return ffail(LOCO_E_INVALID, "FLAC: frame header CRC-8 mismatch at byte " + std::to_string(o));
        int bps = ss_code == 0 ? si.bps : kSampleBits[ss_code];
        if (bps < 0) return ffail(LOCO_E_INVALID, "FLAC: reserved sample size code");
        int channels;
        if (ch_code < 8) channels = ch_code + 1;
        else if (ch_code <= 10) channels = 2;
        else return ffail(LOCO_E_INVALID, "FLAC: reserved channel assignment");
        if (channels != si.channels || bps != si.bps) return ffail(LOCO_E_INVALID, "FLAC: a frame disagrees with STREAMINFO on channels / bits per sample");
        buf.resize((size_t)channels * blocksize);
        for (int c = 0; c < channels; ++c) {
            const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
            if (!read_subframe(br, buf.data() + (size_t)c * blocksize, blocksize, bps + (side ? 1 : 0)))
                return ffail(LOCO_E_INVALID, std::string(br.bad ? "FLAC: truncated frame at byte " : "FLAC: malformed subframe in the frame at byte ") + std::to_string(o));
        }
        br.align();
        const size_t body = br.pos() / 8;
        const uint16_t want16 = (uint16_t)br.bits(16);
        if (br.bad) return ffail(LOCO_E_INVALID, "FLAC: truncated frame");
        if (crc16(d + o, body) != want16) return ffail(LOCO_E_INVALID, "FLAC: frame CRC-16 mismatch at byte " + std::to_string(o));
        int64_t* c0 = buf.data();
        int64_t* c1 = buf.data() + blocksize;
        if (ch_code == 8) {
            for (int i = 0; i < blocksize; ++i) c1[i] = wsub(c0[i], c1[i]);
        } else if (ch_code == 9) {
            for (int i = 0; i < blocksize; ++i) c0[i] = wadd(c0[i], c1[i]);
        } else if (ch_code == 10) {
            for (int i = 0; i < blocksize; ++i) {
                const int64_t side = c1[i], mid = (int64_t)(((uint64_t)c0[i] << 1) | (uint64_t)(side & 1));
                c0[i] = wadd(mid, side) >> 1;
                c1[i] = wsub(mid, side) >> 1;
            }
        }
        int take = blocksize;
        if (si.total > 0 && decoded + take > si.total) take = (int)(si.total - decoded);
        if (verify_md5) {
            pcm.resize((size_t)take * channels * bytes_ps);
            size_t w = 0;
            if (bytes_ps == 2 && channels == 1) {  // the corpus case (16-bit mono): little-endian int16, two bytes per sample
                for (int i = 0; i < take; ++i) {
                    const uint64_t v = (uint64_t)buf[i];
                    pcm[w] = (uint8_t)v;
                    pcm[w + 1] = (uint8_t)(v >> 8);
                    w += 2;
                }
            } else {
                for (int i = 0; i < take; ++i)
                    for (int c = 0; c < channels; ++c) {
                        const int64_t v = buf[(size_t)c * blocksize + i];
                        for (int b = 0; b < bytes_ps; ++b) pcm[w++] = (uint8_t)((uint64_t)v >> (8 * b));
                    }
            }
            md5.update(pcm.data(), pcm.size());
        }
        sink(buf.data(), blocksize, take, channels);
        decoded += take;
        o += body + 2;
    }
    if (si.total > 0 && decoded != si.total)
        return ffail(LOCO_E_INVALID, "FLAC: stream ends after " + std::to_string(decoded) + " of " + std::to_string(si.total) + " samples");
    if (verify_md5) {
        bool unset = true;
        for (int i = 0; i < 16; ++i) unset = unset && si.md5[i] == 0;
        if (!unset) {
            uint8_t got[16];
            md5.finish(got);
            if (memcmp(got, si.md5, 16) != 0) return ffail(LOCO_E_INVALID, "FLAC: MD5 of the decoded samples differs from the signature in STREAMINFO");
        }
    }
    if (n_out) *n_out = decoded;
    return LOCO_OK;
}

}  // namespace

extern "C" {

const char* loco_flac_last_error(void) { return g_flac_err.c_str(); }

int loco_flac_info(const void* data, size_t nbytes, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample, int64_t* total_samples) {
    if (!data) return ffail(LOCO_E_INVALID, "loco_flac_info: null argument");
    StreamInfo si;
    const int rc = parse_header(reinterpret_cast<const uint8_t*>(data), nbytes, si);
    if (rc) return rc;
    if (sample_rate) *sample_rate = si.rate;
    if (channels) *channels = si.channels;
    if (bits_per_sample) *bits_per_sample = si.bps;
    if (total_samples) *total_samples = si.total;
    return LOCO_OK;
}

int loco_flac_decode(const void* data, size_t nbytes, float* mono, int32_t* pcm, int64_t capacity, int64_t* n_samples, int32_t verify_md5) {
    if (!data || (!mono && !pcm)) return ffail(LOCO_E_INVALID, "loco_flac_decode: null argument");
    StreamInfo si;
    int rc = parse_header(reinterpret_cast<const uint8_t*>(data), nbytes, si);
    if (rc) return rc;
    const float scale = 1.0f / (float)(1ull << (si.bps - 1));
    int64_t at = 0;
    bool overflow = false;
    auto sink = [&](const int64_t* s, int blocksize, int take, int channels) {
        if (at + take > capacity) { overflow = true; return; }
        if (!pcm && channels == 1) {  // the corpus case: one channel, float output only
            for (int i = 0; i < take; ++i) mono[at + i] = (float)s[i] * scale;
            at += take;
            return;
        }
        for (int i = 0; i < take; ++i) {
            if (pcm)
                for (int c = 0; c < channels; ++c) pcm[(at + i) * channels + c] = (int32_t)s[(size_t)c * blocksize + i];
            if (mono) {
                // soundfile's float32 read is sample * 2^-(bits-1) per channel (exact), the reference's mono mix the mean of those
                float acc = 0.f;
                for (int c = 0; c < channels; ++c) acc += (float)s[(size_t)c * blocksize + i] * scale;
                mono[at + i] = channels == 1 ? acc : acc / (float)channels;
            }
        }
        at += take;
    };
    int64_t n = 0;
    rc = decode_stream(reinterpret_cast<const uint8_t*>(data), nbytes, si, verify_md5 != 0, sink, &n);
    if (rc) return rc;
    if (overflow) return ffail(LOCO_E_WORKSPACE, "loco_flac_decode: output capacity " + std::to_string(capacity) + " < " + std::to_string(n) + " samples");
    if (n_samples) *n_samples = n;
    return LOCO_OK;
}

}  // extern "C"

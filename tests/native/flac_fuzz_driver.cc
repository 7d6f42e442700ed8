// TEST INFRASTRUCTURE (tests/test_flac_sanitized.py): runs every stream of a corpus file through the FLAC entry points of
// include/loco_asr.h.  Built together with loco-asr_amd/csrc/flac_decode.hip by g++ with -fsanitize=address,undefined
// -fno-sanitize-recover=all: any out-of-bounds access, signed overflow or bad shift on a damaged stream aborts the process.
// Corpus format: repeated [uint32 little-endian length][bytes].  Prints "streams=<n> decoded=<n> refused=<n>".
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/loco_asr.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    long streams = 0, decoded = 0, refused = 0;
    std::vector<uint8_t> buf;
    for (;;) {
        uint32_t len;
        if (fread(&len, 4, 1, f) != 1) break;
        // an exact-size heap copy: reads past the end of the stream are ASan errors, not reads of the next test case
        uint8_t* data = new uint8_t[len ? len : 1];
        if (len && fread(data, 1, len, f) != len) return 2;
        ++streams;
        int32_t sr = 0, ch = 0, bits = 0;
        int64_t total = 0;
        if (loco_flac_info(data, len, &sr, &ch, &bits, &total) != 0) {
            if (!*loco_flac_last_error()) return 3;  // a refusal must say why
            ++refused;
            delete[] data;
            continue;
        }
        int64_t cap = total > 0 ? total : 1 << 20;
        if (cap > (1 << 22)) cap = 1 << 22;  // a damaged STREAMINFO may claim 2^36 samples: the decoder must then report the capacity, not write
        bool ok = false;
        for (int pass = 0; pass < 3; ++pass) {
            std::vector<float> mono(pass != 1 ? (size_t)cap : 0);
            std::vector<int32_t> pcm(pass != 0 ? (size_t)cap * (size_t)ch : 0);
            int64_t n = -1;
            const int rc = loco_flac_decode(data, len, pass != 1 ? mono.data() : nullptr, pass != 0 ? pcm.data() : nullptr, cap, &n, pass == 2 ? 0 : 1);
            if (rc == 0) {
                if (n < 0 || n > cap) return 4;
                ok = true;
            } else if (!*loco_flac_last_error()) {
                return 3;
            }
        }
        // a capacity that is too small is an error code, never a write
        {
            std::vector<float> tiny(7);
            int64_t n = 0;
            const int rc = loco_flac_decode(data, len, tiny.data(), nullptr, 7, &n, 0);
            if (rc == 0 && n > 7) return 4;
        }
        ok ? ++decoded : ++refused;
        delete[] data;
    }
    fclose(f);
    printf("streams=%ld decoded=%ld refused=%ld\n", streams, decoded, refused);
    return 0;
}

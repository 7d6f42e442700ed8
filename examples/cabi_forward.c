/* The drop-in boundary without Python or torch: a plain C host drives libloco_asr.so through include/loco_asr.h.
 *
 *   gcc examples/cabi_forward.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lloco-asr_amd -lloco_asr \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/loco-asr_amd -Wl,-rpath,/opt/rocm/lib -o cabi_forward
 *   ./cabi_forward weights.bin manifest.txt wave.f32 B L out.f32
 *
 * manifest.txt: one line per tensor "<hf key> <ndim> <d0> [<d1> ...]" in the order the fp32 data appear in weights.bin
 * (keys exactly as `prenet.*` / `wrapped_encoder.*` of the reference's two state dicts, include/loco_asr.h).
 * wave.f32: B*L fp32 samples; out.f32 receives B*T*768 fp32 embeddings.  tests/test_gpu_cabi_c.py builds and runs this
 * and compares the file with what the Python wrapper returns (bit for bit: it is the same library call).  The second part shows
 * loco_forward_async: three forwards of the one handle in flight on three streams, each with its own workspace and status block;
 * the third loco_forward_packed: the same batch twice plus its first clip as a batch of its own, one launch sequence.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "loco_asr.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_LOCO(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s: [%d] %s\n", #x, rc_, loco_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 7) { fprintf(stderr, "usage: %s weights.bin manifest.txt wave.f32 B L out.f32\n", argv[0]); return 1; }
    const int B = atoi(argv[4]);
    const long L = atol(argv[5]);
    if (loco_abi_version() != 1) { fprintf(stderr, "unexpected ABI version %d\n", loco_abi_version()); return 1; }

    loco_config cfg;
    loco_default_config(&cfg);
    loco_encoder* enc = loco_create(&cfg);
    if (!enc) { fprintf(stderr, "loco_create: %s\n", loco_last_error()); return 3; }

    /* ---- weights: host buffers are fine, the library copies them to the device */
    FILE* fw = fopen(argv[1], "rb");
    FILE* fm = fopen(argv[2], "r");
    if (!fw || !fm) { perror("weights/manifest"); return 1; }
    char key[256];
    int ndim;
    while (fscanf(fm, "%255s %d", key, &ndim) == 2) {
        int64_t shape[4];
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) { long d; if (fscanf(fm, "%ld", &d) != 1) return 1; shape[i] = d; n *= (size_t)d; }
        float* buf = (float*)malloc(n * sizeof(float));
        if (!buf || fread(buf, sizeof(float), n, fw) != n) { fprintf(stderr, "short read for %s\n", key); return 1; }
        CHECK_LOCO(loco_set_weight(enc, key, buf, shape, ndim));
        free(buf);
    }
    fclose(fw);
    fclose(fm);
    char missing[512];
    if (loco_missing_weights(enc, missing, sizeof missing)) { fprintf(stderr, "missing weights: %s\n", missing); return 3; }
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_LOCO(loco_finalize_weights(enc, stream));

    /* ---- input, output, workspace: plain device pointers */
    const long T = loco_output_frames(L);
    const size_t nin = (size_t)B * L, nout = (size_t)B * T * 768;
    float* hwav = (float*)malloc(nin * sizeof(float));
    FILE* fx = fopen(argv[3], "rb");
    if (!fx || fread(hwav, sizeof(float), nin, fx) != nin) { fprintf(stderr, "cannot read %s\n", argv[3]); return 1; }
    fclose(fx);
    float *dwav, *dout;
    void* ws;
    const size_t wsb = loco_workspace_bytes(enc, B, L);
    CHECK_HIP(hipMalloc((void**)&dwav, nin * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dout, nout * sizeof(float)));
    CHECK_HIP(hipMalloc(&ws, wsb));
    CHECK_HIP(hipMemcpyAsync(dwav, hwav, nin * sizeof(float), hipMemcpyHostToDevice, stream));
    /* loco_forward is the asynchronous entry point; loco_forward_checked = forward + stream synchronisation + the numeric-range
     * status of precision mode f16x3, with a second pass on the exact-fp32 kernels should this input leave the fp16 planes' range */
    int32_t used_fp32 = 0;
    CHECK_LOCO(loco_forward_checked(enc, dwav, NULL, B, L, dout, NULL, NULL, ws, wsb, stream, &used_fp32));
    float* hout = (float*)malloc(nout * sizeof(float));
    CHECK_HIP(hipMemcpyAsync(hout, dout, nout * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    FILE* fo = fopen(argv[6], "wb");
    if (!fo || fwrite(hout, sizeof(float), nout, fo) != nout) { perror("out"); return 1; }
    fclose(fo);
    printf("encoded %d x %ld samples -> [%d, %ld, 768], workspace %.1f MB%s\n", B, L, B, T, wsb / 1e6,
           used_fp32 ? " (re-run on the exact-fp32 kernels: activation range)" : "");

    /* ---- several forwards of the same handle in flight (the reference's batch-of-two loop at GPU speed): each forward owns a
     * stream, a workspace and a status block; loco_forward_async keeps nothing of a forward in the handle.  Here the same batch is
     * encoded NF times concurrently and every copy must equal the one above bit for bit. */
    enum { NF = 3 };
    hipStream_t st[NF];
    void* wsk[NF];
    float* outk[NF];
    void* status[NF];
    for (int k = 0; k < NF; ++k) {
        CHECK_HIP(hipStreamCreate(&st[k]));
        CHECK_HIP(hipMalloc(&wsk[k], wsb));
        CHECK_HIP(hipMalloc((void**)&outk[k], nout * sizeof(float)));
        CHECK_HIP(hipHostMalloc(&status[k], loco_status_bytes(), hipHostMallocDefault)); /* pinned: the status copy stays asynchronous */
    }
    for (int k = 0; k < NF; ++k)  /* enqueue all of them before waiting for any */
        CHECK_LOCO(loco_forward_async(enc, -1, dwav, NULL, B, L, outk[k], NULL, NULL, wsk[k], wsb, st[k], status[k]));
    float* hk = (float*)malloc(nout * sizeof(float));
    int same = 0;
    for (int k = 0; k < NF; ++k) {
        CHECK_HIP(hipStreamSynchronize(st[k]));
        CHECK_LOCO(loco_status_check(status[k], NULL, 0)); /* LOCO_E_RANGE here = re-run THIS batch with precision 0 */
        CHECK_HIP(hipMemcpy(hk, outk[k], nout * sizeof(float), hipMemcpyDeviceToHost));
        same += memcmp(hk, hout, nout * sizeof(float)) == 0;
    }
    printf("%d forwards in flight on %d streams: %d of %d bit-identical to the single forward\n", NF, NF, same, NF);

    /* ---- several reference batches as ONE launch sequence (loco_forward_packed): the batch above twice, and between the two its
     * first clip cut to half its length as a batch of its own.  Every clip carries the padded length of ITS OWN batch (pad_len) and
     * its number of samples (valid_len, instead of a mask); rows of the two full copies must equal the single forward up to the fp32
     * summation order of the GEMMs (include/loco_asr.h), and the short batch must not have disturbed them. */
    const int BP = 2 * B + 1;
    const long Ls = (L / 2) & ~7L, Ts = loco_output_frames(Ls);
    int64_t* pad_len = (int64_t*)malloc(BP * sizeof(int64_t));
    int64_t* valid_len = (int64_t*)malloc(BP * sizeof(int64_t));
    float* hpack = (float*)calloc((size_t)BP * L, sizeof(float));
    for (int b = 0; b < BP; ++b) {
        const int src = b < B ? b : (b == B ? 0 : b - B - 1);
        pad_len[b] = valid_len[b] = (b == B) ? Ls : L;
        memcpy(hpack + (size_t)b * L, hwav + (size_t)src * L, (size_t)pad_len[b] * sizeof(float));
    }
    float *dpack, *dpout;
    void* wsp;
    const size_t wspb = loco_workspace_bytes(enc, BP, L);
    CHECK_HIP(hipMalloc((void**)&dpack, (size_t)BP * L * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dpout, (size_t)BP * T * 768 * sizeof(float)));
    CHECK_HIP(hipMalloc(&wsp, wspb));
    CHECK_HIP(hipMemcpyAsync(dpack, hpack, (size_t)BP * L * sizeof(float), hipMemcpyHostToDevice, st[0]));
    CHECK_LOCO(loco_forward_packed(enc, -1, dpack, NULL, valid_len, BP, L, pad_len, dpout, NULL, NULL, wsp, wspb, st[0], status[0]));
    CHECK_HIP(hipStreamSynchronize(st[0]));
    CHECK_LOCO(loco_status_check(status[0], NULL, 0));
    float* hp = (float*)malloc((size_t)BP * T * 768 * sizeof(float));
    CHECK_HIP(hipMemcpy(hp, dpout, (size_t)BP * T * 768 * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int copy = 0; copy < 2; ++copy) {
        const float* got = hp + (size_t)(copy ? B + 1 : 0) * T * 768;
        double num = 0.0, den = 0.0;
        for (size_t i = 0; i < nout; ++i) { const double d = (double)got[i] - hout[i]; num += d * d; den += (double)hout[i] * hout[i]; }
        const double rel = den > 0 ? num / den : 1.0;
        if (rel > worst) worst = rel;
    }
    double short_norm = 0.0;
    for (size_t i = 0; i < (size_t)Ts * 768; ++i) short_norm += (double)hp[(size_t)B * T * 768 + i] * hp[(size_t)B * T * 768 + i];
    printf("packed forward of %d clips (batches of %d, 1, %d): worst squared relative L2 of the two full batches against the single forward %.3e; "
           "the short batch has %ld frames, norm^2 %.4e\n", BP, B, B, worst, Ts, short_norm);
    loco_destroy(enc);
    return (same == NF && worst < 25e-12 && short_norm > 0.0) ? 0 : 4;
}

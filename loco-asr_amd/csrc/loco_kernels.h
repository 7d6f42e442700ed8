// Internal launcher declarations shared by the kernel translation units and the C-ABI (loco_api.hip).
// gfx950 only: 64-lane wavefronts, fp32-input MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace loco {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHidden = 768;
constexpr int kHeads = 12;
constexpr int kHeadDim = 64;
constexpr int kFfn = 3072;
constexpr int kConvDim = 512;
constexpr int kPosK = 128;
constexpr int kPosGroups = 16;
constexpr int kPosCg = 48;  // channels per group
constexpr int kRelMax = 160;
constexpr int kRelN = 320;
constexpr int kQkv = 3 * kHidden;

// Workspace of the split-K path: ks * tiles_m * tiles_n <= 256 tiles of at most 256 x 128 partial sums.
constexpr size_t kSplitKBytes = (size_t)256 * 256 * 128 * sizeof(float);
constexpr int kSplitKMaxM = 8192;

enum Epilogue { kEpiNone = 0, kEpiGelu = 1, kEpiResidual = 2, kEpiQkvScatter = 3, kEpiPosConv = 4 };

struct GemmArgs {
    const float* A;
    const float* W;
    const float* bias;  // may be null
    const float* R;     // residual (kEpiResidual), same batch strides as C
    float* C;
    int M, N, K;
    long lda, ldw, ldc, ldr;
    int nb1, nb2;
    long sA1, sA2, sC1, sC2;
    int epilogue;
};

// split-precision GEMM operands: fp16 hi/lo planes with the strides of the fp32 tensor they stand for
struct GemmSplitArgs {
    const _Float16* Ahi;
    const _Float16* Alo;
    const _Float16* Whi;
    const _Float16* Wlo;
    const float* bias;  // may be null
    const float* R;     // fp32 residual (kEpiResidual)
    float* C;           // fp32 output, or null when Chi/Clo are given
    _Float16* Chi;      // split output (the next GEMM's A operand), or null
    _Float16* Clo;
    int M, N, K;
    long lda, ldw, ldc, ldr;
    int nb1, nb2;
    long sA1, sA2, sC1, sC2;
    int epilogue;
    // kEpiQkvScatter (fused q|k|v projection feeding the split-precision attention): columns [0,768) -> Chi/Clo planes
    // [M,768] (q), [768,1536) -> the k planes, [1536,2304) -> the v planes, each pair qkv_stride halves behind the previous one (attention
    // transposes V with its LDS read)
    // kEpiResidual: the residual may also be given as fp16 hi/lo planes (same element offsets and ldr as R); it is then hi + lo,
    // exact in fp32.  The encoder's residual stream lives only in that form between layers (LayerNorm writes no fp32 copy).
    const _Float16* Rhi = nullptr;
    const _Float16* Rlo = nullptr;
    // split-K for small problems (launch_gemm_split decides): fp32 partial sums [ks][M][N], >= kSplitKBytes when set
    float* splitk_ws = nullptr;
    long qkv_stride = 0;  // kEpiQkvScatter: k planes at Chi / Clo + qkv_stride, v planes at + 2 qkv_stride (halves)
    int T = 0;
    // kEpiPosConv (grouped positional conv as a GEMM over the group-major halo layout, see launch_group_major_split):
    // z1 = clip, z2 = group; C = R + GELU(acc + bias) + sin_table[pos(t)], pos = t+2 for t < frames[z1] else 1
    // The weight planes hold W * 2^k (k chosen per tensor at load time so that max|W| lands in [2^13, 2^14): fp16's 5-bit
    // exponent then serves the weights' own spread instead of their absolute level); the accumulator is multiplied by
    // out_scale = 2^-k before bias / activation -- exact, a power of two.
    float out_scale = 1.0f;
    int terms = 3;  // 3: A_hi W_hi + A_lo W_hi + A_hi W_lo;  2: the W_lo term dropped (precision mode "f16x2")
    // true while another stream of the same forward is launching kernels of its own (the two half-batch schedule): a partly filled
    // last round is then filled by that stream's workgroups, and the tile choice stops paying for whole rounds
    bool co_scheduled = false;
    // Convolution as a GEMM over overlapping rows (A row t = input rows stride*t .. stride*t + ktaps - 1, K = ktaps * C): with
    // ktaps > 1 the k axis is walked (64-channel block, tap slot, 32-channel half), tap slots in the order 0, 2, 1, and the weight
    // planes are stored in that order (launch_permute_conv_k).  An input row that is tap 2 of output t is tap 0 of output t+1
    // (stride 2): in a tap-major walk its two uses are 32 k-tiles apart, by when 1.5 MB per CU have gone through the XCD's
    // 4 MiB L2 and the row is fetched again (conv1: 1.5 x its 6.3 GB operand).  Here they are two k-tiles apart, and the two
    // 64-byte halves of every 128-byte line are consecutive k-tiles.
    int ktaps = 1;
    int kchan = 0;       // channels per tap (0: K / ktaps); set by the split-K path, whose slices see only part of K
    int kt_per_z2 = 0;   // split-K with ktaps > 1: slice z2 starts at k-tile z2 * kt_per_z2 of that walk (its A offset is not linear)
    // Range tracking: where the output is written as fp16 hi/lo planes, max|x| of what was written is folded into
    // range_slot[0..7] (see range_commit); null = not tracked.
    float* range_slot = nullptr;
    long sW2 = 0;                   // weight stride per z2 (0 = shared weights)
    // z1 may itself be a pair (outer, inner), z1 = outer * z1_inner + inner: sA1 then strides the outer part, sA1i / sW1i the inner
    // one (C still strides the combined z1).  Used by the split-K path of the grouped positional conv, whose batch dimensions
    // (clip, group) are both taken: inner = K slice.
    int z1_inner = 1;
    long sA1i = 0, sW1i = 0;
    long sBias2 = 0;                // bias stride per z2
    const float* sin_table = nullptr;
    const int32_t* frames = nullptr;
};
// x [B,T,768] fp32 -> fp16 hi/lo planes in group-major layout [B][16][T+128][48] with 64 zero frames before and after:
// output frame t of group g then reads the CONTIGUOUS run rows t .. t+127 (128 taps x 48 channels = K 6144, lda = 48)
// rows_clip (device, [B], may be null): packed forward -- rows t >= rows_clip[b] of clip b read as zeros (the conv's zero padding
// starts where the clip's own reference batch ends)
hipError_t launch_group_major_split(const float* x, void* hi, void* lo, int B, int T, hipStream_t s, float* range_slot = nullptr,
                                    const int32_t* rows_clip = nullptr);
// folded positional-conv weight [g][tap][o][i] -> [g][o][tap*48 + i] (the GEMM's W, ldw = 6144)
hipError_t launch_pos_w_for_gemm(const float* wf, float* out, hipStream_t s);
hipError_t launch_attention_f16x3(const _Float16* qhi, const _Float16* qlo, const _Float16* khi, const _Float16* klo,
                                  const _Float16* vhi, const _Float16* vlo, const float* qp, const int32_t* frames,
                                  _Float16* ctx_hi, _Float16* ctx_lo, float* ctx, int B, int T, hipStream_t s,
                                  const _Float16* pe_hi = nullptr, const _Float16* pe_lo = nullptr, float pe_scale = 1.0f);
// pe_hi / pe_lo != nullptr: the relative-position table is COMPUTED by the kernel (Qp = q . pe_k^T * pe_scale, pe planes [320][64])
// into `qp`, which is then scratch of the launch ([B,12,T,320] fp32) instead of an input.
hipError_t launch_gemm_split(const GemmSplitArgs& a, hipStream_t s);
void reload_gemm_knobs();  // re-read the LOCO_GEMM_* A/B knobs from the environment (gemm_f16x3.hip)
void reload_attention_knobs();  // ... and LOCO_ATTN_LONG (attention_f16x3.hip)
// hi/lo planes of x * scale (scale a power of two)
hipError_t launch_split_f16(const float* x, void* hi, void* lo, long n, hipStream_t s, float scale = 1.0f);
// max |x| over n elements -> *out (one float, device); out must be zeroed by the caller
hipError_t launch_absmax(const float* x, long n, float* out, hipStream_t s);

// diagnostics: count the non-finite elements of a buffer (out[0] += count, out[1] = min(first index)); LOCO_DEBUG_NONFINITE=1
hipError_t launch_count_nonfinite(const void* x, long n, bool half, unsigned long long* out, hipStream_t s);

// ---- range tracking of tensors stored as fp16 hi/lo planes (precision mode f16x3) ------------------------------------------
// hi = fp16(x) is inf from |x| >= 65520 on, and the pair (hi, lo) keeps 22 significant bits only while lo = fp16(x - hi) is a
// normal fp16 number, i.e. for |x| >~ 2^-3; below that the ABSOLUTE error levels off at 2^-25.  A tensor is therefore
// represented to fp32 class relative to its own scale as long as its largest element sits well inside (2^-6, 65504) -- the
// range include/loco_asr.h guarantees.
//   * Tensors whose range follows from the weights alone are checked on the host, at no run-time cost: LayerNorm outputs
//     (|y| <= sqrt(D) max|gamma| + max|beta|), the attention context (a convex combination of V rows) -- loco_api.hip,
//     static_range_check.
//   * The unbounded ones -- conv0's GroupNorm + GELU output (its bound grows with sqrt(frames per clip): useless for 10-minute
//     clips), GELU outputs of conv layers 1-5 and of the feed-forward intermediate, q|k|v, the feature projection -- are tracked
//     where they are produced: the GEMM epilogue (and group_major_split) folds max|x| of what it
//     writes, taken on the fp32 value before conversion, into its stage's status word.  8 shards per stage (workgroup id mod
//     8) so that no single address takes every workgroup's atomic; the word is read EARLY (range_peek, before the epilogue's
//     arithmetic, so the load's latency is hidden) and a wave whose maximum is already covered skips the atomic -- the word
//     only grows, so a stale read costs at most a redundant atomic.  The wave maximum is taken with four DPP steps inside
//     the rows of 16 lanes and four v_readlane across them: no LDS traffic, no waits.
// The host reads the words after the forward (loco_forward_status) and decides (include/loco_asr.h).
constexpr int kRangeShards = 8;
constexpr int kRangeMaxStages = 96;
constexpr int kFiniteStage = kRangeMaxStages - 1;  // reserved: set to +inf by the forward's last LayerNorm when a row is not finite
__device__ __forceinline__ unsigned range_peek(const float* slot) {
    if (!slot) return 0xffffffffu;
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(slot) + (blockIdx.x & (kRangeShards - 1)), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
// max over the wave's 64 lanes of a NON-NEGATIVE value (lanes without a partner read 0); every lane must be active
__device__ __forceinline__ float wave_max_nonneg(float v) {
#define LOCO_DPP_MAX(ctrl) \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true)))
    LOCO_DPP_MAX(0xB1);   // quad_perm [1,0,3,2]: lane ^ 1
    LOCO_DPP_MAX(0x4E);   // quad_perm [2,3,0,1]: lane ^ 2
    LOCO_DPP_MAX(0x141);  // row_half_mirror: the other quad of each 8
    LOCO_DPP_MAX(0x140);  // row_mirror: the other half of each row of 16
#undef LOCO_DPP_MAX
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ void range_commit(float* slot, float amax, unsigned seen) {
    if (!slot) return;
    const unsigned bits = __float_as_uint(wave_max_nonneg(amax));  // non-negative floats order like their bit patterns; NaN never enters (fmaxf)
    if ((threadIdx.x & 63) == 0 && bits > seen)
        atomicMax(reinterpret_cast<unsigned*>(slot) + (blockIdx.x & (kRangeShards - 1)), bits);
}

// The same for kernels of MANY small workgroups (the split-K reductions: thousands of 256-thread blocks that all start at once
// and all peek a zero word): one atomic per workgroup instead of one per wave -- the wave maxima meet in LDS first.  Every thread
// of the block must call it.  (With one atomic per wave the reduction behind FFN1's split-K GEMM took 72 us at 498 rows, 50 of
// them in ~6 000 atomics on eight addresses: a quarter of a whole forward of the reference's 2 x 5 s batch.)
__device__ __forceinline__ void range_commit_block(float* slot, float amax, unsigned seen) {
    if (!slot) return;
    __shared__ float wmax[16];
    const float w = wave_max_nonneg(amax);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = wmax[0];
        for (unsigned i = 1; i < (blockDim.x + 63) / 64; ++i) m = fmaxf(m, wmax[i]);
        const unsigned bits = __float_as_uint(m);
        if (bits > seen) atomicMax(reinterpret_cast<unsigned*>(slot) + (blockIdx.x & (kRangeShards - 1)), bits);
    }
}

// exact GELU 0.5*x*(1+erf(x/sqrt2))

// Exact (erf) GELU, HF transformers/activations.py:83-89, without erff's two divergent branches.
//   erfc(t) = exp(-t q(t)) for t >= 0 with q(t) = -ln(erfc(t)) / t smooth and slowly varying (q(0) = 2/sqrt(pi)); with
//   s = |x|, t = s/sqrt(2) and the 1/sqrt(2) and log2(e) factors folded into the coefficients:  e = 2^(-s Q(s)) = erfc(s/sqrt 2),
//   GELU(x) = x (1 - e/2) for x >= 0,   x e/2 for x < 0        -- no cancellation in either branch.
// Q: degree-9 fit of q on t in [0, 4.1] (erfc(4.1) = 6.7e-9; |x| is clamped there, GELU is x resp. 0 to 2e-8 beyond).
// Against an fp64 GELU on [-9, 9]: relative error <= 2.2e-7 for x > 0 (torch's fp32 GELU: 3.7e-7), absolute error <= the
// rounding of x itself; relative L2 on N(0, 1.5) inputs 2.9e-8.  10 FMAs + one v_exp_f32 instead of ~45 instructions.
__device__ __forceinline__ float gelu_erf(float x) {
    constexpr float kSMax = 5.79827547f;  // 4.1 * sqrt(2)
    const float s = fminf(fabsf(x), kSMax);
    float q = 2.171883651e-08f;
    q = fmaf(q, s, -6.759613029e-07f);
    q = fmaf(q, s, 9.013814633e-06f);
    q = fmaf(q, s, -6.522983313e-05f);
    q = fmaf(q, s, 2.421164681e-04f);
    q = fmaf(q, s, 7.379760791e-05f);
    q = fmaf(q, s, -7.028903347e-03f);
    q = fmaf(q, s, 5.248807371e-02f);
    q = fmaf(q, s, 4.592096508e-01f);
    q = fmaf(q, s, 1.151104808e+00f);
    const float h = 0.5f * __builtin_amdgcn_exp2f(-(q * s));  // erfc(|x| / sqrt 2) / 2
    // the clamp of the negative branch is a compare + select, not fmaxf: fmaxf(NaN, c) = c would launder a NaN input into a finite
    // number (HF's GELU propagates it, and the forward's finite check must see it)
    return x >= 0.f ? x * (1.0f - h) : (x < -kSMax ? -kSMax : x) * h;
}

// fp16 hi/lo split of two value pairs (a0,a1) and (b0,b1): hi = fp16(x) (hipcc emits one v_cvt_pk_f16_f32 per pair),
// lo = fp16(x - hi) by one mixed-precision FMA per value (fp32 x, fp16 hi: the difference is exact in fp32, so lo is
// rounded once -- bit for bit what `lo = (_Float16)(x - (float)hi)` gives, at 6 instructions for 4 values instead of 16).
//  * the inputs are pinned first: left to itself hipcc folds the producing multiply into v_fma_mix*_f16 for lo but converts
//    hi from the fp32-rounded product (one fp16 ulp of hi on ties);
//  * hi stays compiler-visible code, so the first instruction that reads a value coming from an MFMA or from the
//    transcendental unit is one hipcc pads itself -- it does not look inside inline asm;
//  * inside the asm the two pairs are interleaved: each v_fma_mixhi (which merges into the register its v_fma_mixlo wrote
//    16 bits of) has an independent instruction in front of it.
__device__ __forceinline__ void split_f16_2pairs(float a0, float a1, float b0, float b1, unsigned& hia, unsigned& loa, unsigned& hib,
                                                 unsigned& lob) {
    typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
    const h2_ ha = {(_Float16)a0, (_Float16)a1}, hb = {(_Float16)b0, (_Float16)b1};
    hia = __builtin_bit_cast(unsigned, ha);
    hib = __builtin_bit_cast(unsigned, hb);
    asm("v_fma_mixlo_f16 %0, %2, 1.0, -%6 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %4, 1.0, -%7 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, 1.0, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, 1.0, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(loa), "=&v"(lob)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(hia), "v"(hib));
}

// The same on two values at once: the polynomial runs on v_pk_fma_f32 (two fp32 FMAs per instruction).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
    constexpr float kSMax = 5.79827547f;
    const f32x2_t s = {fminf(fabsf(x.x), kSMax), fminf(fabsf(x.y), kSMax)};
#define LOCO_Q2(c) (f32x2_t){(c), (c)}
    f32x2_t q = LOCO_Q2(2.171883651e-08f);
    q = __builtin_elementwise_fma(q, s, LOCO_Q2(-6.759613029e-07f));
    q = __builtin_elementwise_fma(q, s, LOCO_Q2(9.013814633e-06f));
    q = __builtin_elementwise_fma(q, s, LOCO_Q2(-6.522983313e-05f));
    q = __builtin_elementwise_fma(q, s, LOCO_Q2(2.421164681e-04f));
    q = __builtin_elementwise_fma(q, s, LOCO_Q2(7.379760791e-05f));
    q = __builtin_elementwise_fma(q, s, LOCO_Q2(-7.028903347e-03f));
    q = __builtin_elementwise_fma(q, s, LOCO_Q2(5.248807371e-02f));
    q = __builtin_elementwise_fma(q, s, LOCO_Q2(4.592096508e-01f));
    q = __builtin_elementwise_fma(q, s, LOCO_Q2(1.151104808e+00f));
#undef LOCO_Q2
    const f32x2_t u = q * s;
    const f32x2_t h = {0.5f * __builtin_amdgcn_exp2f(-u.x), 0.5f * __builtin_amdgcn_exp2f(-u.y)};
    f32x2_t r;
    r.x = x.x >= 0.f ? x.x * (1.0f - h.x) : (x.x < -kSMax ? -kSMax : x.x) * h.x;  // NaN-propagating clamp, see gelu_erf
    r.y = x.y >= 0.f ? x.y * (1.0f - h.y) : (x.y < -kSMax ? -kSMax : x.y) * h.y;
    return r;
}

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, long rows, int dim, float eps,
                            hipStream_t s, void* yhi = nullptr, void* ylo = nullptr, float* nonfinite_slot = nullptr);

constexpr int kConv0Parts = 64;    // partial-moment blocks per clip
constexpr int kConv0Moments = 65;  // 10 first + 55 second moments
size_t conv0_scratch_bytes(int B);
hipError_t launch_conv0_gn_gelu(const float* wav, int B, long L, const float* w, const float* gn_w, const float* gn_b,
                                float* out, void* scratch, float eps, hipStream_t s, void* out_hi = nullptr,
                                void* out_lo = nullptr, float* range_slot = nullptr, const int32_t* t0_clip = nullptr);
// t0_clip (device, [B], may be null): packed forward -- GroupNorm statistics of clip b over its first t0_clip[b] conv frames only,
// frames beyond written as zeros
hipError_t launch_frame_counts(const int32_t* mask, int B, long L, int32_t* frames, hipStream_t s);
size_t normalize_scratch_bytes(int B);
hipError_t launch_normalize_waveform(const float* wav, const int32_t* mask, int B, long L, float pad, float* out, void* scratch,
                                     hipStream_t s);
hipError_t launch_token_counts(const int32_t* mask, int B, int T, int32_t* frames, hipStream_t s);
hipError_t launch_text_prenet(const int32_t* ids, const float* embed, int vocab, const float* alpha, const float* pe, int B, int T,
                              float* out, hipStream_t s);
hipError_t launch_text_pe_table(float* pe, int rows, hipStream_t s);
hipError_t launch_pos_conv(const float* h, const float* wf, const float* bias, const float* sin_table,
                           const int32_t* frames, float* out, int B, int T, hipStream_t s, const int32_t* rows_clip = nullptr);
hipError_t launch_attention(const float* qkv, const float* qp, const int32_t* frames, float* ctx, int B, int T,
                            hipStream_t s, void* ctx_hi = nullptr, void* ctx_lo = nullptr);

// weight preparation (run once in loco_finalize_weights)
hipError_t launch_relayout_conv_weight(const float* w, float* out, int N, int C, int k, hipStream_t s);  // [N,C,k]->[N,k*C]
hipError_t launch_fold_pos_conv(const float* g, const float* v, float* out, hipStream_t s);  // -> [16][128][48][48]
hipError_t launch_scale_copy(const float* src, float* dst, long n, float scale, hipStream_t s);
// conv weight [N][taps][C] (tap-major K) -> [N][C/64][tap slot][2][32] (the k order of GemmSplitArgs::ktaps; slots = taps 0, 2, 1)
hipError_t launch_permute_conv_k(const float* src, float* dst, int N, int taps, int C, hipStream_t s);
hipError_t launch_sinusoid_table(float* tab, int rows, hipStream_t s);

// sample-rate conversion (resample.hip)
int resample_design(int sr_in, int sr_out, int* L_out, int* M_out, int* K_out, float* taps);
hipError_t launch_resample(const float* x, int B, long n_in, long x_stride, const float* taps, int L, int M, int K, float* y, long n_out,
                           long y_stride, hipStream_t s);

inline long conv_out_len(long n, int k, int s) { return n < k ? 0 : (n - k) / s + 1; }

}  // namespace loco

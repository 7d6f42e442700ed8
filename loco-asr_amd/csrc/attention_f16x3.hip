// Self-attention core on the split-precision fp16 x3 MFMA (precision mode f16x3) -- same algorithm, layout tricks and
// relative-position handling as attention_f32.hip (flash-style online softmax, swapped product S^T = K Q^T so the
// query sits on the lane, per-row constant bias for clipped tiles, coalesced + transposed band for the diagonal),
// with both matrix products issued as three v_mfma_f32_32x32x16_f16 per fp32-class product:
//
//   S^T  += K_lo Q_hi^T + K_hi Q_lo^T + K_hi Q_hi^T          (q, k arrive as fp16 hi/lo planes from the QKV GEMM epilogue)
//   O^T  += V_lo^T P_hi + V_hi^T P_lo + V_hi^T P_hi           (P = exp2(..) is split in registers, V^T planes come
//                                                              TRANSPOSED [head*64+d][t] from the same epilogue)
//
// 48 MFMAs x 32 cycles per 64-key tile and wave instead of 128 x 64 cycles: 5.3x fewer matrix-pipe cycles; softmax
// statistics, the running (m, l) and the O accumulators stay fp32.
//
// P as the next MFMA's B operand without touching LDS: the C/D fragment of S^T holds, in lane-half h, register 8s+j,
// key 16s + 8(j>>2) + 4h + (j&3) of a 32-key sub-tile; the 32x32x16 B operand wants k = 8h + j from that lane half, so
// registers 8s..8s+7 ARE k-step s up to a permutation of k -- the same permutation is applied to the V^T operand, which
// therefore reads two 8-byte runs of 4 consecutive keys {16s+4h, 16s+8+4h} from a [d][key] LDS image.
//
// LDS per stage: K planes 64 keys x (64+8) halves (144-byte rows: conflict-free ds_read_b128), V^T planes 64 d x (64+4)
// halves (136-byte rows: conflict-free ds_read_b64); two stages + the per-wave bias transpose scratch = 78.5 KiB, two
// workgroups per CU.
#include "loco_kernels.h"

namespace loco {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int AX_BQ = 128, AX_BK = 64;
constexpr int AX_LDK = kHeadDim + 8;   // halves per K row
constexpr int AX_LDV = AX_BK + 4;      // halves per V^T row
constexpr int AX_KPL = AX_BK * AX_LDK; // halves per K plane
constexpr int AX_VPL = kHeadDim * AX_LDV;
constexpr int AX_STAGE = 2 * AX_KPL + 2 * AX_VPL;

template <bool OUT_SPLIT>
__global__ __launch_bounds__(256, 2) void attention_f16x3_kernel(const _Float16* __restrict__ qhi, const _Float16* __restrict__ qlo,
                                                                 const _Float16* __restrict__ khi, const _Float16* __restrict__ klo,
                                                                 const _Float16* __restrict__ vthi, const _Float16* __restrict__ vtlo,
                                                                 const float* __restrict__ qp, const int32_t* __restrict__ frames,
                                                                 _Float16* __restrict__ ctx_hi, _Float16* __restrict__ ctx_lo,
                                                                 float* __restrict__ ctx, int T, int Tp, int nqb) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * AX_STAGE];
    __shared__ float bias_stage[4][32 * 17];

    // XCD-aware work map: workgroups whose ids are congruent mod 8 share an XCD (and its private L2).  Each XCD is given a
    // contiguous run of (clip, head, query-block) items with the query block fastest, so the query blocks of one
    // (clip, head) -- which all stream the same K/V tiles -- run side by side on ONE L2 instead of eight.
    int qblk, head, b;
    {
        const int nblk = gridDim.x;
        const int q8 = nblk >> 3, r8 = nblk & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        qblk = w - (w / nqb) * nqb;
        const int rest = w / nqb;
        head = rest - (rest / kHeads) * kHeads;
        b = rest / kHeads;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int iw0 = qblk * AX_BQ + wave * 32;
    const int iq = iw0 + r;
    const int iqc = iq < T ? iq : T - 1;

    int nvalid = frames ? frames[b] : T;
    if (nvalid <= 0 || nvalid > T) nvalid = T;
    const int ntiles = (nvalid + AX_BK - 1) / AX_BK;
    constexpr float kLog2e = 1.4426950408889634f;

    // Q fragments (B operand of S^T): element j of k-step ks = Q[iq][16 ks + 8 h + j]
    h8 qh[4], ql[4];
    {
        const long qo = ((long)b * T + iqc) * kHidden + head * kHeadDim + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qh[ks] = *reinterpret_cast<const h8*>(qhi + qo + 16 * ks);
            ql[ks] = *reinterpret_cast<const h8*>(qlo + qo + 16 * ks);
        }
    }
    const float* qprow = qp + (((long)b * kHeads + head) * T + iqc) * kRelN;
    const float c_past = qprow[kRelN - 1];  // i - j >= 159
    const float c_future = qprow[0];        // i - j <= -160

    // staging: 16-byte piece f = tid + 256 u of a 64 x 128-byte plane tile -> row f/8, piece f%8 (8 halves)
    const int srow = tid >> 3, spc = (tid & 7) * 8;
    const long kbase = (long)b * T * kHidden + head * kHeadDim + spc;
    const long vbase = ((long)b * kHidden + head * kHeadDim) * Tp + spc;
    h8 sk[4], sv[4];  // [hi u0, hi u1, lo u0, lo u1]
#define AX_LOAD_TILE(t)                                                                    \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                        \
        int j_ = (t) * AX_BK + srow + 32 * u;                                              \
        j_ = j_ < T ? j_ : T - 1;                                                          \
        sk[u] = *reinterpret_cast<const h8*>(khi + kbase + (long)j_ * kHidden);            \
        sk[2 + u] = *reinterpret_cast<const h8*>(klo + kbase + (long)j_ * kHidden);        \
        const long vo_ = vbase + (long)(srow + 32 * u) * Tp + (long)(t) * AX_BK;           \
        sv[u] = *reinterpret_cast<const h8*>(vthi + vo_);                                  \
        sv[2 + u] = *reinterpret_cast<const h8*>(vtlo + vo_);                              \
    }
#define AX_STORE_TILE(st_)                                                                                       \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                              \
        _Float16* kb_ = lds + (st_) * AX_STAGE + (srow + 32 * u) * AX_LDK + spc;                                 \
        *reinterpret_cast<h8*>(kb_) = sk[u];                                                                     \
        *reinterpret_cast<h8*>(kb_ + AX_KPL) = sk[2 + u];                                                        \
        _Float16* vb_ = lds + (st_) * AX_STAGE + 2 * AX_KPL + (srow + 32 * u) * AX_LDV + spc;                    \
        *reinterpret_cast<h4*>(vb_) = __builtin_shufflevector(sv[u], sv[u], 0, 1, 2, 3);                         \
        *reinterpret_cast<h4*>(vb_ + 4) = __builtin_shufflevector(sv[u], sv[u], 4, 5, 6, 7);                     \
        *reinterpret_cast<h4*>(vb_ + AX_VPL) = __builtin_shufflevector(sv[2 + u], sv[2 + u], 0, 1, 2, 3);        \
        *reinterpret_cast<h4*>(vb_ + AX_VPL + 4) = __builtin_shufflevector(sv[2 + u], sv[2 + u], 4, 5, 6, 7);    \
    }

    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    AX_LOAD_TILE(0)
    AX_STORE_TILE(0)
    if (ntiles > 1) { AX_LOAD_TILE(1) }
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < ntiles; ++t) {
        const _Float16* kb = lds + cur * AX_STAGE;
        const _Float16* vb = kb + 2 * AX_KPL;
        if (t + 1 < ntiles) {
            AX_STORE_TILE(cur ^ 1)
            if (t + 2 < ntiles) { AX_LOAD_TILE(t + 2) }
        }
        const int j0 = t * AX_BK;

        // ---- S^T = K Q^T: 2 sub-tiles x 4 k-steps x 3 MFMAs
        f32x16 s[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[st][e] = 0.f;
            const _Float16* kr = kb + (st * 32 + r) * AX_LDK + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h8 kh = *reinterpret_cast<const h8*>(kr + 16 * ks);
                const h8 kl = *reinterpret_cast<const h8*>(kr + AX_KPL + 16 * ks);
                s[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s[st], 0, 0, 0);
                s[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s[st], 0, 0, 0);
                s[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s[st], 0, 0, 0);
            }
        }
        // scores stay in natural units; log2(e) is folded into the exponent FMA below
        // ---- relative-position bias + key mask (see attention_f32.hip)
        const int dmin = iw0 - (j0 + AX_BK - 1);
        const int dmax = iw0 + 31 - j0;
        float cb = 0.f;
        if (dmin >= kRelMax - 1) {
            cb = c_past;
        } else if (dmax <= -kRelMax) {
            cb = c_future;
        } else {
            float* sc = bias_stage[wave];
            const float* qpb = qp + ((long)b * kHeads + head) * T * kRelN;
            const int lj = lane & 15, lq = lane >> 4;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int j = j0 + 32 * st + 16 * half + lj;
                    float bv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = iw0 + 4 * u + lq;
                        int rel = i - j;
                        rel = rel < -kRelMax ? -kRelMax : (rel > kRelMax - 1 ? kRelMax - 1 : rel);
                        bv[u] = qpb[(long)(i < T ? i : T - 1) * kRelN + rel + kRelMax];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) sc[(4 * u + lq) * 17 + lj] = bv[u];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int e8 = 0; e8 < 8; ++e8) {
                        const int e = 8 * half + e8;
                        s[st][e] += sc[r * 17 + (e8 & 3) + 8 * (e8 >> 2) + 4 * h];
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (j0 + AX_BK > nvalid) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int j = j0 + 32 * st + (e & 3) + 8 * (e >> 2) + 4 * h;
                    s[st][e] = j < nvalid ? s[st][e] : -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[0][e]);
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[1][e]);
        {   // the other lane half holds the other 32 keys of this query: one permlane32 swap instead of an LDS permute
            const unsigned mu = __builtin_bit_cast(unsigned, mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
            mx = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1])) + cb;
        }

        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
        m_run = m_new;
        const float dsh = (cb - m_new) * kLog2e;
        f32x2 ps2 = {0.f, 0.f};
        const f32x2 k2 = {kLog2e, kLog2e}, d2 = {dsh, dsh};
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
        }

        // ---- O^T += V^T P^T: per sub-tile 2 k-steps; P split in registers, V^T runs of 4 keys from the [d][key] image
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // p = exp2(s log2e + dsh); hi = fp16(p) packed by one cvt_pk per pair, lo = fp16(p - hi) by one mixed-precision
                // FMA per element (fp32 p, fp16 hi: the difference is exact, so lo is rounded once, like the host split)
                u32x4 uh, ul;
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    const f32x2 sx = {s[st][8 * s2 + 2 * jp], s[st][8 * s2 + 2 * jp + 1]};
                    const f32x2 ax = __builtin_elementwise_fma(sx, k2, d2);
                    const f32x2 pv = {__builtin_amdgcn_exp2f(ax.x), __builtin_amdgcn_exp2f(ax.y)};
                    ps2 += pv;
                    unsigned hi, lo;
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(pv.x), "v"(pv.y));
                    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(pv.x), "v"(hi));
                    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(pv.y), "v"(hi));
                    uh[jp] = hi;
                    ul[jp] = lo;
                }
                const h8 ph = __builtin_bit_cast(h8, uh), pl = __builtin_bit_cast(h8, ul);
                const int kofs = 32 * st + 16 * s2 + 4 * h;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const _Float16* vr = vb + (32 * dt + r) * AX_LDV + kofs;
                    const h4 a0 = *reinterpret_cast<const h4*>(vr), a1 = *reinterpret_cast<const h4*>(vr + 8);
                    const h4 c0 = *reinterpret_cast<const h4*>(vr + AX_VPL), c1 = *reinterpret_cast<const h4*>(vr + AX_VPL + 8);
                    const h8 vh = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const h8 vl = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7);
                    if (dt == 0) {
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o0, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o0, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o0, 0, 0, 0);
                    } else {
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o1, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o1, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o1, 0, 0, 0);
                    }
                }
            }
        }
        l_run = l_run * alpha + (ps2.x + ps2.y);

        __syncthreads();
        cur ^= 1;
    }
#undef AX_LOAD_TILE
#undef AX_STORE_TILE

    // ---- normalise and store: o{0,1}[e] = O[iq][d = 32 dt + (e&3) + 8 (e>>2) + 4h]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (iq < T) {
        const long obase = ((long)b * T + iq) * kHidden + head * kHeadDim + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float a[4], c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = o0[4 * g4 + e] * inv;
                c[e] = o1[4 * g4 + e] * inv;
            }
            if (OUT_SPLIT) {
                h4 ah, al, ch, cl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    asm volatile("" : "+v"(a[e]), "+v"(c[e]));
                    ah[e] = (_Float16)a[e]; al[e] = (_Float16)(a[e] - (float)ah[e]);
                    ch[e] = (_Float16)c[e]; cl[e] = (_Float16)(c[e] - (float)ch[e]);
                }
                *reinterpret_cast<h4*>(ctx_hi + obase + 8 * g4) = ah;
                *reinterpret_cast<h4*>(ctx_lo + obase + 8 * g4) = al;
                *reinterpret_cast<h4*>(ctx_hi + obase + 32 + 8 * g4) = ch;
                *reinterpret_cast<h4*>(ctx_lo + obase + 32 + 8 * g4) = cl;
            } else {
                *reinterpret_cast<float4*>(ctx + obase + 8 * g4) = make_float4(a[0], a[1], a[2], a[3]);
                *reinterpret_cast<float4*>(ctx + obase + 32 + 8 * g4) = make_float4(c[0], c[1], c[2], c[3]);
            }
        }
    }
}

hipError_t launch_attention_f16x3(const _Float16* qhi, const _Float16* qlo, const _Float16* khi, const _Float16* klo,
                                  const _Float16* vthi, const _Float16* vtlo, const float* qp, const int32_t* frames,
                                  _Float16* ctx_hi, _Float16* ctx_lo, float* ctx, int B, int T, int Tp, hipStream_t s) {
    if (B <= 0 || T <= 0 || B > 65535 || Tp < T || (Tp % AX_BK) != 0) return hipErrorInvalidValue;
    const int nqb = (T + AX_BQ - 1) / AX_BQ;
    const long nblk = (long)nqb * kHeads * B;
    if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblk);
    if (ctx_hi)
        hipLaunchKernelGGL(attention_f16x3_kernel<true>, grid, dim3(256), 0, s, qhi, qlo, khi, klo, vthi, vtlo, qp, frames, ctx_hi, ctx_lo,
                           ctx, T, Tp, nqb);
    else
        hipLaunchKernelGGL(attention_f16x3_kernel<false>, grid, dim3(256), 0, s, qhi, qlo, khi, klo, vthi, vtlo, qp, frames, ctx_hi, ctx_lo,
                           ctx, T, Tp, nqb);
    return hipGetLastError();
}

}  // namespace loco

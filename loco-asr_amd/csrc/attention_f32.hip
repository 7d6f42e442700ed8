// Self-attention core of SpeechT5Attention (HF modeling_speecht5.py:911-982) as a flash-style fp32 kernel:
//
//   ctx[b,i,head,:] = softmax_j( q_i . k_j  +  qp[i, clip(i-j,-160,159)+160]  +  mask_j ) v_j
//
// with q already scaled by 1/8 (folded into the fused QKV projection, HF :891) and qp = q_scaled pe_k^T the
// compact form of HF's materialised [T,T,64] relative-position tensor (HF :432-441, :939-945; identity
// verified in SURVEY.md §7).  Neither the [T,T] scores nor the [T,T,64] table ever exist, so T = 29 999
// (10-minute clips) costs memory linear in T.
//
// Work split: one workgroup per (query block of 128, head, clip), XCD-aware 1-D grid; 4 waves per workgroup, each wave owns 32 query rows; K/V tiles of
// 64 keys are staged in LDS (K padded to 68 floats/row for conflict-free ds_read_b128, V unpadded: its
// reads are lane-contiguous) and shared by the 4 waves.  Two LDS buffers, one barrier per tile; global
// loads run two tiles ahead (they are parked in registers for one iteration, then published to the idle
// buffer at the top of the next); K fragments / V values are read one 16-/8-MFMA group ahead of the
// MFMAs that consume them, and the last PV group of a tile issues after the barrier so the matrix pipe
// never waits on LDS.
//
// MFMA orientation ("swapped" QK^T): S^T = K Q^T, so the C/D fragment puts the QUERY on the lane and the
// 32+32 keys of the tile in that lane's registers.  Row max / row sum are then register-local plus one
// exchange with lane^32, the running (m, l) state and the rescale of O^T are lane-local, and P^T feeds the
// second product O^T = V^T P^T directly as the MFMA B operand -- no LDS round trip, no transposes.
// v_mfma_f32_32x32x2_f32 sums over k = lane>>5 only, so each lane half owns a contiguous half of the
// head dimension (QK^T) / the key rows {jr, jr+4} of each accumulator register (PV).
//
// Relative-position bias: tiles whose every (i-j) is clipped read one per-row constant (qp[i,0] or
// qp[i,319]) held in a register; only the ~(320+96)/64 tiles around the diagonal gather from qp (L2).
// Masked keys (j >= frames[b]) get -inf before the online softmax, which is what HF's additive finfo.min
// becomes after exp(); padded QUERY rows are computed like any other (HF does, and the reference pickles them).
#include "loco_kernels.h"

namespace loco {

constexpr int AT_BQ = 128;
constexpr int AT_BK = 64;
constexpr int AT_LDK = kHeadDim + 4;
constexpr int AT_KT = AT_BK * AT_LDK;    // floats per K buffer
constexpr int AT_VT = AT_BK * kHeadDim;  // floats per V buffer

typedef _Float16 h4_t __attribute__((ext_vector_type(4)));

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ qkv, const float* __restrict__ qp,
                                                           const int32_t* __restrict__ frames, float* __restrict__ ctx,
                                                           _Float16* __restrict__ ctx_hi, _Float16* __restrict__ ctx_lo,
                                                           int T, int nqb) {
    // two K/V buffers: tile t+1 is written while tile t is consumed -> ONE barrier per tile
    __shared__ __attribute__((aligned(16))) float kl[2 * AT_KT];
    __shared__ __attribute__((aligned(16))) float vl[2 * AT_VT];
    __shared__ float bias_stage[4][32 * 17];  // per-wave transpose scratch for the diagonal band (32 queries x 16 keys)

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
    const int iw0 = qblk * AT_BQ + wave * 32;  // first query row of this wave
    const int iq = iw0 + r;
    const int iqc = iq < T ? iq : T - 1;

    int nvalid = frames ? frames[b] : T;
    if (nvalid <= 0 || nvalid > T) nvalid = T;
    const int ntiles = (nvalid + AT_BK - 1) / AT_BK;

    const float* base = qkv + (long)b * T * kQkv + head * kHeadDim;
    constexpr float kLog2e = 1.4426950408889634f;

    // Q fragment (B operand of S^T = K Q^T): Q[iq][32h + kk], kk = 0..31
    float q[32];
    {
        const f32x4* qr = reinterpret_cast<const f32x4*>(base + (long)iqc * kQkv + 32 * h);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = qr[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[4 * i + e] = v[e] * kLog2e;  // scores come out of the MFMA in the log2 domain
        }
    }
    const float* qprow = qp + (((long)b * kHeads + head) * T + iqc) * kRelN;
    const float c_past = qprow[kRelN - 1] * kLog2e;  // i - j >= 159
    const float c_future = qprow[0] * kLog2e;        // i - j <= -160

    // staging map: 16-byte piece f = tid + 256*u -> key row f/16, 4-float column f%16
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;
    f32x4 pk[4], pv[4];
#define AT_LOAD_TILE(t)                                                          \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                              \
        int j_ = (t) * AT_BK + srow + 16 * u;                                    \
        j_ = j_ < T ? j_ : T - 1;                                                \
        const float* rowp_ = base + (long)j_ * kQkv + sc4;                       \
        pk[u] = *reinterpret_cast<const f32x4*>(rowp_ + kHidden);                \
        pv[u] = *reinterpret_cast<const f32x4*>(rowp_ + 2 * kHidden);            \
    }
#define AT_STORE_TILE(buf)                                                                          \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                 \
        *reinterpret_cast<f32x4*>(kl + (buf) * AT_KT + (srow + 16 * u) * AT_LDK + sc4) = pk[u];     \
        *reinterpret_cast<f32x4*>(vl + (buf) * AT_VT + (srow + 16 * u) * kHeadDim + sc4) = pv[u];   \
    }
    // K fragment group G = (st, half): 4 x 16 bytes of row (32 st + r), head dims 32h + 16 half + [0,16)
#define AT_LOAD_K(kb, G, F)                                                                          \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                    \
        F[c] = *reinterpret_cast<const f32x4*>((kb) + (((G) >> 1) * 32 + r) * AT_LDK + 32 * h + 16 * ((G) & 1) + 4 * c);
#define AT_MFMA_K(G, F)                                                                              \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                    \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                \
            s[(G) >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(F[c][e], q[16 * ((G) & 1) + 4 * c + e], s[(G) >> 1], 0, 0, 0);
    // V group G = (st, e-quad): rows 32 st + 8 quad + 4h + {0..3}, columns r and 32 + r
#define AT_LOAD_V(vb, G, F)                                                                          \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
        const float* vr_ = (vb) + (((G) >> 2) * 32 + 8 * ((G) & 3) + 4 * h + e) * kHeadDim + r;      \
        F[2 * e] = vr_[0];                                                                           \
        F[2 * e + 1] = vr_[32];                                                                      \
    }
#define AT_MFMA_V(G, F)                                                                              \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(F[2 * e], s[(G) >> 2][4 * ((G) & 3) + e], o0, 0, 0, 0);     \
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F[2 * e + 1], s[(G) >> 2][4 * ((G) & 3) + e], o1, 0, 0, 0); \
    }
    // P for group G: exp2(s - m) in place, row-sum accumulated
#define AT_EXP(G)                                                                                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
        const float p_ = __builtin_amdgcn_exp2f(s[(G) >> 2][4 * ((G) & 3) + e] + dsh);               \
        s[(G) >> 2][4 * ((G) & 3) + e] = p_;                                                         \
        ps += p_;                                                                                    \
    }

    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    AT_LOAD_TILE(0)
    AT_STORE_TILE(0)
    if (ntiles > 1) { AT_LOAD_TILE(1) }
    __syncthreads();

    f32x4 ka[4], kb2[4];
    float va[8], vb2[8];
    f32x16 s[2];
    int cur = 0;
    AT_LOAD_K(kl, 0, ka)
    for (int t = 0; t < ntiles; ++t) {
        const float* kcur = kl + cur * AT_KT;
        const float* vcur = vl + cur * AT_VT;
        // publish tile t+1 (loaded one iteration ago) into the other buffer, then start fetching tile t+2
        if (t + 1 < ntiles) {
            AT_STORE_TILE(cur ^ 1)
            if (t + 2 < ntiles) { AT_LOAD_TILE(t + 2) }
        }
        const int j0 = t * AT_BK;

        // ---- S^T = K Q^T (64 MFMAs), K fragments one group ahead of the MFMAs
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[st][e] = 0.f;
        AT_LOAD_K(kcur, 1, kb2)
        __builtin_amdgcn_sched_barrier(0);
        AT_MFMA_K(0, ka)
        __builtin_amdgcn_sched_barrier(0);
        AT_LOAD_K(kcur, 2, ka)
        __builtin_amdgcn_sched_barrier(0);
        AT_MFMA_K(1, kb2)
        __builtin_amdgcn_sched_barrier(0);
        AT_LOAD_K(kcur, 3, kb2)
        __builtin_amdgcn_sched_barrier(0);
        AT_MFMA_K(2, ka)
        __builtin_amdgcn_sched_barrier(0);
        AT_LOAD_V(vcur, 0, va)  // first V group rides under the last K group
        __builtin_amdgcn_sched_barrier(0);
        AT_MFMA_K(3, kb2)
        __builtin_amdgcn_sched_barrier(0);

        // ---- relative-position bias + key mask.  Tiles whose every (i-j) is clipped add ONE per-row constant cb,
        // which is folded into the exponent shift (p = exp2(s + cb - m)) instead of touching the 32 scores; only
        // the diagonal band gathers per-element biases; only the last tile can hold masked keys.
        const int dmin = iw0 - (j0 + AT_BK - 1);  // smallest i-j in this wave tile
        const int dmax = iw0 + 31 - j0;           // largest
        float cb = 0.f;
        if (dmin >= kRelMax - 1) {
            cb = c_past;
        } else if (dmax <= -kRelMax) {
            cb = c_future;
        } else {
            // Diagonal band: bias[i][j] = qp[i][clip(i-j)+160].  In the MFMA layout a lane owns ONE query and 32 keys,
            // so a direct gather touches 64 different cache lines per load instruction.  Instead each wave reads the
            // band with the KEY on the lane (32 consecutive table entries per row: coalesced), transposes 32x16
            // pieces through a private LDS scratch and adds them to the scores.
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
                    for (int u = 0; u < 8; ++u) sc[66 * u + 16 * lq + (lq >> 1) + lj] = bv[u];  // row r at 16 r + r / 2: attention_f16x3.hip, AX_BAND_ADD
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int e8 = 0; e8 < 8; ++e8) {
                        const int e = 8 * half + e8;  // keys 16*half + (e8&3) + 8*(e8>>2) + 4h
                        s[st][e] = fmaf(sc[16 * r + (r >> 1) + 4 * h + (e8 & 3) + 8 * (e8 >> 2)], kLog2e, s[st][e]);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (j0 + AT_BK > nvalid) {
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
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) + cb;

        // ---- online softmax (the query row lives on lanes r and r+32); O is rescaled only when some row's max moved
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        const float dsh = cb - m_new;
        float ps = 0.f;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
        }
        AT_EXP(0)

        // ---- O^T += V^T P^T (64 MFMAs): V values one group ahead, next group's exp2 beside this group's MFMAs
#define AT_PV_STEP(G, FA, FB)                     \
        AT_LOAD_V(vcur, (G) + 1, FB)              \
        __builtin_amdgcn_sched_barrier(0);        \
        AT_MFMA_V(G, FA)                          \
        AT_EXP((G) + 1)                           \
        __builtin_amdgcn_sched_barrier(0);
        AT_PV_STEP(0, va, vb2)
        AT_PV_STEP(1, vb2, va)
        AT_PV_STEP(2, va, vb2)
        AT_PV_STEP(3, vb2, va)
        AT_PV_STEP(4, va, vb2)
        AT_PV_STEP(5, vb2, va)
        AT_PV_STEP(6, va, vb2)
#undef AT_PV_STEP
        l_run = l_run * alpha + ps;

        __syncthreads();  // tile t+1 is complete in the other buffer; everyone is done with this one except group 7's V (in registers)
        cur ^= 1;
        if (t + 1 < ntiles) { AT_LOAD_K(kl + cur * AT_KT, 0, ka) }
        __builtin_amdgcn_sched_barrier(0);
        AT_MFMA_V(7, vb2)  // the last 8 MFMAs cover the barrier skew and the LDS latency of the next tile's first K group
        __builtin_amdgcn_sched_barrier(0);
    }
#undef AT_LOAD_TILE
#undef AT_STORE_TILE
#undef AT_LOAD_K
#undef AT_MFMA_K
#undef AT_LOAD_V
#undef AT_MFMA_V
#undef AT_EXP

    // ---- normalise and store: o{0,1}[e] = O[iq][d = 32*dt + (e&3) + 8*(e>>2) + 4h]
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
            if (SPLIT) {  // fp16 hi/lo planes: the A operand of the split-precision out-projection GEMM
                h4_t ah, al, ch, cl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    asm volatile("" : "+v"(a[e]), "+v"(c[e]));
                    ah[e] = (_Float16)a[e]; al[e] = (_Float16)(a[e] - (float)ah[e]);
                    ch[e] = (_Float16)c[e]; cl[e] = (_Float16)(c[e] - (float)ch[e]);
                }
                *reinterpret_cast<h4_t*>(ctx_hi + obase + 8 * g4) = ah;
                *reinterpret_cast<h4_t*>(ctx_lo + obase + 8 * g4) = al;
                *reinterpret_cast<h4_t*>(ctx_hi + obase + 32 + 8 * g4) = ch;
                *reinterpret_cast<h4_t*>(ctx_lo + obase + 32 + 8 * g4) = cl;
            } else {
                *reinterpret_cast<float4*>(ctx + obase + 8 * g4) = make_float4(a[0], a[1], a[2], a[3]);
                *reinterpret_cast<float4*>(ctx + obase + 32 + 8 * g4) = make_float4(c[0], c[1], c[2], c[3]);
            }
        }
    }
}

hipError_t launch_attention(const float* qkv, const float* qp, const int32_t* frames, float* ctx, int B, int T,
                            hipStream_t s, void* ctx_hi, void* ctx_lo) {
    if (B <= 0 || T <= 0 || B > 65535) return hipErrorInvalidValue;
    const int nqb = (T + AT_BQ - 1) / AT_BQ;
    const long nblk = (long)nqb * kHeads * B;
    if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblk);
    if (ctx_hi)
        hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(256), 0, s, qkv, qp, frames, ctx, (_Float16*)ctx_hi, (_Float16*)ctx_lo, T, nqb);
    else
        hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(256), 0, s, qkv, qp, frames, ctx, (_Float16*)nullptr, (_Float16*)nullptr, T, nqb);
    return hipGetLastError();
}

}  // namespace loco

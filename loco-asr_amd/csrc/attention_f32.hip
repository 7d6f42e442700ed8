// Self-attention core of SpeechT5Attention (HF modeling_speecht5.py:911-982) as a flash-style fp32 kernel:
//
//   ctx[b,i,head,:] = softmax_j( q_i . k_j  +  qp[i, clip(i-j,-160,159)+160]  +  mask_j ) v_j
//
// with q already scaled by 1/8 (folded into the fused QKV projection, HF :891) and qp = q_scaled pe_k^T the
// compact form of HF's materialised [T,T,64] relative-position tensor (HF :432-441, :939-945; identity
// verified in SURVEY.md §7).  Neither the [T,T] scores nor the [T,T,64] table ever exist, so T = 29 999
// (10-minute clips) costs memory linear in T.
//
// Work split: grid (T/128, 12 heads, B); 4 waves per workgroup, each wave owns 32 query rows; K/V tiles of
// 64 keys are staged in LDS (K padded to 68 floats/row for conflict-free ds_read_b128, V unpadded: its
// reads are lane-contiguous) and shared by the 4 waves; the next tile's global loads are in flight while
// the current tile is consumed.
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

__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ qkv, const float* __restrict__ qp,
                                                           const int32_t* __restrict__ frames, float* __restrict__ ctx,
                                                           int T) {
    __shared__ __attribute__((aligned(16))) float kl[AT_BK * AT_LDK];
    __shared__ __attribute__((aligned(16))) float vl[AT_BK * kHeadDim];

    const int b = blockIdx.z, head = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int iw0 = blockIdx.x * AT_BQ + wave * 32;  // first query row of this wave
    const int iq = iw0 + r;
    const int iqc = iq < T ? iq : T - 1;

    int nvalid = frames ? frames[b] : T;
    if (nvalid <= 0 || nvalid > T) nvalid = T;
    const int ntiles = (nvalid + AT_BK - 1) / AT_BK;

    const float* base = qkv + (long)b * T * kQkv + head * kHeadDim;

    // Q fragment (B operand of S^T = K Q^T): Q[iq][32h + kk], kk = 0..31
    float q[32];
    {
        const float4* qr = reinterpret_cast<const float4*>(base + (long)iqc * kQkv + 32 * h);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 v = qr[i];
            q[4 * i + 0] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w;
        }
    }
    const float* qprow = qp + (((long)b * kHeads + head) * T + iqc) * kRelN;
    const float c_past = qprow[kRelN - 1];  // i - j >= 159
    const float c_future = qprow[0];        // i - j <= -160

    // staging map: float4 f = tid + 256*u -> key row f/16, 4-float column f%16
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;
    f32x4 pk[4], pv[4];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int j = t * AT_BK + srow + 16 * u;
            j = j < T ? j : T - 1;
            const float* rowp = base + (long)j * kQkv + sc4;
            pk[u] = *reinterpret_cast<const f32x4*>(rowp + kHidden);
            pv[u] = *reinterpret_cast<const f32x4*>(rowp + 2 * kHidden);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<f32x4*>(kl + (srow + 16 * u) * AT_LDK + sc4) = pk[u];
            *reinterpret_cast<f32x4*>(vl + (srow + 16 * u) * kHeadDim + sc4) = pv[u];
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    constexpr float kLog2e = 1.4426950408889634f;

    load_tile(0);
    store_tile();
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) load_tile(t + 1);
        const int j0 = t * AT_BK;

        // ---- S^T = K Q^T : 2 sub-tiles of 32 keys x 32 queries
        f32x16 s[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[st][e] = 0.f;
            const float* kr = kl + (st * 32 + r) * AT_LDK + 32 * h;
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kr + 4 * k4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    s[st] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], q[4 * k4 + e], s[st], 0, 0, 0);
            }
        }

        // ---- relative-position bias, key mask, log2 domain
        const int dmin = iw0 - (j0 + AT_BK - 1);  // smallest i-j in this wave tile
        const int dmax = iw0 + 31 - j0;           // largest
        float mx = -INFINITY;
        if (dmin >= kRelMax - 1 || dmax <= -kRelMax) {
            const float cb = dmin >= kRelMax - 1 ? c_past : c_future;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int j = j0 + 32 * st + (e & 3) + 8 * (e >> 2) + 4 * h;
                    float v = (s[st][e] + cb) * kLog2e;
                    v = j < nvalid ? v : -INFINITY;
                    s[st][e] = v;
                    mx = fmaxf(mx, v);
                }
        } else {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int j = j0 + 32 * st + (e & 3) + 8 * (e >> 2) + 4 * h;
                    int rel = iq - j;
                    rel = rel < -kRelMax ? -kRelMax : (rel > kRelMax - 1 ? kRelMax - 1 : rel);
                    float v = (s[st][e] + qprow[rel + kRelMax]) * kLog2e;
                    v = j < nvalid ? v : -INFINITY;
                    s[st][e] = v;
                    mx = fmaxf(mx, v);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));

        // ---- online softmax (the query row lives on lanes r and r+32)
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float ps = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(s[st][e] - m_new);
                s[st][e] = p;
                ps += p;
            }
        l_run = l_run * alpha + ps;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }

        // ---- O^T += V^T P^T : register e of P^T carries keys {jr, jr+4} (one per lane half)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float* vr = vl + (st * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * kHeadDim + r;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], s[st][e], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], s[st][e], o1, 0, 0, 0);
            }

        __syncthreads();  // every wave is done reading this tile
        if (more) store_tile();
        __syncthreads();
    }

    // ---- normalise and store: o{0,1}[e] = O[iq][d = 32*dt + (e&3) + 8*(e>>2) + 4h]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (iq < T) {
        float* orow = ctx + ((long)b * T + iq) * kHidden + head * kHeadDim + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float4 a, c;
            a.x = o0[4 * g4 + 0] * inv; a.y = o0[4 * g4 + 1] * inv; a.z = o0[4 * g4 + 2] * inv; a.w = o0[4 * g4 + 3] * inv;
            c.x = o1[4 * g4 + 0] * inv; c.y = o1[4 * g4 + 1] * inv; c.z = o1[4 * g4 + 2] * inv; c.w = o1[4 * g4 + 3] * inv;
            *reinterpret_cast<float4*>(orow + 8 * g4) = a;
            *reinterpret_cast<float4*>(orow + 32 + 8 * g4) = c;
        }
    }
}

hipError_t launch_attention(const float* qkv, const float* qp, const int32_t* frames, float* ctx, int B, int T,
                            hipStream_t s) {
    if (B <= 0 || T <= 0 || B > 65535) return hipErrorInvalidValue;
    dim3 grid((T + AT_BQ - 1) / AT_BQ, kHeads, B);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, s, qkv, qp, frames, ctx, T);
    return hipGetLastError();
}

}  // namespace loco

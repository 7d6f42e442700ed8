// LayerNorm (one 64-lane wavefront per row, shuffle reductions), frame counts, weight preparation and
// the sinusoidal position table.  All HBM-bound byte movers: float4 accesses, no LDS.
#include "loco_kernels.h"

namespace loco {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// y = (x - mean) * rsqrt(var + eps) * gamma + beta, biased variance, two-pass over registers
// (HF nn.LayerNorm sites: modeling_speecht5.py:501,1023,1025,1276).  NV4 = dim / 256.
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));

// SPLIT: additionally (or instead, when y == nullptr) write the result as fp16 hi/lo planes -- the A operand of the
// split-precision GEMM that consumes it (gemm_f16x3.hip).
template <int NV4, bool SPLIT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y,
                                                        _Float16* __restrict__ yhi, _Float16* __restrict__ ylo, long rows,
                                                        float eps, float* __restrict__ nonfinite_slot) {
    constexpr int D = NV4 * 256;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    float4 v[NV4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        v[i] = xr[lane + 64 * i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
    // Finite check of the forward's LAST LayerNorm (loco_api.hip): an inf or NaN anywhere in the row makes its mean, hence its
    // variance, NaN -- one wave-uniform compare per row, and an atomic only for a row that is bad (the word stays 0 otherwise).
    // The range words cannot see NaNs (fmaxf drops them); whatever a stage upstream broke ends up in this row statistic.
    if (nonfinite_slot && lane == 0 && !(rstd > 0.f)) atomicMax(reinterpret_cast<unsigned*>(nonfinite_slot) + (blockIdx.x & 7), 0x7f800000u);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const float4 gg = reinterpret_cast<const float4*>(g)[lane + 64 * i];
        const float4 bb = reinterpret_cast<const float4*>(b)[lane + 64 * i];
        float o[4];
        o[0] = v[i].x * rstd * gg.x + bb.x;
        o[1] = v[i].y * rstd * gg.y + bb.y;
        o[2] = v[i].z * rstd * gg.z + bb.z;
        o[3] = v[i].w * rstd * gg.w + bb.w;
        if (y) reinterpret_cast<float4*>(y + row * D)[lane + 64 * i] = make_float4(o[0], o[1], o[2], o[3]);
        if (SPLIT) {
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            unsigned h0, l0, h1, l1;
            split_f16_2pairs(o[0], o[1], o[2], o[3], h0, l0, h1, l1);
            reinterpret_cast<u32x2_t*>(yhi + row * D)[lane + 64 * i] = u32x2_t{h0, h1};
            reinterpret_cast<u32x2_t*>(ylo + row * D)[lane + 64 * i] = u32x2_t{l0, l1};
        }
    }
}

hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, long rows, int dim, float eps,
                            hipStream_t s, void* yhi, void* ylo, float* nonfinite_slot) {
    if (rows <= 0 || (!y && !yhi) || ((yhi == nullptr) != (ylo == nullptr))) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    _Float16* hi = (_Float16*)yhi;
    _Float16* lo = (_Float16*)ylo;
    if (dim == 768 && !hi)
        hipLaunchKernelGGL((layernorm_kernel<3, false>), dim3(grid), dim3(256), 0, s, x, g, b, y, hi, lo, rows, eps, nonfinite_slot);
    else if (dim == 768)
        hipLaunchKernelGGL((layernorm_kernel<3, true>), dim3(grid), dim3(256), 0, s, x, g, b, y, hi, lo, rows, eps, nonfinite_slot);
    else if (dim == 512 && !hi)
        hipLaunchKernelGGL((layernorm_kernel<2, false>), dim3(grid), dim3(256), 0, s, x, g, b, y, hi, lo, rows, eps, nonfinite_slot);
    else if (dim == 512)
        hipLaunchKernelGGL((layernorm_kernel<2, true>), dim3(grid), dim3(256), 0, s, x, g, b, y, hi, lo, rows, eps, nonfinite_slot);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// frames[b] = conv-chain output length of sum(mask[b,:])   (HF modeling:569-598; HF uses cumsum(-1)[-1] = the sum)
// Stage 1 streams the mask with 16-byte loads over a (parts, B) grid and adds per-block counts into frames[b]
// (integer atomics: order-independent, so the result is exact and reproducible); stage 2 converts in place.
__global__ __launch_bounds__(256) void mask_count_kernel(const int32_t* __restrict__ mask, long L, int32_t* __restrict__ frames) {
    __shared__ int part[4];
    const int b = blockIdx.y;
    const int32_t* m = mask + (long)b * L;
    const long per = (((L + gridDim.x - 1) / gridDim.x) + 3) & ~3L;
    const long i0 = (long)blockIdx.x * per;
    long i1 = i0 + per;
    i1 = i1 < L ? i1 : L;
    i1 = i1 < i0 ? i0 : i1;  // blocks past the end contribute nothing
    int n = 0;
    if ((reinterpret_cast<uintptr_t>(m) & 15) == 0) {
        const long v1 = i0 + ((i1 - i0) & ~3L);
        for (long i = i0 + 4 * threadIdx.x; i < v1; i += 1024) {
            const int4 v = *reinterpret_cast<const int4*>(m + i);
            n += (v.x + v.y) + (v.z + v.w);
        }
        for (long i = v1 + threadIdx.x; i < i1; i += 256) n += m[i];
    } else {
        for (long i = i0 + threadIdx.x; i < i1; i += 256) n += m[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&frames[b], (part[0] + part[1]) + (part[2] + part[3]));
}

__global__ void frames_from_counts_kernel(int32_t* __restrict__ frames, int B, long L, int have_mask) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    long n = have_mask ? (long)frames[b] : L;
    const int ks[7] = {10, 3, 3, 3, 3, 2, 2}, ss[7] = {5, 2, 2, 2, 2, 2, 2};
    for (int i = 0; i < 7; ++i) {
        // torch.div(n - k, s, rounding_mode="floor") + 1: floor division also for negative n - k
        const long d = n - ks[i];
        n = (d >= 0 ? d / ss[i] : -((-d + ss[i] - 1) / ss[i])) + 1;
    }
    frames[b] = (int32_t)n;
}

hipError_t launch_frame_counts(const int32_t* mask, int B, long L, int32_t* frames, hipStream_t s) {
    if (B <= 0 || B > 65535) return hipErrorInvalidValue;
    if (mask) {
        hipError_t e = hipMemsetAsync(frames, 0, (size_t)B * sizeof(int32_t), s);
        if (e != hipSuccess) return e;
        long parts = (L + 65535) / 65536;  // >= 64 Ki samples per block
        parts = parts < 1 ? 1 : (parts > 256 ? 256 : parts);
        hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)parts, B), dim3(256), 0, s, mask, L, frames);
    }
    hipLaunchKernelGGL(frames_from_counts_kernel, dim3((B + 63) / 64), dim3(64), 0, s, frames, B, L, mask ? 1 : 0);
    return hipGetLastError();
}

// ---- zero_mean_unit_var_norm on the device ("next" row f-4) -------------------------------------------------------
// HF feature_extraction_speecht5.py:119-138 (do_normalize): per clip, over its n = sum(mask) UNPADDED samples,
//   y[t] = (x[t] - mean) / sqrt(var + 1e-7)   for t < n   (population variance),   y[t] = padding_value   for t >= n.
// Moments are accumulated in fp64 in a fixed order (thread-strided partial sums, a fixed shared-memory tree, then the
// parts in sequence), so the result is bitwise reproducible; numpy's fp32 pairwise sums differ from it by ~1e-7 relative.
constexpr int kNormParts = 128;

__global__ __launch_bounds__(256) void wav_moments_kernel(const float* __restrict__ x, long L, const int32_t* __restrict__ counts,
                                                          double* __restrict__ part) {
    const int b = blockIdx.y;
    const long n = counts ? counts[b] : L;
    const long chunk = ((n + kNormParts - 1) / kNormParts + 3) & ~3L;
    long i0 = (long)blockIdx.x * chunk, i1 = i0 + chunk;
    i1 = i1 < n ? i1 : n;
    const float* xb = x + (long)b * L;
    double s = 0.0, q = 0.0;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const double v = xb[i];
        s += v;
        q += v * v;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + w];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[((long)b * kNormParts + blockIdx.x) * 2] = sh[0][0];
        part[((long)b * kNormParts + blockIdx.x) * 2 + 1] = sh[1][0];
    }
}

__global__ void wav_norm_coeffs_kernel(const double* __restrict__ part, const int32_t* __restrict__ counts, long L, int B,
                                       float* __restrict__ coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const long n = counts ? counts[b] : L;
    double s = 0.0, q = 0.0;
    for (int p = 0; p < kNormParts; ++p) {
        s += part[((long)b * kNormParts + p) * 2];
        q += part[((long)b * kNormParts + p) * 2 + 1];
    }
    const double mean = n > 0 ? s / (double)n : 0.0;
    double var = n > 0 ? q / (double)n - mean * mean : 0.0;
    var = var > 0.0 ? var : 0.0;
    coef[2 * b] = (float)mean;
    coef[2 * b + 1] = (float)(1.0 / sqrt(var + 1e-7));
}

__global__ __launch_bounds__(256) void wav_norm_apply_kernel(const float* __restrict__ x, long L, const int32_t* __restrict__ counts,
                                                             const float* __restrict__ coef, float pad, float* __restrict__ out) {
    const int b = blockIdx.y;
    const long n = counts ? counts[b] : L;
    const float mean = coef[2 * b], rstd = coef[2 * b + 1];
    const float* xb = x + (long)b * L;
    float* ob = out + (long)b * L;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256)
        ob[i] = i < n ? (xb[i] - mean) * rstd : pad;
}

size_t normalize_scratch_bytes(int B) { return ((size_t)B * (kNormParts * 2 * sizeof(double) + 2 * sizeof(float) + sizeof(int32_t)) + 255) & ~size_t(255); }

hipError_t launch_normalize_waveform(const float* wav, const int32_t* mask, int B, long L, float pad, float* out, void* scratch,
                                     hipStream_t s) {
    if (B <= 0 || B > 65535 || L <= 0) return hipErrorInvalidValue;
    double* part = reinterpret_cast<double*>(scratch);
    float* coef = reinterpret_cast<float*>(part + (size_t)B * kNormParts * 2);
    int32_t* counts = nullptr;
    if (mask) {
        counts = reinterpret_cast<int32_t*>(coef + (size_t)2 * B);
        hipError_t e = hipMemsetAsync(counts, 0, (size_t)B * sizeof(int32_t), s);
        if (e != hipSuccess) return e;
        long parts = (L + 65535) / 65536;
        parts = parts < 1 ? 1 : (parts > 256 ? 256 : parts);
        hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)parts, B), dim3(256), 0, s, mask, L, counts);
    }
    hipLaunchKernelGGL(wav_moments_kernel, dim3(kNormParts, B), dim3(256), 0, s, wav, L, counts, part);
    hipLaunchKernelGGL(wav_norm_coeffs_kernel, dim3((B + 63) / 64), dim3(64), 0, s, part, counts, L, B, coef);
    long blocks = (L + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(wav_norm_apply_kernel, dim3((unsigned)blocks, B), dim3(256), 0, s, wav, L, counts, coef, pad, out);
    return hipGetLastError();
}

// Token-level valid counts for the text front end: frames[b] = sum_t mask[b,t] (the encoder's key mask is a prefix mask:
// the tokenizer pads on the right), or T without a mask.
__global__ void token_counts_kernel(const int32_t* __restrict__ mask, int B, int T, int32_t* __restrict__ frames) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = T;
    if (mask) {
        n = 0;
        for (int t = 0; t < T; ++t) n += mask[(long)b * T + t] != 0;
    }
    frames[b] = n;
}

hipError_t launch_token_counts(const int32_t* mask, int B, int T, int32_t* frames, hipStream_t s) {
    if (B <= 0 || T <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(token_counts_kernel, dim3((B + 63) / 64), dim3(64), 0, s, mask, B, T, frames);
    return hipGetLastError();
}

// SpeechT5TextEncoderPrenet (HF modeling_speecht5.py: embed_tokens + SpeechT5ScaledPositionalEncoding):
//   out[b,t,:] = embed[ids[b,t], :] + alpha * pe[t, :]          (dropout is the identity in eval)
// one thread = 4 channels; ids outside [0, vocab) are clamped (the host wrapper rejects them, as nn.Embedding would).
__global__ void text_prenet_kernel(const int32_t* __restrict__ ids, const float* __restrict__ embed, int vocab,
                                   const float* __restrict__ alpha, const float* __restrict__ pe, int T, long n4,
                                   float* __restrict__ out) {
    const float a = alpha[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long m = i / (kHidden / 4);
        const int c4 = (int)(i - m * (kHidden / 4));
        const int t = (int)(m % T);
        int id = ids[m];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4 ev = reinterpret_cast<const f32x4*>(embed + (long)id * kHidden)[c4];
        const f32x4 pv = reinterpret_cast<const f32x4*>(pe + (long)t * kHidden)[c4];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(ev[e], __fmul_rn(a, pv[e]));  // HF: emb + (alpha * pe), two roundings: no FMA
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

hipError_t launch_text_prenet(const int32_t* ids, const float* embed, int vocab, const float* alpha, const float* pe, int B, int T,
                              float* out, hipStream_t s) {
    if (B <= 0 || T <= 0 || vocab <= 0) return hipErrorInvalidValue;
    const long n4 = (long)B * T * (kHidden / 4);
    const long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(text_prenet_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, ids, embed, vocab, alpha, pe, T,
                       n4, out);
    return hipGetLastError();
}

// The library's own table for C callers (the Python host uploads the table computed with HF's torch expression):
// pe[p, 2k] = sin(p * w_k), pe[p, 2k+1] = cos(p * w_k), w_k = exp(2k * -(ln 10000 / 768))
__global__ void text_pe_table_kernel(float* __restrict__ pe, int rows) {
    const long total = (long)rows * (kHidden / 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % (kHidden / 2));
        const int pos = (int)(i / (kHidden / 2));
        const float w = expf((float)(2 * k) * -(logf(10000.0f) / (float)kHidden));
        const float ang = (float)pos * w;
        pe[(long)pos * kHidden + 2 * k] = sinf(ang);
        pe[(long)pos * kHidden + 2 * k + 1] = cosf(ang);
    }
}

hipError_t launch_text_pe_table(float* pe, int rows, hipStream_t s) {
    hipLaunchKernelGGL(text_pe_table_kernel, dim3(256), dim3(256), 0, s, pe, rows);
    return hipGetLastError();
}

// Conv1d weight [N, C, k] -> tap-major [N, k*C] so that a channels-last input row run is the GEMM's A row.
__global__ void relayout_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int C, int k) {
    const long total = (long)N * C * k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int tap = (int)((i / C) % k);
        const int n = (int)(i / ((long)C * k));
        out[i] = w[((long)n * C + c) * k + tap];
    }
}

hipError_t launch_relayout_conv_weight(const float* w, float* out, int N, int C, int k, hipStream_t s) {
    hipLaunchKernelGGL(relayout_conv_weight_kernel, dim3(1024), dim3(256), 0, s, w, out, N, C, k);
    return hipGetLastError();
}

// weight_norm(dim=2) of the positional conv (HF modeling:358-379): w[o,i,tap] = g[tap] * v[o,i,tap] / ||v[:,:,tap]||,
// the norm over all 768*48 (o,i) pairs of one tap; written as [group][tap][o_local][i].
__global__ __launch_bounds__(256) void fold_pos_conv_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                            float* __restrict__ out) {
    __shared__ double part[4];
    const int tap = blockIdx.x;
    double ss = 0.0;
    for (int e = threadIdx.x; e < kHidden * kPosCg; e += 256) {
        const double x = v[(long)e * kPosK + tap];
        ss += x * x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float norm = (float)sqrt(part[0] + part[1] + part[2] + part[3]);
    const float scale = g[tap] / norm;
    for (int e = threadIdx.x; e < kHidden * kPosCg; e += 256) {
        const int i = e % kPosCg, o = e / kPosCg;
        const int grp = o / kPosCg, ol = o % kPosCg;
        out[(((long)grp * kPosK + tap) * kPosCg + ol) * kPosCg + i] = v[(long)e * kPosK + tap] * scale;
    }
}

hipError_t launch_fold_pos_conv(const float* g, const float* v, float* out, hipStream_t s) {
    hipLaunchKernelGGL(fold_pos_conv_kernel, dim3(kPosK), dim3(256), 0, s, g, v, out);
    return hipGetLastError();
}

__global__ void scale_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = src[i] * scale;
}

hipError_t launch_scale_copy(const float* src, float* dst, long n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(scale_copy_kernel, dim3(512), dim3(256), 0, s, src, dst, n, scale);
    return hipGetLastError();
}

__global__ void permute_conv_k_kernel(const float* __restrict__ src, float* __restrict__ dst, long total, int taps, int C) {
    const int K = taps * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kp = (int)(i % K);  // destination position: k-tile kt = (64-channel block, tap slot, half), channel cc of the half
        const long n = i / K;
        const int cc = kp & 31, kt = kp >> 5;
        const int half = kt & 1, q = kt >> 1, slot = q % taps, cbp = q / taps;
        const int tap = taps == 3 ? ((0x18 >> (2 * slot)) & 3) : slot;  // slots 0, 1, 2 -> taps 0, 2, 1
        dst[i] = src[n * K + (long)tap * C + (2 * cbp + half) * 32 + cc];
    }
}

hipError_t launch_permute_conv_k(const float* src, float* dst, int N, int taps, int C, hipStream_t s) {
    if (N <= 0 || taps <= 0 || taps > 3 || C <= 0 || (C & 63)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(permute_conv_k_kernel, dim3(1024), dim3(256), 0, s, src, dst, (long)N * taps * C, taps, C);
    return hipGetLastError();
}

// Sinusoidal position table (HF modeling:305-321): row p = [sin(p*w_k) | cos(p*w_k)], w_k = exp(-k*ln(1e4)/383),
// every step rounded to fp32 like the torch expression it restates; row 1 (the padding row) is zero.
__global__ void sinusoid_table_kernel(float* __restrict__ tab, int rows) {
    const long total = (long)rows * (kHidden / 2);
    const float c = (float)(-9.210340371976184 / (double)(kHidden / 2 - 1));  // -ln(1e4)/383 rounded once from double, as torch does
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % (kHidden / 2));
        const int p = (int)(i / (kHidden / 2));
        const float w = expf(__fmul_rn((float)k, c));
        const float ang = __fmul_rn((float)p, w);
        float sv = sinf(ang), cv = cosf(ang);
        if (p == 1) sv = cv = 0.f;
        tab[(long)p * kHidden + k] = sv;
        tab[(long)p * kHidden + kHidden / 2 + k] = cv;
    }
}

hipError_t launch_sinusoid_table(float* tab, int rows, hipStream_t s) {
    hipLaunchKernelGGL(sinusoid_table_kernel, dim3(1024), dim3(256), 0, s, tab, rows);
    return hipGetLastError();
}

// max |x| (used once per weight tensor at load time to choose its power-of-two plane scale)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}

hipError_t launch_absmax(const float* x, long n, float* out, hipStream_t s) {
    if (n <= 0 || !x || !out) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, x, n, out);
    return hipGetLastError();
}

// diagnostics (LOCO_DEBUG_NONFINITE=1): number of non-finite elements of an fp32 or fp16 buffer -> out[0] += count, out[1] = first index + 1
template <typename T>
__global__ void count_nonfinite_kernel(const T* __restrict__ x, long n, unsigned long long* __restrict__ out) {
    unsigned long long cnt = 0, first = ~0ull;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = (float)x[i];
        if (!(fabsf(v) <= 3.4e38f)) {
            ++cnt;
            if ((unsigned long long)i < first) first = (unsigned long long)i;
        }
    }
    if (cnt) {
        atomicAdd(out, cnt);
        atomicMin(out + 1, first);
    }
}

hipError_t launch_count_nonfinite(const void* x, long n, bool half, unsigned long long* out, hipStream_t s) {
    if (n <= 0 || !x || !out) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (half) hipLaunchKernelGGL(count_nonfinite_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)x, n, out);
    else hipLaunchKernelGGL(count_nonfinite_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, n, out);
    return hipGetLastError();
}

}  // namespace loco

// Feature-encoder layer 0: Conv1d(1 -> 512, k = 10, s = 5, no bias) + GroupNorm(512 groups) + GELU
// (HF modeling_speecht5.py:260-281), channels-last output [B, T0, 512].
//
// GroupNorm with one group per channel normalises each (clip, channel) over the WHOLE padded time axis
// (SURVEY.md §7 hard part 3/5: the zero tail is part of the statistics).  The layer-0 activation is
// 6.3 GB at 30 s x 32, so we never make a pass over it just for statistics.  Because the conv has a
// single input channel, y_c[t] = sum_k w[c,k] x[5t+k] gives
//     mean_c   = sum_k  w[c,k] m1[k],                 m1[k]    = mean_t x[5t+k]
//     E[y_c^2] = sum_kk' w[c,k] w[c,k'] m2[k,k'],     m2[k,k'] = mean_t x[5t+k] x[5t+k']
// i.e. all 512 channel statistics follow from 10 first and 55 second moments of the strided waveform:
// one read of the 1.9 MB clip instead of a pass over 196 MB of activations.  Moments and the quadratic
// form are accumulated in fp64 (var = E[y^2] - mean^2 is then safe) through a fixed-order two-stage
// reduction, so results are bitwise reproducible run to run.
//
//   k1 conv0_moments   : grid (kConv0Parts, B)  partial sums -> scratch[b][part][65]   (fp64)
//   k2 conv0_gn_coeffs : grid (B)               mean[b,c], scale[b,c] = gamma_c * rstd (fp32)
//   k3 conv0_apply     : out = GELU((conv(x) - mean) * scale + beta), HBM-write-bound (2 KB row per frame)
#include "loco_kernels.h"

namespace loco {

size_t conv0_scratch_bytes(int B) {
    // fp64 partial moments + fp64 totals + fp32 (mean, scale) per (b, c)
    return (size_t)B * (kConv0Parts + 1) * kConv0Moments * sizeof(double) + (size_t)B * kConvDim * 2 * sizeof(float);
}

__global__ __launch_bounds__(256) void conv0_moments_kernel(const float* __restrict__ wav, long L, long T0,
                                                            double* __restrict__ partial, const int32_t* __restrict__ t0_clip) {
    __shared__ double red[4][kConv0Moments];
    const int b = blockIdx.y, part = blockIdx.x;
    const float* x = wav + (long)b * L;
    // packed forward (loco_forward_packed): the statistics of clip b run over the conv frames of ITS OWN reference batch's padded
    // length, cut into the same kConv0Parts pieces as a forward of that batch alone would cut them -- the same fp64 sums, bit for bit
    if (t0_clip) T0 = t0_clip[b];
    const long per = (T0 + kConv0Parts - 1) / kConv0Parts;
    const long t_begin = part * per, t_end = (t_begin + per < T0) ? t_begin + per : T0;
    double acc[kConv0Moments];
#pragma unroll
    for (int i = 0; i < kConv0Moments; ++i) acc[i] = 0.0;
    for (long t = t_begin + threadIdx.x; t < t_end; t += 256) {
        double v[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) v[k] = (double)x[5 * t + k];
        int idx = 10;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            acc[k] += v[k];
#pragma unroll
            for (int k2 = k; k2 < 10; ++k2) acc[idx++] += v[k] * v[k2];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kConv0Moments; ++i) {
        double s = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < kConv0Moments)
        partial[((long)b * kConv0Parts + part) * kConv0Moments + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(512) void conv0_gn_coeffs_kernel(const double* __restrict__ partial, double* __restrict__ total,
                                                              const float* __restrict__ w, const float* __restrict__ gn_w,
                                                              long T0, float eps, float* __restrict__ mean_out,
                                                              float* __restrict__ scale_out, const int32_t* __restrict__ t0_clip) {
    __shared__ double m[kConv0Moments];
    const int b = blockIdx.x;
    if (t0_clip) T0 = t0_clip[b];
    if (threadIdx.x < kConv0Moments) {
        double s = 0.0;
        for (int p = 0; p < kConv0Parts; ++p) s += partial[((long)b * kConv0Parts + p) * kConv0Moments + threadIdx.x];
        s /= (double)T0;
        m[threadIdx.x] = s;
        total[(long)b * kConv0Moments + threadIdx.x] = s;
    }
    __syncthreads();
    const int c = threadIdx.x;  // 512 threads = 512 channels
    double wk[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) wk[k] = (double)w[c * 10 + k];
    double mean = 0.0, ey2 = 0.0;
    int idx = 10;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        mean += wk[k] * m[k];
#pragma unroll
        for (int k2 = k; k2 < 10; ++k2) {
            const double term = wk[k] * wk[k2] * m[idx++];
            ey2 += (k2 == k) ? term : 2.0 * term;
        }
    }
    double var = ey2 - mean * mean;
    var = var > 0.0 ? var : 0.0;
    mean_out[b * kConvDim + c] = (float)mean;
    scale_out[b * kConvDim + c] = (float)((double)gn_w[c] / sqrt(var + (double)eps));
}

// 256 threads, thread j owns channels (2j, 2j+1); a block walks kFramesPerBlock consecutive frames of one clip.
// The waveform window of the block sits in LDS and is read by broadcast (every lane the same address).
constexpr int kFramesPerBlock = 64;

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

template <bool SPLIT>
__global__ __launch_bounds__(256) void conv0_apply_kernel(const float* __restrict__ wav, long L, long T0,
                                                          const float* __restrict__ w, const float* __restrict__ gn_b,
                                                          const float* __restrict__ mean, const float* __restrict__ scale,
                                                          float* __restrict__ out, _Float16* __restrict__ out_hi,
                                                          _Float16* __restrict__ out_lo, float* __restrict__ range_slot,
                                                          const int32_t* __restrict__ t0_clip) {
    __shared__ float xs[kFramesPerBlock * 5 + 8];
    const unsigned range_seen = SPLIT ? range_peek(range_slot) : 0u;  // read early: the load's latency hides under the taps
    float amax = 0.f;
    const int b = blockIdx.y;
    const long t0 = (long)blockIdx.x * kFramesPerBlock;
    const int nt = (int)((T0 - t0 < kFramesPerBlock) ? (T0 - t0) : kFramesPerBlock);
    const float* x = wav + (long)b * L + 5 * t0;
    const int nx = nt * 5 + 5;
    for (int i = threadIdx.x; i < nx; i += 256) xs[i] = x[i];

    const int c0 = threadIdx.x * 2;
    float w0[10], w1[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        w0[k] = w[c0 * 10 + k];
        w1[k] = w[(c0 + 1) * 10 + k];
    }
    const float mu0 = mean[b * kConvDim + c0], mu1 = mean[b * kConvDim + c0 + 1];
    const float sc0 = scale[b * kConvDim + c0], sc1 = scale[b * kConvDim + c0 + 1];
    const float be0 = gn_b[c0], be1 = gn_b[c0 + 1];
    __syncthreads();

    const long obase = ((long)b * T0 + t0) * kConvDim + c0;
    // packed forward: frames at and beyond the clip's own conv length do not exist in its reference batch; they are written as
    // zeros (the bias-free conv layers behind keep them zero), never read by a frame that does exist
    long left = t0_clip ? (long)t0_clip[b] - t0 : (long)nt;
    const int nreal = (int)(left < 0 ? 0 : (left < nt ? left : nt));
    for (int t = 0; t < nt; ++t) {
        float y0 = 0.f, y1 = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float xv = xs[5 * t + k];
            y0 = fmaf(w0[k], xv, y0);
            y1 = fmaf(w1[k], xv, y1);
        }
        // The ten taps stay scalar FMAs on purpose.  A version with the two channels packed (v_pk_fma_f32 on the taps) gave
        // sporadically different values in the first clips of a batch -- only while kernels of another stream were resident
        // on the same CUs (tools/race_probe.py: 6-11 of 12 trials; this form: 0 of 40).  hipcc had compiled those taps to
        // IN-PLACE packed FMAs whose destination pair is also the source read with an op_sel cross-selection
        // (`v_pk_fma_f32 v[32:33], v[4:5], v[32:33], v[42:43] op_sel:[0,1,0]`); no other kernel of this library contains that
        // form (tests/test_isa_patterns.py keeps it that way).
        const f32x2_t g_ = gelu_erf2(f32x2_t{fmaf(y0 - mu0, sc0, be0), fmaf(y1 - mu1, sc1, be1)});
        float r0 = t < nreal ? g_.x : 0.f, r1 = t < nreal ? g_.y : 0.f;
        if (SPLIT) {  // fp16 hi/lo planes: the A operand of the split-precision conv1 GEMM
            asm volatile("" : "+v"(r0), "+v"(r1));
            amax = fmaxf(amax, fmaxf(fabsf(r0), fabsf(r1)));
            h2_t hi, lo;
            hi[0] = (_Float16)r0; hi[1] = (_Float16)r1;
            lo[0] = (_Float16)(r0 - (float)hi[0]); lo[1] = (_Float16)(r1 - (float)hi[1]);
            *reinterpret_cast<h2_t*>(out_hi + obase + (long)t * kConvDim) = hi;
            *reinterpret_cast<h2_t*>(out_lo + obase + (long)t * kConvDim) = lo;
        } else {
            *reinterpret_cast<float2*>(out + obase + (long)t * kConvDim) = make_float2(r0, r1);
        }
    }
    // range tracking of the planes written (loco_kernels.h): max|x| of this workgroup's 64 frames x 512 channels, one atomic per
    // workgroup and only when the stage's word does not cover it yet
    if (SPLIT) range_commit_block(range_slot, amax, range_seen);
}

hipError_t launch_conv0_gn_gelu(const float* wav, int B, long L, const float* w, const float* gn_w, const float* gn_b,
                                float* out, void* scratch, float eps, hipStream_t s, void* out_hi, void* out_lo, float* range_slot,
                                const int32_t* t0_clip) {
    const long T0 = conv_out_len(L, 10, 5);
    if (B <= 0 || T0 <= 0) return hipErrorInvalidValue;
    double* partial = reinterpret_cast<double*>(scratch);
    double* total = partial + (size_t)B * kConv0Parts * kConv0Moments;
    float* mean = reinterpret_cast<float*>(total + (size_t)B * kConv0Moments);
    float* scale = mean + (size_t)B * kConvDim;
    hipLaunchKernelGGL(conv0_moments_kernel, dim3(kConv0Parts, B), dim3(256), 0, s, wav, L, T0, partial, t0_clip);
    hipLaunchKernelGGL(conv0_gn_coeffs_kernel, dim3(B), dim3(512), 0, s, partial, total, w, gn_w, T0, eps, mean, scale, t0_clip);
    const unsigned nblk = (unsigned)((T0 + kFramesPerBlock - 1) / kFramesPerBlock);
    if (out_hi)
        hipLaunchKernelGGL(conv0_apply_kernel<true>, dim3(nblk, B), dim3(256), 0, s, wav, L, T0, w, gn_b, mean, scale, out,
                           (_Float16*)out_hi, (_Float16*)out_lo, range_slot, t0_clip);
    else
        hipLaunchKernelGGL(conv0_apply_kernel<false>, dim3(nblk, B), dim3(256), 0, s, wav, L, T0, w, gn_b, mean, scale, out,
                           (_Float16*)nullptr, (_Float16*)nullptr, (float*)nullptr, t0_clip);
    return hipGetLastError();
}

}  // namespace loco

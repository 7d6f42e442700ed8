// Positional convolution embedding + sinusoidal positions of the speech prenet
// (HF modeling_speecht5.py:389-397 conv/SamePad/GELU, :555-556 residual add, :558-564 sinusoid add):
//
//   out[b,t,c] = h[b,t,c] + GELU(bias[c] + sum_{tap<128} sum_{i<48} h[b, t+tap-64, 48g+i] * w[48g+o, i, tap])
//                + sin_table[pos(b,t)][c],      c = 48g+o, zero padding outside [0,T), pos = t+2 if t < frames[b] else 1
//
// (Conv1d k=128, padding 64 yields T+1 frames; SamePadLayer drops the last, so frame t reads h[t-64 .. t+63].)
//
// Implicit GEMM per (clip, group, 128-frame tile): M = 128 frames, N = 48 output channels, K = 128 taps x 48.
// The A operand is Toeplitz -- row t+1 is row t shifted by one frame -- so the tile's whole input halo
// (128+127 frames x 48 channels, 53 KB) is staged in LDS ONCE and every tap reads it at a shifted row.
// N = 48 = 3 x 16 picks v_mfma_f32_16x16x4_f32 (same FLOP/clk as the 32x32x2 form).  Each lane quarter
// kq = lane>>4 owns input channels 12kq..12kq+11 so A and B fragments are 3 x 16-byte reads per tap.
// Weights (weight-norm already folded, laid out [group][tap][o][i]) are only 1.2 MB per group and are
// read straight from L2 into registers, one tap ahead of the MFMAs.
#include "loco_kernels.h"

namespace loco {

constexpr int PC_BM = 128;
constexpr int PC_ROWS = PC_BM + kPosK - 1;  // 255 input frames
constexpr int PC_LD = kPosCg + 4;           // 52 floats per LDS row

__global__ __launch_bounds__(256, 2) void pos_conv_kernel(const float* __restrict__ h, const float* __restrict__ wf,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ sin_table,
                                                          const int32_t* __restrict__ frames, float* __restrict__ out,
                                                          int T, const int32_t* __restrict__ rows_clip) {
    __shared__ __attribute__((aligned(16))) float xl[PC_ROWS * PC_LD];
    const int b = blockIdx.z, g = blockIdx.y;
    const int t0 = blockIdx.x * PC_BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, kq = lane >> 4;

    const float* hb = h + (long)b * T * kHidden + g * kPosCg;
    const int tin = rows_clip ? rows_clip[b] : T;  // packed forward: the clip's own reference batch ends here, zeros follow
    for (int f = tid; f < PC_ROWS * (kPosCg / 4); f += 256) {
        const int row = f / (kPosCg / 4), c4 = f % (kPosCg / 4);
        const int t = t0 - kPosK / 2 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < tin) v = *reinterpret_cast<const float4*>(hb + (long)t * kHidden + c4 * 4);
        *reinterpret_cast<float4*>(xl + row * PC_LD + c4 * 4) = v;
    }
    __syncthreads();

    f32x4 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* wg = wf + (long)g * kPosK * kPosCg * kPosCg + lr * kPosCg + 12 * kq;  // + tap*2304 + ns*16*48 + 4j
    const float* xa = xl + (wave * 32 + lr) * PC_LD + 12 * kq;                           // + (tap + 16ms)*PC_LD + 4j

    f32x4 bcur[3][3], bnext[3][3];
#pragma unroll
    for (int ns = 0; ns < 3; ++ns)
#pragma unroll
        for (int j = 0; j < 3; ++j) bcur[ns][j] = *reinterpret_cast<const f32x4*>(wg + ns * 16 * kPosCg + 4 * j);

    for (int tap = 0; tap < kPosK; ++tap) {
        const int tn = tap + 1 < kPosK ? tap + 1 : tap;
#pragma unroll
        for (int ns = 0; ns < 3; ++ns)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                bnext[ns][j] = *reinterpret_cast<const f32x4*>(wg + (long)tn * kPosCg * kPosCg + ns * 16 * kPosCg + 4 * j);
        f32x4 a[2][3];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int j = 0; j < 3; ++j) a[ms][j] = *reinterpret_cast<const f32x4*>(xa + (tap + 16 * ms) * PC_LD + 4 * j);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int ns = 0; ns < 3; ++ns)
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ms][j][e], bcur[ns][j][e], acc[ms][ns], 0, 0, 0);
#pragma unroll
        for (int ns = 0; ns < 3; ++ns)
#pragma unroll
            for (int j = 0; j < 3; ++j) bcur[ns][j] = bnext[ns][j];
    }

    // epilogue: acc[ms][ns][e] = conv at frame t0 + 32*wave + 16*ms + 4*kq + e, channel 48g + 16ns + lr
    const int nvalid = frames ? frames[b] : T;
#pragma unroll
    for (int ns = 0; ns < 3; ++ns) {
        const int col = 16 * ns + lr;
        const int c = g * kPosCg + col;
        const float bv = bias[c];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = wave * 32 + 16 * ms + 4 * kq + e;
                const int t = t0 + row;
                if (t < T) {
                    const int pos = t < nvalid ? t + 2 : 1;
                    const float hv = xl[(row + kPosK / 2) * PC_LD + col];
                    out[((long)b * T + t) * kHidden + c] =
                        hv + gelu_erf(acc[ms][ns][e] + bv) + sin_table[(long)pos * kHidden + c];
                }
            }
        }
    }
}

hipError_t launch_pos_conv(const float* h, const float* wf, const float* bias, const float* sin_table,
                           const int32_t* frames, float* out, int B, int T, hipStream_t s, const int32_t* rows_clip) {
    if (B <= 0 || T <= 0 || B > 65535) return hipErrorInvalidValue;
    dim3 grid((T + PC_BM - 1) / PC_BM, kPosGroups, B);
    hipLaunchKernelGGL(pos_conv_kernel, grid, dim3(256), 0, s, h, wf, bias, sin_table, frames, out, T, rows_clip);
    return hipGetLastError();
}

}  // namespace loco

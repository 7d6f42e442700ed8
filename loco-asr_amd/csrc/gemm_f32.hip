// fp32 GEMM on the CDNA4 matrix cores: C[m,n] = epi(sum_k A[m,k] * W[n,k] + bias[n]) (+ R[m,n]).
//
// Both operands are K-contiguous ("NT"), which is how every contraction of the SpeechT5 encoder path
// presents itself once activations are kept time-major / channels-last:
//   * nn.Linear            y = x W^T + b                (HF modeling_speecht5.py:502,867-870,994,1000)
//   * Conv1d layers 1..6   out[t,n] = sum_{tap,c} X[s*t+tap, c] W[n,c,tap]: with channels-last X the row of
//     output t is the CONTIGUOUS run X[s*t : s*t+k, :], so the conv is this GEMM with lda = s*C,
//     K = k*C and the weight re-laid tap-major (HF modeling:216-228)
//   * relative-position table  Qp = q_scaled pe_k^T  (compact form of HF modeling:939-945)
//
// Instruction: v_mfma_f32_32x32x2_f32 -- exact fp32 (bitwise an fmaf chain), 64 cycles/SIMD for 4096 FLOP,
// chip peak 157.3 TFLOP/s.  bf16/fp16 operands miss the 1e-3 embedding tolerance (BASELINE.md precision
// probe), gfx950 has no xf32, so fp32-in MFMA is the matrix-core path for this workload.
//
// Tiling: 128x128 block tile, BK = 32, 256 threads = 4 waves as 2(M) x 2(N), each wave 64x64 = 2x2 MFMA
// tiles (64 accumulator VGPRs + 64 for the blocked sums), 2 workgroups per CU.  A/W tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, round 4; global -> registers -> LDS with the loads two k-tiles ahead until then) into two unpadded,
// XOR-swizzled buffers with one barrier per k-tile; the DMA of tile kt + 1 flies under the MFMAs of tile kt.  Measured on the
// encoder's shapes (MI355X, QKV GEMM 47968x2304x768):
//   LDS fragments a quarter tile ahead, last quarter after the barrier          116 -> 128 TFLOP/s (with the register staging's
//                                                                               two-ahead global loads)
//   D = W_tile * A_tile^T so a lane owns 4 consecutive n       95 -> 120 TFLOP/s on the N=768 GEMMs
//   -> 16-byte epilogue stores (the dword store tail was issue-bound)
// The 32x32x2 instruction consumes k = {lane>>5}; lane-half h owns the contiguous k range [16h, 16h+16) of
// the tile so that fragments are read with ds_read_b128 (four 16-byte pieces per row and lane half, at the positions the swizzle
// gave them: conflict-free, see the kernel).
//
// Block -> tile map is XCD-aware: blocks b and b+8 share an XCD (and its private 4 MiB L2), so each XCD
// is given a contiguous run of tiles with n fastest -- every n-tile of one A row-panel is computed on
// the XCD that already holds that panel.
#include "loco_kernels.h"

namespace loco {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kGemmThreads = 256;

struct TileCoord {
    int z, mt, nt;
};

__device__ __forceinline__ TileCoord map_block(int bid, int nblk, int tiles_m, int tiles_n) {
    // bijective XCD remap (blocks congruent mod 8 share an XCD): XCD x gets tiles [start_x, start_x + cnt_x)
    const int q = nblk >> 3, r = nblk & 7;
    const int x = bid & 7, i = bid >> 3;
    const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    TileCoord c;
    c.nt = t % tiles_n;
    const int rest = t / tiles_n;
    c.mt = rest % tiles_m;
    c.z = rest / tiles_m;
    return c;
}

typedef __attribute__((address_space(3))) void* f32_lptr_t;

template <int EPI>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_f32_kernel(GemmArgs p, int tiles_m, int tiles_n, int nblk) {
    // two buffers of (BM + BN) rows x 32 floats, UNPADDED: tiles arrive by LDS-DMA (one wave instruction = 1 KiB = eight whole rows), the
    // 16-byte pieces of a row XOR-swizzled on the SOURCE address so that the fragment reads below stay conflict-free (see swz)
    __shared__ __attribute__((aligned(1024))) float lds[2][(BM + BN) * BK];

    const TileCoord tc = map_block(blockIdx.x, nblk, tiles_m, tiles_n);
    const int z1 = tc.z / p.nb2, z2 = tc.z % p.nb2;
    const float* __restrict__ A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const float* __restrict__ W = p.W;
    const long coff = z1 * p.sC1 + z2 * p.sC2;
    const int m0 = tc.mt * BM, n0 = tc.nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: kept in an SGPR (the DMA asm takes "s" operands)
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // ---- staging: global -> LDS directly (global_load_lds_dwordx4; round 4).  Until then the tiles went global -> registers -> LDS
    // with the loads two k-tiles ahead: 64 staging VGPRs, eight ds_write_b128 per lane and k-tile (13 cycles each on the store path)
    // and, once the blocked accumulation had filled the register file, spills.  A DMA instruction of wave w moves rows 32 w + 8 j ..
    // + 7 of a tile (j = 0 .. 3, A and W alike): lane -> row lane / 8, stored piece lane % 8, which holds source piece
    // (lane % 8) ^ swz(row), swz(row) = {row bit 4, row bit 3, row bit 1}.  A ds_read_b128 is served in four groups of 16 lanes
    // ({0-3, 12-15, 20-27}, ...): with one 128-byte row per lane a group must cover both row parities x 8 different pieces, which is
    // what those three row bits separate (the K tile of attention_f16x3.hip has the same shape and the same swizzle).
    // Per lane: a constant 32-bit byte offset per instruction (row clamped at M / N: rows past the end are computed on valid data
    // and never stored); the k advance goes into the wave-uniform base.
    const int drow = lane >> 3, dpos = lane & 7;
    unsigned oa[4], ow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 32 * wave + 8 * j + drow;
        const int swz = (((row >> 3) & 3) << 1) | ((row >> 1) & 1);
        const int ra = m0 + row < p.M ? row : p.M - 1 - m0;
        const int rw = n0 + row < p.N ? row : p.N - 1 - n0;
        oa[j] = 4u * (unsigned)(ra * p.lda) + 16u * (unsigned)(dpos ^ swz);
        ow[j] = 4u * (unsigned)(rw * p.ldw) + 16u * (unsigned)(dpos ^ swz);
    }
    const char* const Abase = reinterpret_cast<const char*>(A + (long)m0 * p.lda);
    const char* const Wbase = reinterpret_cast<const char*>(W + (long)n0 * p.ldw);
    const unsigned lds0 = (unsigned)(unsigned long)(f32_lptr_t)&lds[0][0];
    constexpr unsigned kBufBytes = (BM + BN) * BK * 4, kWOff = BM * BK * 4;
#define LOCO_DMA16(base_, voff_, ldsb_) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsb_), "v"(voff_), "s"(base_) : "memory")
#define LOCO_DMA_TILE(kt_, buf_)                                                                             \
    {                                                                                                        \
        const char* ab_ = Abase + (long)(kt_) * (BK * 4);                                                    \
        const char* wb_ = Wbase + (long)(kt_) * (BK * 4);                                                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                      \
            const unsigned d_ = lds0 + (unsigned)(buf_) * kBufBytes + (unsigned)(32 * wave + 8 * j) * (BK * 4); \
            LOCO_DMA16(ab_, oa[j], d_);                                                                      \
            LOCO_DMA16(wb_, ow[j], d_ + kWOff);                                                              \
        }                                                                                                    \
    }

    // BLOCKED ACCUMULATION (round 3).  One v_mfma_f32_32x32x2_f32 adds two products to its accumulator, so a dot product over K is a
    // chain of K/2 roundings -- 384 for K = 768, 1 536 for the second feed-forward GEMM -- where a CPU GEMM's vector lanes and
    // unrolled partial sums make it a tree.  On well-conditioned models nobody sees the difference; on the outlier-channel weight
    // family (golden g10) this mode was 2-4x torch's fp32 error per layer and 2e-4 of HF-in-float64 by layer 7 -- the "exact"
    // fallback less accurate than the default split mode it backs up.  Every four k-tiles the running block sum `acc` is folded into
    // `tot` (64 vector adds per lane) and restarted: chains of 64 roundings per block and K / 128 block sums.  g10: 2.0e-4 -> 1.0e-4
    // at the worst layer, 1.1e-4 -> 4.3e-5 at the last (HF's own fp32 pass: 5.9e-5).  (Folding every eight k-tiles instead gave
    // 1.14e-4 / 6.2e-5.)  The second accumulator set once filled the register file of the two-workgroups-per-CU form (spills, 0.80 ->
    // 0.69-0.74 of the fp32 MFMA peak); buffer-load staging (0.80 again) and then the LDS-DMA staging above removed that pressure.
    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; tot[i][j][e] = 0.f; }
#define LOCO_FLUSH_ACC(every_)                                                                  \
    if (((kt + 1) & ((every_) - 1)) == 0 && kt + 1 < nk) {                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                        \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                  \
                tot[i_][j_] += acc[i_][j_];                                                     \
                _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) acc[i_][j_][e_] = 0.f;        \
            }                                                                                   \
    }

    const int nk = p.K / BK;
    LOCO_DMA_TILE(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment offsets of this lane inside a buffer (floats): A rows wm*64 + {0,32} + r, W rows BM + wn*64 + {0,32} + r; the lane's
    // k range [16 h, 16 h + 16) is pieces 4 h .. 4 h + 3 of the row, read where the swizzle put them (swz(row) = swz(r): the row
    // offsets 64 wm + 32 i do not touch bits 4, 3, 1)
    const int fswz = (((r >> 3) & 3) << 1) | ((r >> 1) & 1);
    const int fa = (wm * 64 + r) * BK;
    const int fw = (BM + wn * 64 + r) * BK;
    int fq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) fq[q] = 4 * ((4 * h + q) ^ fswz);

#define LOCO_LOAD_FRAGS(buf, k4, A0, A1, B0, B1)                                        \
    A0 = *reinterpret_cast<const f32x4*>((buf) + fa + fq[k4]);                          \
    A1 = *reinterpret_cast<const f32x4*>((buf) + fa + 32 * BK + fq[k4]);                \
    B0 = *reinterpret_cast<const f32x4*>((buf) + fw + fq[k4]);                          \
    B1 = *reinterpret_cast<const f32x4*>((buf) + fw + 32 * BK + fq[k4]);
#define LOCO_MFMA16(A0, A1, B0, B1)                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                     \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(B0[e], A0[e], acc[0][0], 0, 0, 0); \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(B1[e], A0[e], acc[0][1], 0, 0, 0); \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(B0[e], A1[e], acc[1][0], 0, 0, 0); \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(B1[e], A1[e], acc[1][1], 0, 0, 0); \
    }

    // Software pipeline: the DMA of tile kt + 1 is issued at the top of k-tile kt into the buffer whose last reads were issued before
    // the barrier that ended k-tile kt - 1; fragments run a quarter tile ahead of the MFMAs, and the LAST quarter of tile kt is
    // computed AFTER the barrier that publishes tile kt + 1 -- its 1024 MFMA cycles cover the barrier skew and the LDS latency of the
    // next tile's first fragments, so a wave never waits on LDS with an idle matrix pipe.
    int cur = 0;
    f32x4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
    LOCO_LOAD_FRAGS(lds[0], 0, xa0, xa1, xb0, xb1)
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) LOCO_DMA_TILE(kt + 1, cur ^ 1)
        const float* lb = lds[cur];
        LOCO_LOAD_FRAGS(lb, 1, ya0, ya1, yb0, yb1)
        __builtin_amdgcn_sched_barrier(0);
        LOCO_MFMA16(xa0, xa1, xb0, xb1)
        __builtin_amdgcn_sched_barrier(0);
        LOCO_LOAD_FRAGS(lb, 2, xa0, xa1, xb0, xb1)
        __builtin_amdgcn_sched_barrier(0);
        LOCO_MFMA16(ya0, ya1, yb0, yb1)
        __builtin_amdgcn_sched_barrier(0);
        LOCO_LOAD_FRAGS(lb, 3, ya0, ya1, yb0, yb1)
        __builtin_amdgcn_sched_barrier(0);
        LOCO_MFMA16(xa0, xa1, xb0, xb1)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt + 1 have landed (the compiler does not count asm DMAs)
        __syncthreads();
        cur ^= 1;
        if (more) { LOCO_LOAD_FRAGS(lds[cur], 0, xa0, xa1, xb0, xb1) }
        __builtin_amdgcn_sched_barrier(0);
        LOCO_MFMA16(ya0, ya1, yb0, yb1)
        __builtin_amdgcn_sched_barrier(0);
        LOCO_FLUSH_ACC(4)
    }
#undef LOCO_LOAD_FRAGS
#undef LOCO_MFMA16
#undef LOCO_FLUSH_ACC
#undef LOCO_DMA_TILE
#undef LOCO_DMA16

    // epilogue.  The MFMAs were issued as D = W_tile * A_tile^T, so acc[i][j][e] is
    //   C[m = m0 + wm*64 + 32i + r][n = n0 + wn*64 + 32j + 8*(e>>2) + 4h + (e&3)]:
    // each lane owns 4 consecutive n per register quad -> 16-byte stores (16 per sub-tile pair instead of 64
    // dword stores; the store tail of a 128x128 tile is issue-bound, not bandwidth-bound).
    float* __restrict__ C = p.C + coff;
    const float* __restrict__ R = (EPI == kEpiResidual) ? p.R + coff : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * h;
                if (n < p.N) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = tot[i][j][4 * g + e] + acc[i][j][4 * g + e];
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (EPI == kEpiGelu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    }
                    if (EPI == kEpiResidual) v += *reinterpret_cast<const f32x4*>(R + (long)m * p.ldr + n);
                    *reinterpret_cast<f32x4*>(C + (long)m * p.ldc + n) = v;
                }
            }
        }
    }
}

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % BK != 0) return hipErrorInvalidValue;
    if ((a.lda | a.ldw | a.sA1 | a.sA2) & 3) return hipErrorInvalidValue;  // float4 staging
    if ((a.N | a.ldc | a.sC1 | a.sC2) & 3) return hipErrorInvalidValue;    // float4 epilogue
    if (a.epilogue == kEpiResidual && (a.ldr & 3)) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.W)) & 15) return hipErrorInvalidValue;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const long nblk = (long)tiles_m * tiles_n * a.nb1 * a.nb2;
    if (nblk <= 0 || nblk > 0x7fffffffL) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblk), block(kGemmThreads);
    // One kernel for every epilogue.  (Until round 4 the GELU GEMMs -- FFN1 and conv layers 1-6, K up to 1 536 -- ran on a BK = 16,
    // three-workgroups-per-CU variant that was 2-3 % faster and had no registers for the blocked accumulation above: chains of 768
    // roundings in exactly the GEMMs with the longest K.  This mode is the accuracy fallback; the variant is gone.)
    switch (a.epilogue) {
        case kEpiNone:
            hipLaunchKernelGGL((gemm_f32_kernel<kEpiNone>), grid, block, 0, s, a, tiles_m, tiles_n, (int)nblk);
            break;
        case kEpiGelu:
            hipLaunchKernelGGL((gemm_f32_kernel<kEpiGelu>), grid, block, 0, s, a, tiles_m, tiles_n, (int)nblk);
            break;
        case kEpiResidual:
            if (!a.R) return hipErrorInvalidValue;
            hipLaunchKernelGGL((gemm_f32_kernel<kEpiResidual>), grid, block, 0, s, a, tiles_m, tiles_n, (int)nblk);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace loco

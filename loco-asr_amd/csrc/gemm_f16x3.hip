// Split-precision GEMM: fp32-class accuracy at the fp16 matrix-core rate.
//
//   C = epi(A W^T + bias) with  A = A_hi + A_lo,  W = W_hi + W_lo  (each part fp16, hi = fp16(x), lo = fp16(x - hi))
//   A W^T ~= A_hi W_hi^T + A_lo W_hi^T + A_hi W_lo^T          (the dropped lo*lo term is ~2^-22 relative)
//
// Every partial product of two fp16 values is exact in the MFMA's fp32 accumulator, so the only errors are the
// 22-bit hi+lo representation and the dropped term: the whole encoder lands at ~1e-6 relative L2 of an fp64
// evaluation, like exact fp32, against 1.6e-3 for plain fp16 (BASELINE.md probe).
// Three v_mfma_f32_16x16x32_f16 (16 cycles, 16 384 FLOP each) replace four v_mfma_f32_32x32x2_f32 (64 cycles,
// 4 096 FLOP each): 5.3x fewer matrix-pipe cycles per algorithmic FLOP.
//
// Operands arrive ALREADY split -- weights once at load time, activations by the epilogue of the kernel that
// produced them (same bytes as fp32: 2+2) -- so this kernel moves exactly the bytes of the fp32 GEMM and spends
// no VALU on conversion.  One kernel template (gemm_f16x3_dma_kernel, below) covers every shape: 256x256, 192x256, 256x128 and
// 128x128 tiles on two LDS rings (A, W) filled by LDS-DMA, 64x64 per wave, D = W_tile * A_tile^T orientation, a software-
// pipelined k-loop with one barrier per k-tile; small problems add split-K with a fixed-order reduction (launch_gemm_split).
#include <cstdlib>

#include "loco_kernels.h"

#ifndef LOCO_GEMM_HACK
#define LOCO_GEMM_HACK 0  // timing-only diagnostic builds, see DMA16 below
#endif

namespace loco {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// Shared epilogue.  v = 4 consecutive output columns n..n+3 of row m (already bias-added).
template <int EPI, bool OUT_SPLIT>
__device__ __forceinline__ void split_gemm_store(const GemmSplitArgs& p, f32x4 v, long coff, int m, int n, float& amax, int z1 = 0,
                                                 int z2 = 0) {
    if (EPI == kEpiGelu) {
#pragma unroll
        for (int e = 0; e < 4; e += 2) { const f32x2_t g_ = gelu_erf2(f32x2_t{v[e], v[e + 1]}); v[e] = g_.x; v[e + 1] = g_.y; }
    }
    if (EPI == kEpiResidual) {
        const long ro = coff + (long)m * p.ldr + n;
        if (p.Rhi) {
            const h4 rh = *reinterpret_cast<const h4*>(p.Rhi + ro), rl = *reinterpret_cast<const h4*>(p.Rlo + ro);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rh[e] + (float)rl[e];
        } else {
            v += *reinterpret_cast<const f32x4*>(p.R + ro);
        }
    }
    if (EPI == kEpiPosConv) {  // hidden + GELU(conv + bias) + sinusoid (HF modeling:555-564); z1 = clip, z2 = group, m = frame
#pragma unroll
        for (int e = 0; e < 4; e += 2) { const f32x2_t g_ = gelu_erf2(f32x2_t{v[e], v[e + 1]}); v[e] = g_.x; v[e + 1] = g_.y; }
        int nvalid = p.frames ? p.frames[z1] : p.T;
        if (nvalid <= 0 || nvalid > p.T) nvalid = p.T;
        const int pos = m < nvalid ? m + 2 : 1;
        v += *reinterpret_cast<const f32x4*>(p.R + coff + (long)m * p.ldr + n);
        v += *reinterpret_cast<const f32x4*>(p.sin_table + (long)pos * kHidden + z2 * kPosCg + n);
        *reinterpret_cast<f32x4*>(p.C + coff + (long)m * p.ldc + n) = v;
        return;
    }
    if (!OUT_SPLIT && EPI != kEpiQkvScatter) {
        *reinterpret_cast<f32x4*>(p.C + coff + (long)m * p.ldc + n) = v;
        return;
    }
    // hi and lo must derive from the SAME rounded fp32 value (left to itself hipcc folds the producing multiply into
    // v_fma_mix*_f16 for lo but converts hi from the fp32-rounded product: one fp16 ulp of hi on ties); the explicit
    // instructions of split_f16_2pairs read the value registers, which settles it.
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));  // range tracking (loco_kernels.h)
    u32x2_t hu, lu;
    {
        unsigned h0, l0, h1, l1;
        split_f16_2pairs(v[0], v[1], v[2], v[3], h0, l0, h1, l1);
        hu = u32x2_t{h0, h1};
        lu = u32x2_t{l0, l1};
    }
    const h4 hi = __builtin_bit_cast(h4, hu), lo = __builtin_bit_cast(h4, lu);
    if (EPI == kEpiQkvScatter) {
        {  // q | k | v: the three plane pairs are qkv_stride apart (launch_gemm_split checks it), v row-major like q and k
            const int third = n < kHidden ? 0 : (n < 2 * kHidden ? 1 : 2);
            const long o = third * p.qkv_stride + (long)m * kHidden + (n - third * kHidden);
            *reinterpret_cast<h4*>(p.Chi + o) = hi;
            *reinterpret_cast<h4*>(p.Clo + o) = lo;
        }
    } else {
        const long o = coff + (long)m * p.ldc + n;
        *reinterpret_cast<h4*>(p.Chi + o) = hi;
        *reinterpret_cast<h4*>(p.Clo + o) = lo;
    }
}

// 16 consecutive output columns n..n+15 of row m (v[j] = columns n+4j..n+4j+3, bias added): the LDS-DMA kernel hands each
// lane such a run (see the W-row permutation there), so planes are written as 16-byte pieces and a row's four lanes
// complete whole 128-byte lines instead of 8-byte fragments.
template <int EPI, bool OUT_SPLIT>
__device__ __forceinline__ void split_gemm_store16(const GemmSplitArgs& p, f32x4 (&v)[4], long coff, int m, int n, float& amax, int z1,
                                                   int z2) {
    constexpr bool wide_epi = EPI == kEpiNone || EPI == kEpiGelu || EPI == kEpiResidual || EPI == kEpiQkvScatter;
    if (!wide_epi || n + 16 > p.N) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + 4 * j < p.N) split_gemm_store<EPI, OUT_SPLIT>(p, v[j], coff, m, n + 4 * j, amax, z1, z2);
        return;
    }
    if (EPI == kEpiGelu) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; e += 2) { const f32x2_t g_ = gelu_erf2(f32x2_t{v[j][e], v[j][e + 1]}); v[j][e] = g_.x; v[j][e + 1] = g_.y; }
    }
    if (EPI == kEpiResidual) {
        const long ro = coff + (long)m * p.ldr + n;
        if (p.Rhi) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const h8 rh = *reinterpret_cast<const h8*>(p.Rhi + ro + 8 * jj), rl = *reinterpret_cast<const h8*>(p.Rlo + ro + 8 * jj);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[2 * jj + (e >> 2)][e & 3] += (float)rh[e] + (float)rl[e];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += *reinterpret_cast<const f32x4*>(p.R + ro + 4 * j);
        }
    }
    if (!OUT_SPLIT && EPI != kEpiQkvScatter) {
        float* cp = p.C + coff + (long)m * p.ldc + n;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (LOCO_GEMM_HACK != 4 || p.M <= 0) *reinterpret_cast<f32x4*>(cp + 4 * j) = v[j];
        return;
    }
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    h8 hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[j][0]), fabsf(v[j][1]))), fmaxf(fabsf(v[j][2]), fabsf(v[j][3])));
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {  // v[2jj], v[2jj+1] -> one h8 of each plane
        u32x4_t hu, lu;
        unsigned h0, l0, h1, l1;
        split_f16_2pairs(v[2 * jj][0], v[2 * jj][1], v[2 * jj][2], v[2 * jj][3], h0, l0, h1, l1);
        hu[0] = h0; hu[1] = h1; lu[0] = l0; lu[1] = l1;
        split_f16_2pairs(v[2 * jj + 1][0], v[2 * jj + 1][1], v[2 * jj + 1][2], v[2 * jj + 1][3], h0, l0, h1, l1);
        hu[2] = h0; hu[3] = h1; lu[2] = l0; lu[3] = l1;
        hi[jj] = __builtin_bit_cast(h8, hu);
        lo[jj] = __builtin_bit_cast(h8, lu);
    }
    _Float16 *dh, *dl;
    if (EPI == kEpiQkvScatter) {
        const int third = n < kHidden ? 0 : (n < 2 * kHidden ? 1 : 2);  // a run of 16 columns never straddles q | k | v (768 = 48 x 16)
        const long o = third * p.qkv_stride + (long)m * kHidden + (n - third * kHidden);  // the plane pairs are qkv_stride apart
        dh = p.Chi + o;
        dl = p.Clo + o;
    } else {
        const long o = coff + (long)m * p.ldc + n;
        dh = p.Chi + o;
        dl = p.Clo + o;
    }
    if (LOCO_GEMM_HACK == 4 && p.M > 0) return;
    *reinterpret_cast<h8*>(dh) = hi[0];
    *reinterpret_cast<h8*>(dh + 8) = hi[1];
    *reinterpret_cast<h8*>(dl) = lo[0];
    *reinterpret_cast<h8*>(dl + 8) = lo[1];
}

constexpr int SBK = 32;

// Diagnostic build only (-DLOCO_GEMM_STAMPS, tools/gemm_stamps.py): wave 0 of every workgroup records where its cycles go --
// prologue / main loop / of which parked at the k-tile wait + barrier / epilogue / store drain -- into a buffer of its own.
// No stamp exists in the shipped library.
#ifdef LOCO_GEMM_STAMPS
__device__ unsigned long long* g_gemm_stamps = nullptr;
#define GEMM_STAMP(t) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define GEMM_STAMP(t) {}
#endif

// ---------------------------------------------------------------------------------------------------------------
// The GEMM kernel.  WM x WN waves, each computing 64 x 64 of a (64 WM) x (64 WN) tile as 4 x NJ accumulators of
// v_mfma_f32_16x16x32_f16, ONE workgroup per CU unless WPS says otherwise.  A and W k-tiles (32 deep, hi and lo planes) stream
// global -> LDS with global_load_lds_dwordx4 (no staging VGPRs, no ds_write) into TWO rings: AST slots for A, WST for W.  The A
// operand is an activation that comes from HBM and is what a k-tile waits for (tools/gemm_stamps.py: with A served from L2 a
// 256x256 k-tile takes 3 500 cycles, from HBM 3 800 - 4 100, the MFMAs alone 3 072), the weights are re-read by every row of
// tiles and sit in L2 / the Infinity Cache: so the 160 KiB of LDS go to THREE A slots and TWO W slots for the 256x256 tile
// (3 x 32 + 2 x 32 KiB) -- an A k-tile has two k-tiles of time to arrive, a W k-tile one.
// The DMA writes 1 KiB contiguously per wave-instruction (16 rows x 64 B of one plane), so rows cannot be padded; bank
// conflicts are removed by an XOR swizzle instead, applied on the per-lane SOURCE address and on the fragment reads alike.
// Tiles are retired with counted s_waitcnt vmcnt(N) (the DMAs of the youngest A k-tile stay in flight) and a raw s_barrier.
//
// Tile forms: 4x4 = 256x256 / 16 waves (half the L2 -> LDS bytes per FLOP of 128x128); 3x4 = 192x256 / 12 waves (N = 768 on
// 47 968 rows: 2.93 rounds of 256 workgroups instead of 2.2 paid as 3); 4x2 = 256x128 / 8 waves; 2x2 = 128x128 / 4 waves
// (small M; two per CU for K <= 128); 8x1 with NJ = 3 = the grouped positional conv (N = 48 per group).
// NJ: 16-column sub-tiles each wave computes.  WPS: waves per SIMD the register allocation must leave room for (0 = one
// workgroup per CU).  TERMS: 3 = A_hi W_hi + A_lo W_hi + A_hi W_lo (fp32 class, the default); 2 = the W_lo term dropped, i.e.
// the weights rounded to fp16 after their per-tensor power-of-two scale (opt-in precision mode "f16x2").
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) const h8* lds_h8p;

// One LDS-DMA instruction in its scalar-base form, global_load_lds_dwordx4 voffset, sbase: written as asm because hipcc turns
// "uniform base + zero-extended lane offset" back into one 64-bit VGPR address per plane (8 VGPRs and, at the 128-register budget of
// the 16-wave form, a spill with a vmcnt(0) reload between the DMAs).  M0 = LDS byte address of the piece (wave-uniform); one wait
// state between the SALU write of M0 and the DMA that reads it.  (M0 is a reserved register to hipcc, which rejects it as a clobber;
// nothing else in this kernel uses it -- gfx9 DS instructions do not -- and tests/test_isa_patterns.py checks that.)
// LOCO_GEMM_HACK (timing-only diagnostic builds, WRONG results; tools/ab/build_variant.sh): 1 = no LDS-DMA is issued (what does the
// L2 -> LDS traffic cost?), 2 = row groups 1 and 3 re-use the A fragments of 0 and 2 (a quarter of the LDS reads gone), 4 = no
// epilogue stores, 6 = only waves 0-3 issue LDS-DMA (with tools/gemm_stamps.py).
#if LOCO_GEMM_HACK == 1
#define DMA16(base_, voff_, ldsb_) asm volatile("" :: "s"(ldsb_), "v"(voff_), "s"(base_) : "memory")
#elif LOCO_GEMM_HACK == 6  // only waves 0-3 (one per SIMD) issue their DMA pieces: do the four waves of a SIMD collide in DMA issue?
#define DMA16(base_, voff_, ldsb_)                                                                                                  \
    if (wave < 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsb_), "v"(voff_), "s"(base_) : "memory")
#else
#define DMA16(base_, voff_, ldsb_) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsb_), "v"(voff_), "s"(base_) : "memory")
#endif
#define VMCNT_LGKM0(n_)                                                                              \
    {                                                                                                \
        static_assert((n_) >= 0 && (n_) <= 63, "vmcnt is a 6-bit field");                            \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n_) : "memory");                         \
    }

template <int EPI, bool OUT_SPLIT, int WM, int WN, int AST, int WST, int NJ = 4, int WPS = 0, int TERMS = 3, bool PERSIST = false>
__global__ __launch_bounds__(64 * WM * WN, WPS ? WPS : (WM * WN) / 4) void gemm_f16x3_dma_kernel(GemmSplitArgs p, int tiles_m,
                                                                                                  int tiles_n, int nblk, int wg_step,
                                                                                                  int col_group) {
    constexpr int DBM = 64 * WM, DBN = 64 * WN, NW_ = WM * WN;
    constexpr int DPA = DBM * SBK, DPW = DBN * SBK;          // halves per A / W plane of one k-tile
    constexpr int ABUF = 2 * DPA, WBUF = 2 * DPW;            // halves per ring slot (hi plane, lo plane)
    constexpr int NDA = DBM / 16 / NW_;                      // 16-row DMA pieces per wave per A plane
    constexpr int NDW = (DBN / 16 + NW_ - 1) / NW_;          // ... per W plane (the last waves may have none)
    constexpr int WPIECES = DBN / 16;
    constexpr int NA = 2 * NDA, NWP = (TERMS == 3 ? 2 : 1) * NDW;  // DMA instructions per wave per k-tile: A side, W side
    static_assert(NDA >= 1 && DBM % (16 * NW_) == 0, "tile too small for the wave count");
    static_assert((AST == WST && AST >= 2 && AST <= 5) || (AST == 3 && WST == 2), "ring depths: equal rings of 2-5 slots, or 3 A + 2 W");
    static_assert(WPIECES % NW_ == 0 || (WST == 2 && AST == 3) || AST == 2, "uneven W pieces: no wait may leave W DMAs in flight");
    static_assert(TERMS == 3 || TERMS == 2, "terms");
    static_assert((size_t)(AST * ABUF + WST * WBUF) * 2 <= 160 * 1024, "LDS");
    static_assert(2 * (DPA + 16 * 3 * SBK) < 65536 && 2 * (DPW + 16 * 3 * SBK) < 65536, "fragment offsets are ds_read immediates");
    // PERMW: the 16 W rows (output columns) fed to MFMA sub-tile j are {16 q + 4 j + e}, q, e = 0..3, instead of 16 j .. 16 j + 15.
    // The accumulator of lane quad q4 then holds columns 16 q4 + 4 j + e: over j = 0..3 one lane owns 16 CONSECUTIVE
    // columns of its row, which the epilogue writes as 16-byte pieces (split_gemm_store16).  Same FLOPs and LDS bytes;
    // only the fragment row of a lane and the W swizzle (piece ^ f((row >> 4) & 3), f = 0,2,3,1) change.  Used for fp16
    // plane outputs (-4..9 % time on FFN1 / conv layers); fp32 outputs keep the plain mapping, whose 16-byte pieces of
    // four neighbouring lanes already form 64-byte runs (the permuted form measured 4..15 % slower there).
    constexpr bool PERMW = NJ == 4 && (OUT_SPLIT || EPI == kEpiQkvScatter);
    __shared__ __attribute__((aligned(16))) _Float16 lds[AST * ABUF + WST * WBUF];  // A ring, then W ring

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int drow = lane >> 2, dpos = lane & 3;
    const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)lds;  // LDS byte address of the rings
    const int nk = p.K / SBK;

    // XCD-aware work map: workgroups whose ids are congruent mod 8 share an XCD (and its private L2); each XCD gets a contiguous
    // run of output tiles, column tile fastest, so that the tiles in flight on one L2 share their A rows.  PERSISTENT form
    // (template flag PERSIST and wg_step > 0: the grid is one workgroup per CU slot): a workgroup walks its XCD's run with stride wg_step, and the DMA
    // stream simply runs on into the next output tile -- its first k-tiles are fetched while this tile's last ones are multiplied
    // and its epilogue runs, so no tile but the first waits for a cold prologue (6-9 % of a tile's life at K = 768 / 1536).
    const int xcd = blockIdx.x & 7;
    int tile_i = blockIdx.x >> 3;  // index inside the XCD's run
    int run_start, run_len;
    {
        const int q = nblk >> 3, rr = nblk & 7;
        run_start = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
        run_len = q + (xcd < rr ? 1 : 0);
    }
    // DMA sources of an output tile: this wave fills NDA 16-row pieces of each A plane and NDW of each W plane; lane -> row
    // lane/4, stored position lane%4, source piece pos ^ swz(row).  Each source = a workgroup-uniform 64-bit base (plane pointer
    // + tile origin + k offset: scalar registers) + a per-lane 32-bit byte offset inside the tile: one VGPR per piece.
    struct Tile {
        const char *bAh, *bAl, *bWh, *bWl;
        unsigned va[NDA], vw[NDW];
        int m0, n0, z1, z2, kt0;
        long coff;
    };
    auto decode = [&](int i, Tile& t_) {
        // Tile order inside a batch item: column tiles in groups of col_group, row tile next, group last.  The 32 workgroups of
        // an XCD that run side by side then cover (32 / col_group) row tiles x col_group column tiles instead of a few rows x ALL
        // columns: every A k-slice fetched into the L2 serves col_group tiles and every W k-slice 32 / col_group -- for square
        // tiles the L2-miss bytes A (tiles_n / c) + W (tiles_m c / 32) are least at c = sqrt(32) (FFN1: 810 -> 626 MB per launch).
        const int t = run_start + i;
        const int per_z = tiles_m * tiles_n;
        const int z = t / per_z;
        const int tz = t - z * per_z;
        const int full = tiles_n / col_group;  // complete column groups
        int g = tz / (tiles_m * col_group), rem, width;
        if (g < full) {
            rem = tz - g * tiles_m * col_group;
            width = col_group;
        } else {
            g = full;
            rem = tz - full * tiles_m * col_group;
            width = tiles_n - full * col_group;
        }
        const int mt = rem / width;
        const int nt = g * col_group + (rem - mt * width);
        t_.z1 = z / p.nb2;
        t_.z2 = z % p.nb2;
        t_.kt0 = t_.z2 * p.kt_per_z2;
        const int z1o = t_.z1 / p.z1_inner, z1i = t_.z1 - z1o * p.z1_inner;
        const long aoff = z1o * p.sA1 + z1i * p.sA1i + t_.z2 * p.sA2;
        const long woff = z1i * p.sW1i + t_.z2 * p.sW2;
        t_.coff = t_.z1 * p.sC1 + t_.z2 * p.sC2;
        t_.m0 = mt * DBM;
        t_.n0 = nt * DBN;
#pragma unroll
        for (int u = 0; u < NDA; ++u) {
            const int row = 16 * (NDA * wave + u) + drow;
            int ra = t_.m0 + row;
            ra = (ra < p.M ? ra : p.M - 1) - t_.m0;
            t_.va[u] = 2u * (unsigned)(ra * (int)p.lda + 8 * (dpos ^ (3 * ((row >> 2) & 1))));
        }
#pragma unroll
        for (int u = 0; u < NDW; ++u) {
            const int row = 16 * (NDW * wave + u) + drow;
            int rw = t_.n0 + row;
            rw = (rw < p.N ? rw : p.N - 1) - t_.n0;
            t_.vw[u] = 2u * (unsigned)(rw * (int)p.ldw + 8 * (dpos ^ (PERMW ? ((0x78 >> (2 * ((row >> 4) & 3))) & 3) : 3 * ((row >> 2) & 1))));
        }
        t_.bAh = reinterpret_cast<const char*>(p.Ahi + aoff + (long)t_.m0 * p.lda);
        t_.bAl = reinterpret_cast<const char*>(p.Alo + aoff + (long)t_.m0 * p.lda);
        t_.bWh = reinterpret_cast<const char*>(p.Whi + woff + (long)t_.n0 * p.ldw);
        t_.bWl = reinterpret_cast<const char*>(p.Wlo + woff + (long)t_.n0 * p.ldw);
    };
    Tile cur, nxt;
    decode(tile_i, cur);
    nxt = cur;

    // k-tile t of tile T_'s A / W operand into ring slot sl_; piece q_ of the wave's NA / NWP.  t is clamped to the last k-tile:
    // where a stream has no successor (the last output tile of a workgroup, or K shorter than the ring) the surplus DMAs re-fetch
    // a k-tile into a slot nobody reads again; the vmcnt(0) at the end of the kernel retires them -- no DMA may land after the
    // workgroup has given up its LDS.
    // byte offset of A's k-tile t: plain, or (GemmSplitArgs::ktaps > 1) the walk (64-channel block, tap slot, 32-channel half) with
    // tap slots 0, 2, 1 -- loco_kernels.h.  Branch-free; ktaps is 1, 2 or 3 (checked by the launcher); t < 32768.
    const int ktaps = p.ktaps, kchan2 = 2 * (p.kchan ? p.kchan : p.K / (p.ktaps > 0 ? p.ktaps : 1));
    auto a_koff = [&](int t) -> long {
        const int q = t >> 1, half = t & 1;
        const int cbp = ktaps == 3 ? (int)(((unsigned)q * 0xAAABu) >> 17) : (ktaps == 2 ? q >> 1 : q);
        const int slot = q - cbp * ktaps;
        const int tap = ktaps == 3 ? ((0x18 >> (2 * slot)) & 3) : slot;  // slots 0, 1, 2 -> taps 0, 2, 1
        const long off = (long)tap * kchan2 + (long)(2 * cbp + half) * (2 * SBK);
        return ktaps == 1 ? (long)t * (2 * SBK) : off;
    };
#define DMA_A(q_, T_, t_, sl_)                                                                                              \
    {                                                                                                                       \
        const int tt_ = (t_) < nk ? (t_) : nk - 1;                                                                          \
        const unsigned d_ = lds0 + 2u * (unsigned)((sl_) * ABUF + ((q_) & 1) * DPA + 16 * (NDA * wave + ((q_) >> 1)) * SBK); \
        DMA16((((q_) & 1) ? (T_).bAl : (T_).bAh) + a_koff((T_).kt0 + tt_), (T_).va[(q_) >> 1], d_);                         \
    }
#define DMA_W(q_, T_, t_, sl_)                                                                                              \
    {                                                                                                                       \
        const int tt_ = (t_) < nk ? (t_) : nk - 1;                                                                          \
        const int u_ = TERMS == 3 ? (q_) >> 1 : (q_);                                                                       \
        const int pl_ = TERMS == 3 ? (q_) & 1 : 0;                                                                          \
        if (WPIECES % NW_ == 0 || NDW * wave + u_ < WPIECES) {                                                              \
            const unsigned d_ = lds0 + 2u * (unsigned)(AST * ABUF + (sl_) * WBUF + pl_ * DPW + 16 * (NDW * wave + u_) * SBK); \
            DMA16((pl_ ? (T_).bWl : (T_).bWh) + (long)tt_ * (2 * SBK), (T_).vw[u_], d_);                                    \
        }                                                                                                                   \
    }

    [[maybe_unused]] unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, rt0 = 0, rt1 = 0;
#ifdef LOCO_GEMM_STAMPS
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) :: "memory");
#endif
    GEMM_STAMP(st0)

    // Fragment reads: lane (r16, q4) reads row r16 of a 16-row sub-tile, 16-byte piece q4 (stored at q4 ^ swizzle(row)); one
    // k-step per k-tile.  Kept as two LDS byte addresses per lane (A side, W side) that step from ring slot to ring slot by a
    // scalar delta; every fragment of a slot is an immediate offset from them.
    const int swz = 3 * ((r16 >> 2) & 1);
    const int fa = (wm * 64 + r16) * SBK + 8 * (q4 ^ swz);
    const int fw = PERMW ? (wn * 64 + 16 * (r16 >> 2) + (r16 & 3)) * SBK + 8 * (q4 ^ ((0x78 >> (2 * (r16 >> 2))) & 3))
                         : (wn * 64 + r16) * SBK + 8 * (q4 ^ swz);
    unsigned a_ad = lds0 + 2u * (unsigned)fa;
    unsigned w_ad = lds0 + 2u * (unsigned)(AST * ABUF + fw);
#define LDS_H8(ad_, halves_) (*(lds_h8p)(unsigned long)((ad_) + 2u * (unsigned)(halves_)))
#define RD_A(i_, s_)                                                  \
    if (LOCO_GEMM_HACK != 2 || ((i_) & 1) == 0) {                     \
        ah[s_] = LDS_H8(a_ad, 16 * (i_) * SBK);                       \
        al[s_] = LDS_H8(a_ad, DPA + 16 * (i_) * SBK);                 \
    } else {                                                          \
        ah[s_] = ah[1 - (s_)];                                        \
        al[s_] = al[1 - (s_)];                                        \
    }
#define RD_W(j_)                                                                         \
    {                                                                                    \
        wh[j_] = LDS_H8(w_ad, (PERMW ? 4 : 16) * (j_) * SBK);                            \
        if (TERMS == 3) wl[j_] = LDS_H8(w_ad, DPW + (PERMW ? 4 : 16) * (j_) * SBK);      \
    }
#define MM(i_, s_, j_)                                                                                                    \
    {                                                                                                                     \
        if (TERMS == 3) acc16[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j_], ah[s_], acc16[i_][j_], 0, 0, 0);   \
        acc16[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j_], al[s_], acc16[i_][j_], 0, 0, 0);                   \
        acc16[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j_], ah[s_], acc16[i_][j_], 0, 0, 0);                   \
    }
#define SB() __builtin_amdgcn_sched_barrier(0);
    // ring slots of the k-tile being multiplied, and the byte steps to the next slot
    int sa = 0, sw = 0;
#define A_STEP() (unsigned)(sa + 1 == AST ? -(AST - 1) * 2 * ABUF : 2 * ABUF)
#define W_STEP() (unsigned)(sw + 1 == WST ? -(WST - 1) * 2 * WBUF : 2 * WBUF)

    // ---- software-pipelined k-loop: ONE barrier per k-tile, placed in the MIDDLE of the tile's MFMAs ---------------------------
    // A wave's fragments of k-tile kt are all in registers (or on their way) well before the tile's MFMAs end, so the barrier X_kt
    // "every wave has read the slots of k-tile kt and every wave's DMA pieces of k-tile kt+1 have landed" can sit behind the reads
    // of the last of the four row groups.  Behind it the slots of k-tile kt are re-filled (W k-tile kt + WST, A k-tile kt + AST)
    // and -- no further barrier needed -- the fragments of k-tile kt+1 are read while the last row group's MFMAs run: A row
    // group 0 into the free A slot, each W sub-tile into the registers of the sub-tile that has just been multiplied for the last
    // time.  The matrix pipe has MFMAs queued on both sides of the barrier; a loop that stops at its barrier with empty
    // registers pays DMA issue + LDS latency (~430 of ~3 500 cycles per k-tile) before the next k-tile's first MFMA.
    // The DMA instructions are spread over the MFMA blocks behind the barrier: issued back to back they keep every wave of
    // the SIMD out of the matrix pipe at the same time (an LDS-DMA instruction holds its wave for 60-180 cycles).
    constexpr int JH = (NJ + 1) / 2;   // W sub-tiles of row group 2 multiplied BEFORE the barrier (all of them there: +1.5...4 % cycles per k-tile)
    constexpr int NG = (NJ + 1) / 2;   // pairs of W sub-tiles
    constexpr int NBLK = 1 + NG;       // MFMA blocks behind the barrier: the rest of row group 2, then row group 3 pair by pair
    constexpr int NPC = NWP + NA;      // DMA pieces behind the barrier: W first (it must have landed one barrier earlier than A)
    h8 ah[2], al[2], wh[NJ], wl[NJ];
    f32x4 acc16[4][NJ];
    // Waits leave "the youngest N instructions" in flight, so the ORDER of issue decides what a count means.  In the loop every
    // barrier is followed by W(kt + WST) then A(kt + AST); for two equal rings of S slots the tiles kt+2 .. kt+S-1 of both sides
    // may stay in flight at X_kt = (S - 2)(NA + NWP), provided the prologue issued tile by tile as well (W_t, A_t).  With three A
    // slots over two W slots the prologue issues all W first (the 12-wave form's W pieces are uneven over the waves: no count
    // may include them) and only the A pieces of the youngest tile stay in flight.
    constexpr int kPendLoop = AST == WST ? (AST - 2) * (NA + NWP) : NA;
    if (AST == WST) {
#pragma unroll
        for (int t = 0; t < AST; ++t) {
#pragma unroll
            for (int q = 0; q < NWP; ++q) DMA_W(q, cur, t, t)
#pragma unroll
            for (int q = 0; q < NA; ++q) DMA_A(q, cur, t, t)
        }
        VMCNT_LGKM0(WPIECES % NW_ == 0 ? (AST - 1) * (NA + NWP) : 0)  // k-tile 0 has landed
    } else {
#pragma unroll
        for (int t = 0; t < WST; ++t)
#pragma unroll
            for (int q = 0; q < NWP; ++q) DMA_W(q, cur, t, t)
#pragma unroll
        for (int t = 0; t < AST; ++t)
#pragma unroll
            for (int q = 0; q < NA; ++q) DMA_A(q, cur, t, t)
        VMCNT_LGKM0((AST - 1) * NA)  // every W k-tile of the prologue and A k-tile 0 have landed
    }
    __builtin_amdgcn_s_barrier();
    GEMM_STAMP(st1)
#pragma unroll
    for (int j = 0; j < NJ; ++j) RD_W(j)
    RD_A(0, 0)
    // the DMA stream: k-tile kt + AST (WST) of this output tile or, past its end, of the next one (selected per k-tile with scalar
    // selects and one v_cndmask per piece offset: no branch inside the pinned loop body)
#define DMA_AFTER_BLOCK(b_)                                                                     \
    {                                                                                           \
        _Pragma("unroll") for (int q = 0; q < NPC; ++q)                                         \
            if (q * NBLK / NPC == (b_)) {                                                       \
                if (q < NWP) DMA_W(q < NWP ? q : 0, sel, tW, sw)                                \
                else DMA_A(q < NWP ? 0 : q - NWP, sel, tA, sa)                                  \
            }                                                                                   \
        SB()                                                                                    \
    }
    for (;;) {
        const bool has_next = PERSIST && wg_step > 0 && tile_i + wg_step < run_len;
        if (PERSIST && has_next) decode(tile_i + wg_step, nxt);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned a_step = A_STEP(), w_step = W_STEP();
            const bool a_cur = kt + AST < nk, w_cur = kt + WST < nk;
            const int tA = (a_cur || !PERSIST) ? kt + AST : kt + AST - nk, tW = (w_cur || !PERSIST) ? kt + WST : kt + WST - nk;
            // PERSIST: the stream's source is this output tile or, past its end, the next one (scalar selects and one v_cndmask per
            // piece offset: no branch inside the pinned body).  Without it (the 16-wave form: its 128 VGPRs have no
            // room for a second tile's lane offsets -- with them hipcc spilled INTO this loop and FFN1 / conv1 ran 8 % slower) the
            // stream stays on this tile; DMA_A / DMA_W clamp the k-tile index, so its last slots are re-fetched and never read.
            Tile sel;
            if (PERSIST) {
                sel.kt0 = a_cur ? cur.kt0 : nxt.kt0;
                sel.bAh = a_cur ? cur.bAh : nxt.bAh;
                sel.bAl = a_cur ? cur.bAl : nxt.bAl;
                sel.bWh = w_cur ? cur.bWh : nxt.bWh;
                sel.bWl = w_cur ? cur.bWl : nxt.bWl;
#pragma unroll
                for (int u = 0; u < NDA; ++u) sel.va[u] = a_cur ? cur.va[u] : nxt.va[u];
#pragma unroll
                for (int u = 0; u < NDW; ++u) sel.vw[u] = w_cur ? cur.vw[u] : nxt.vw[u];
            } else {
                sel = cur;
            }
            // MFMA blocks between two pins hold two W sub-tiles (two independent accumulator chains of TERMS MFMAs each, which hipcc
            // interleaves): three dependent MFMAs back to back leave the matrix pipe to the other waves of the SIMD for 2 x 16 cycles
            SB()
            MM(0, 0, 0)  // first: only fragments read a block ago are waited for here
            if (NJ > 1) MM(0, 0, 1)
            // (the empty asm pins these MFMAs here: instruction selection otherwise places the last one of a chain, a pure node whose
            // only user is the loop-carried copy, at the END of the body -- across every sched_barrier -- and keeps the old A
            // fragments alive for it)
            asm volatile("" : "+v"(acc16[0][0]), "+v"(acc16[0][NJ > 1 ? 1 : 0]));
            SB()
            RD_A(1, 1)
            SB()
#pragma unroll
            for (int j = 2; j < NJ; ++j) MM(0, 0, j)
            SB()
            RD_A(2, 0)
            SB()
#pragma unroll
            for (int j = 0; j < NJ; ++j) MM(1, 1, j)
            SB()
            RD_A(3, 1)
            SB()
#pragma unroll
            for (int j = 0; j < JH; ++j) MM(2, 0, j)
            SB()
            // X_kt: this wave's reads of k-tile kt's slots are complete (lgkmcnt) and its DMA pieces of k-tile kt+1 have landed (vmcnt:
            // kPendLoop instructions -- the k-tiles beyond kt+1 -- may stay in flight; stores of the previous output tile's epilogue
            // are younger still and only make the wait longer, never shorter)
            VMCNT_LGKM0(kPendLoop)
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int j = JH; j < NJ; ++j) MM(2, 0, j)
            SB()
            DMA_AFTER_BLOCK(0)
            a_ad += a_step;  // both sides now address the next k-tile's slots (past the end of the stream: stale bytes, never used)
            w_ad += w_step;
            RD_A(0, 0)
            SB()
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int j = 2 * g; j < 2 * g + 2 && j < NJ; ++j) MM(3, 1, j)
                SB()
                DMA_AFTER_BLOCK(1 + g)
#pragma unroll
                for (int j = 2 * g; j < 2 * g + 2 && j < NJ; ++j) RD_W(j)
                SB()
            }
            sa = sa + 1 == AST ? 0 : sa + 1;
            sw = sw + 1 == WST ? 0 : sw + 1;
            SB()
        }
        GEMM_STAMP(st2)

        // Epilogue of the output tile `cur`.  The accumulators hold 2^k times the product (the weight planes are pre-scaled,
        // GemmSplitArgs::out_scale): one exact multiply restores it.  amax = max|x| of what this lane writes into fp16 planes.
        {
            const int m0 = cur.m0, n0 = cur.n0, z1 = cur.z1, z2 = cur.z2;
            const long coff = cur.coff;
            const float osc = p.out_scale;
            float amax = 0.f;
            const unsigned seen = (OUT_SPLIT || EPI == kEpiQkvScatter) ? range_peek(p.range_slot) : 0u;  // early: its latency hides below
            if (PERMW) {
                // acc16[i][j][e] = C[m = m0 + wm*64 + 16 i + r16][n = n0 + wn*64 + 16 q4 + 4 j + e]
                const int n = n0 + wn * 64 + 16 * q4;
                if (n < p.N) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = m0 + wm * 64 + 16 * i + r16;
                        if (m >= p.M) continue;
                        f32x4 v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = acc16[i][j < NJ ? j : 0] * osc;
                            if (p.bias && n + 4 * j < p.N) v[j] += *reinterpret_cast<const f32x4*>(p.bias + z2 * p.sBias2 + n + 4 * j);
                        }
                        split_gemm_store16<EPI, OUT_SPLIT>(p, v, coff, m, n, amax, z1, z2);
                    }
                }
                range_commit(p.range_slot, amax, seen);
            } else {
                // acc16[i][j][e] = C[m = m0 + wm*64 + 16 i + r16][n = n0 + wn*64 + 16 j + 4 q4 + e]
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + wm * 64 + 16 * i + r16;
                    if (m >= p.M) continue;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int n = n0 + wn * 64 + 16 * j + 4 * q4;
                        if (n < p.N) {
                            f32x4 v = acc16[i][j] * osc;
                            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + z2 * p.sBias2 + n);
                            split_gemm_store<EPI, OUT_SPLIT>(p, v, coff, m, n, amax, z1, z2);
                        }
                    }
                }
                if (OUT_SPLIT || EPI == kEpiQkvScatter) range_commit(p.range_slot, amax, seen);
            }
        }
        if (!PERSIST || !has_next) break;
        cur = nxt;
        tile_i += wg_step;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus DMAs of the last k-tiles: none may land after this workgroup's LDS is given away
#undef DMA_AFTER_BLOCK
#undef A_STEP
#undef W_STEP
#undef SB
#undef MM
#undef RD_W
#undef RD_A
#undef LDS_H8
#undef DMA_W
#undef DMA_A
#ifdef LOCO_GEMM_STAMPS
    {
        GEMM_STAMP(st3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GEMM_STAMP(st4)
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1) :: "memory");
        if (g_gemm_stamps && tid == 0) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));
            unsigned long long* o_ = g_gemm_stamps + 8l * blockIdx.x;
            o_[0] = st1 - st0; o_[1] = st2 - st1; o_[2] = 0; o_[3] = st3 - st2; o_[4] = st4 - st3; o_[5] = rt0; o_[6] = rt1;
            o_[7] = hw;
        }
    }
#endif
}
#undef VMCNT_LGKM0
#undef DMA16

// ---------------------------------------------------------------------------------------------------------------
// The grouped positional conv with its A operand RESIDENT in LDS.  As a GEMM over the group-major halo layout (A row t = the
// contiguous run of 128 taps x 48 channels starting at input row t, lda = 48, K = 6144) every k-tile of every output row is a
// different 64-byte window of the SAME 640 input rows of a 512-frame tile: the generic kernel DMA's 12.6 MB through L2 -> LDS per
// tile to multiply 123 KB of unique data, 64 KiB per k-tile against 576 MFMA cycles per wave -- exactly the CU's LDS-DMA rate, so
// that kernel is bound by it (285 TFLOP/s against ~400 for the other GEMMs).  Here the 640 rows x 48 channels (hi + lo planes) are
// loaded ONCE and a fragment is read where it lies: chunk g8 = 4 kt + q4 of a row's k axis is tap g8 / 6, channels 8 (g8 % 6) .. + 7,
// i.e. LDS row (frame + tap), one 16-byte piece.  Only the weights stream (6 KiB per k-tile, three ring slots).  Same MFMA
// sequence per output element as the generic kernel (k-tiles ascending, the three terms in its order): bit-identical results
// (tools/bit_compare.py).
// LDS image of the block (round 4).  A ds_read_b128 is served in four groups of 16 lanes, and each group holds EIGHT rows of one
// k-chunk column and the eight OTHER rows of the next column ({0-3, 12-15} of q4 = 0 with {4-11} of q4 = 1, ...:
// MI355X_MICROARCH.md, LDS): any row-major image -- round 3 had rows padded to 7 x 16 B -- puts the two halves of a group one
// 16-byte slot apart, and two eight-element sets that tile the sixteen slots cannot stay disjoint under a shift by one (7 of 8
// lanes collided: SQ_LDS_BANK_CONFLICT = SQ_BUSY_CYCLES).  Lane quarter q4 only ever reads chunks of ITS parity (g8 = 4 kt + q4, six
// chunks per tap), so the block is stored as two regions -- even chunks, odd chunks -- of 48-byte rows (three slots, no padding:
// 3 r mod 16 is a bijection on sixteen consecutive rows), the odd region a multiple of 256 B behind the even one: both halves of a
// group then see the same slot pattern on complementary row sets, sixteen different slots.  Per k-tile a lane's chunk moves two
// places within its region (+32 B, across the row end included: three chunks per row per region).  20 KB less LDS, ten DMA
// instructions fewer per plane, no re-read padding pieces.
// NI = row groups of 16 frames per wave: 4 -> 512 frames per tile (the form of round 3), 2 -> 256 frames per tile (round 4: clips of
// utterance length -- a pack's 170 ... 300 frames -- fill half of a 512-frame tile; the launcher picks the cheaper form per T).
constexpr int PCR_S = 24;                                            // row pitch within a region (halves)
constexpr int PCR_WPL = kPosCg * SBK;                                // halves per W plane of one k-tile
constexpr int PCR_WST = 3;                                           // W ring slots
template <int NI> struct PcrGeom {
    static constexpr int BM = 8 * 16 * NI, ROWS = BM + kPosK;        // frames per tile, input rows per tile
    static constexpr int REG = ROWS * PCR_S, APL = 2 * REG;          // halves per region (even / odd chunks), per A plane
    static_assert((REG * 2) % 256 == 0 && (ROWS * 3) % 64 == 0, "region = whole bank rows and whole DMA instructions");
    static_assert((2 * APL + PCR_WST * 2 * PCR_WPL) * 2 <= 160 * 1024, "LDS");
};
#define PCR_DMA16(base_, voff_, ldsb_) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsb_), "v"(voff_), "s"(base_) : "memory")

template <int TERMS, int NI>
__global__ __launch_bounds__(512, 2) void pos_conv_resident_kernel(GemmSplitArgs p) {
    constexpr int PCR_BM = PcrGeom<NI>::BM, PCR_ROWS = PcrGeom<NI>::ROWS, PCR_REG = PcrGeom<NI>::REG, PCR_APL = PcrGeom<NI>::APL;
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * PCR_APL + PCR_WST * 2 * PCR_WPL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q4 = lane >> 4;
    const int t0 = blockIdx.x * PCR_BM, g = blockIdx.y, b = blockIdx.z;
    const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)lds;
    const int rows_total = p.M + kPosK;  // rows of the (clip, group) block: frames + the 64-frame halo on both sides
    const char* const abase_h = reinterpret_cast<const char*>(p.Ahi + b * p.sA1 + g * p.sA2);
    const char* const abase_l = reinterpret_cast<const char*>(p.Alo + b * p.sA1 + g * p.sA2);
    const char* const wbase_h = reinterpret_cast<const char*>(p.Whi + g * p.sW2);
    const char* const wbase_l = reinterpret_cast<const char*>(p.Wlo + g * p.sW2);

    // ---- the resident A block: piece P of a plane = LDS bytes 16 P .. 16 P + 15 = region P / 1920 (chunk parity), row (P % 1920) / 3,
    //      chunk 2 (P % 3) + parity; one instruction moves 64 pieces; rows past the block's end re-read its last row (they feed frames >= T)
    constexpr int kPiecesPerRegion = PCR_ROWS * 3, kInstrPerPlane = 2 * kPiecesPerRegion / 64;
    for (int n = wave; n < 2 * kInstrPerPlane; n += 8) {
        const int pl = n >= kInstrPerPlane, ni = pl ? n - kInstrPerPlane : n;
        const int P = 64 * ni + lane;
        const int par = P >= kPiecesPerRegion, Q = par ? P - kPiecesPerRegion : P;
        const int row = Q / 3, jj = Q - 3 * row;
        int grow = t0 + row;
        grow = grow < rows_total ? grow : rows_total - 1;
        const unsigned vo = (unsigned)grow * (2u * kPosCg) + 16u * (unsigned)(2 * jj + par);
        const unsigned d = lds0 + 2u * (unsigned)(pl * PCR_APL) + 1024u * (unsigned)ni;
        PCR_DMA16(pl ? abase_l : abase_h, vo, d);
    }
    // ---- W k-tile kt -> ring slot: waves 0-5 move one 16-row piece each (plane = wave / 3); lane -> row lane / 4, stored piece lane % 4
    //      holding source piece (lane % 4) ^ swz(row), swz = 3 * ((row >> 2) & 1) -- the generic kernel's conflict-free W image
    const int wrow = 16 * (wave % 3) + (lane >> 2), wpos = lane & 3;
    const unsigned wvo = (unsigned)wrow * (unsigned)(2 * p.ldw) + 16u * (unsigned)(wpos ^ (3 * ((wrow >> 2) & 1)));
    const int nk = p.K / SBK;
#define PCR_DMA_W(kt_, slot_)                                                                                                   \
    if (wave < 6) {                                                                                                             \
        const int kk_ = (kt_) < nk ? (kt_) : nk - 1;                                                                            \
        const unsigned d_ = lds0 + 2u * (unsigned)(2 * PCR_APL + (slot_) * 2 * PCR_WPL + (wave / 3) * PCR_WPL + 16 * (wave % 3) * SBK); \
        PCR_DMA16((wave >= 3 ? wbase_l : wbase_h) + (long)kk_ * (2 * SBK), wvo, d_);                                             \
    }
    PCR_DMA_W(0, 0)
    PCR_DMA_W(1, 1)

    // fragment addresses (halves).  A: row-group i of this wave = frames 16 NI wave + 16 i + r16; chunk g8 = 4 kt + q4 -> tap g8 / 6,
    // channel 8 (g8 % 6) = region q4 & 1, row frame + tap, piece (g8 % 6) / 2: within its region a lane's chunk index (g8 - parity) / 2
    // grows by two per k-tile and the region is row-major with three chunks per row, so aoff simply advances 16 halves per k-tile.
    int aoff = (q4 & 1) * PCR_REG + (16 * NI * wave + r16) * PCR_S + 8 * (q4 >> 1);
    const int woff = r16 * SBK + 8 * (q4 ^ (3 * ((r16 >> 2) & 1)));
    f32x4 acc[NI][3];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // W(kt) has landed (its DMA was issued two k-tiles ago; W(kt + 1) may still be in flight: 1 instruction per wave), and -- first
        // iteration -- so has the resident block; every wave has finished reading the slot W(kt + 2) is about to overwrite
        if (kt == 0 && wave >= 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // waves 6, 7 move no weights: their last A piece
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const int s2 = slot + 2 >= PCR_WST ? slot + 2 - PCR_WST : slot + 2;
            PCR_DMA_W(kt + 2, s2)
        }
        const _Float16* wb = lds + 2 * PCR_APL + slot * 2 * PCR_WPL;
        h8 wh[3], wl[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            wh[j] = *reinterpret_cast<const h8*>(wb + 16 * j * SBK + woff);
            if (TERMS == 3) wl[j] = *reinterpret_cast<const h8*>(wb + PCR_WPL + 16 * j * SBK + woff);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const h8 ah = *reinterpret_cast<const h8*>(lds + aoff + 16 * i * PCR_S);
            const h8 al = *reinterpret_cast<const h8*>(lds + PCR_APL + aoff + 16 * i * PCR_S);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (TERMS == 3) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], ah, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], al, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah, acc[i][j], 0, 0, 0);
            }
        }
        aoff += 16;
        slot = slot + 1 == PCR_WST ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus W DMAs of the last two k-tiles: none may land after the LDS is given away
#undef PCR_DMA_W

    // ---- epilogue: acc[i][j][e] = C[frame t0 + 64 wave + 16 i + r16][output 16 j + 4 q4 + e], the generic kernel's kEpiPosConv store
    const long coff = b * p.sC1 + g * p.sC2;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = t0 + 16 * NI * wave + 16 * i + r16;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = 16 * j + 4 * q4;
            f32x4 v = acc[i][j] * p.out_scale;
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + g * p.sBias2 + n);
            split_gemm_store<kEpiPosConv, false>(p, v, coff, m, n, amax, b, g);
        }
    }
}
#undef PCR_DMA16

// Sum of the ks partial results of the split-K path (fixed order) + bias, then the shared epilogue.  Thread = 4 columns; a grid of
// at most 1024 blocks strides over the output (few, fat workgroups: one range atomic each, range_commit_block).
template <int EPI, bool OUT_SPLIT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmSplitArgs p, int ks) {
    const long n4 = (long)p.M * (p.N / 4);
    float amax = 0.f;
    const unsigned seen = (OUT_SPLIT || EPI == kEpiQkvScatter) ? range_peek(p.range_slot) : 0u;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int m = (int)(i / (p.N / 4));
        const int n = 4 * (int)(i - (long)m * (p.N / 4));
        const float* part = p.splitk_ws + (long)m * p.N + n;
        f32x4 v = *reinterpret_cast<const f32x4*>(part);
        for (int k = 1; k < ks; ++k) v += *reinterpret_cast<const f32x4*>(part + (long)k * p.M * p.N);
        v *= p.out_scale;  // the partial sums carry the weight planes' 2^k
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        split_gemm_store<EPI, OUT_SPLIT>(p, v, 0, m, n, amax);
    }
    if (OUT_SPLIT || EPI == kEpiQkvScatter) range_commit_block(p.range_slot, amax, seen);  // every thread of the block gets here
}

// The same reduction for the fused q|k|v projection (kEpiQkvScatter), one 64-row x 64-column tile per workgroup: a thread owns four
// columns of a row, 16 threads complete a 128-byte run of one plane row.  (While V was stored transposed this kernel sent the V tiles
// through LDS to write them with the frame on the lane; V is row-major now, like q and k.)
__global__ __launch_bounds__(256) void splitk_reduce_qkv_kernel(GemmSplitArgs p, int ks) {
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int c4 = tid & 15, r0 = tid >> 4;
    float amax = 0.f;
    const unsigned seen = range_peek(p.range_slot);
    const int n = n0 + 4 * c4;
    const f32x4 bias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int third = n0 < kHidden ? 0 : (n0 < 2 * kHidden ? 1 : 2);  // block-uniform: 768 = 12 x 64
    _Float16* const dh = p.Chi + third * p.qkv_stride;
    _Float16* const dl = p.Clo + third * p.qkv_stride;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int ml = r0 + 16 * rr, m = m0 + ml;
        if (m >= p.M) continue;
        const float* part = p.splitk_ws + (long)m * p.N + n;
        f32x4 v = *reinterpret_cast<const f32x4*>(part);
        for (int k = 1; k < ks; ++k) v += *reinterpret_cast<const f32x4*>(part + (long)k * p.M * p.N);
        v = v * p.out_scale + bias;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        unsigned h0, l0, h1, l1;
        split_f16_2pairs(v[0], v[1], v[2], v[3], h0, l0, h1, l1);
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const long o = (long)m * kHidden + (n - third * kHidden);
        *reinterpret_cast<u32x2_t*>(dh + o) = u32x2_t{h0, h1};
        *reinterpret_cast<u32x2_t*>(dl + o) = u32x2_t{l0, l1};
    }
    range_commit_block(p.range_slot, amax, seen);
}

// The split-K reduction of the grouped positional conv: partial sums [clip][slice][group][frame][48], epilogue kEpiPosConv.
__global__ __launch_bounds__(256) void splitk_reduce_posconv_kernel(GemmSplitArgs p, int ks) {
    constexpr int n4s = kPosCg / 4;
    const long total = (long)p.nb1 * p.nb2 * p.M * n4s;
    float amax = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = 4 * (int)(i % n4s);
        long rest = i / n4s;
        const int m = (int)(rest % p.M);
        rest /= p.M;
        const int g = (int)(rest % p.nb2), b = (int)(rest / p.nb2);
        const long gstride = (long)p.M * kPosCg, sstride = (long)p.nb2 * gstride;
        const float* part = p.splitk_ws + ((long)b * ks * p.nb2 + g) * gstride + (long)m * kPosCg + n;
        f32x4 v = *reinterpret_cast<const f32x4*>(part);
        for (int k = 1; k < ks; ++k) v += *reinterpret_cast<const f32x4*>(part + k * sstride);
        v *= p.out_scale;
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + g * p.sBias2 + n);
        split_gemm_store<kEpiPosConv, false>(p, v, b * p.sC1 + g * p.sC2, m, n, amax, b, g);
    }
}

// A/B knobs of the tile dispatch (tools/gemm_split_bench.py, tools/gemm_stamps.py, tests/test_gpu_ops.py): LOCO_GEMM_TILE=<1..8>
// forces one tile form, LOCO_GEMM_TILE_NARROW=<n> replaces the 256x128 form, LOCO_GEMM_NOPERSIST / _NOCOLGROUP / _NO192 switch
// one mechanism off.  They are read from the environment ONCE (first launch) -- no getenv on the launch path, which runs ~67 times
// per forward and possibly on several host threads -- and again only when a tool asks for it (loco_debug_reload_gemm_knobs).
struct GemmKnobs {
    int tile = 0, narrow = 0;
    bool nopersist = false, nocolgroup = false, no192 = false, nosplitk = false, posconv_generic = false;
    int posconv_ni = 0;
};
static GemmKnobs read_gemm_knobs() {
    GemmKnobs k;
    const char* v;
    if ((v = getenv("LOCO_GEMM_TILE"))) k.tile = atoi(v);
    if ((v = getenv("LOCO_GEMM_TILE_NARROW"))) k.narrow = atoi(v);
    k.nopersist = getenv("LOCO_GEMM_NOPERSIST") != nullptr;
    k.nocolgroup = getenv("LOCO_GEMM_NOCOLGROUP") != nullptr;
    k.no192 = getenv("LOCO_GEMM_NO192") != nullptr;
    k.posconv_generic = getenv("LOCO_POSCONV_GENERIC") != nullptr;  // A/B: the positional conv on the generic GEMM kernel (A re-fetched per k-tile)
    if ((v = getenv("LOCO_POSCONV_NI"))) k.posconv_ni = (atoi(v) == 2 || atoi(v) == 4) ? atoi(v) : 0;  // A/B: force the resident kernel's tile (256 / 512 frames)
    k.nosplitk = getenv("LOCO_GEMM_NOSPLITK") != nullptr;  // A/B: small problems as ONE launch each (no partial sums, no reduction kernel)
    return k;
}
static GemmKnobs& gemm_knobs() {
    static GemmKnobs k = read_gemm_knobs();
    return k;
}
void reload_gemm_knobs() { gemm_knobs() = read_gemm_knobs(); }

// One tile form for every epilogue / output kind: WM x WN waves of 64 x 64, AST / WST ring slots, WPS as in the kernel template.
template <int WM, int WN, int AST, int WST, int WPS, int TERMS = 3>
static hipError_t launch_tile(const GemmSplitArgs& a, hipStream_t s) {
    constexpr int bm = 64 * WM, bn = 64 * WN;
    const int tm = (a.M + bm - 1) / bm, tn = (a.N + bn - 1) / bn;
    const long nb = (long)tm * tn * a.nb1 * a.nb2;
    if (nb <= 0 || nb > 0x7fffffffL) return hipErrorInvalidValue;
    const bool sp = a.Chi != nullptr;
    // Persistent form: one workgroup per CU slot (256 CUs x WPS-per-CU), each walking its XCD's run of tiles, when there are more
    // tiles than slots and the k-loop is at least as long as the A ring (the DMA stream looks AST k-tiles ahead, into the next
    // output tile at most).  LOCO_GEMM_NOPERSIST=1 (gemm_knobs) launches one workgroup per tile, for A/B runs and the stamp tool.
    // The forms with 12 waves or fewer have the registers for it (168+ per wave); the 16-wave form (128) does not and runs one tile per
    // workgroup (it gained 0-1 % from it; the 192x256 form gains 3-7 %, the table GEMM on the two-per-CU 128x128 form 12 %).
    constexpr bool kPersist = WM * WN <= 12;
    constexpr int slots = 256 * (WPS ? (4 * WPS) / (WM * WN) : 1);
    const bool persist = kPersist && nb > slots && a.K / SBK >= AST && !gemm_knobs().nopersist;
    const unsigned grid = persist ? (unsigned)slots : (unsigned)nb;
    const int wg_step = persist ? slots / 8 : 0;
    // column tiles per group of the in-XCD tile order (see the kernel's decode): the divisor-like value nearest sqrt(32)
    const int ngroups = (tn + 5) / 6;
    const int col_group = gemm_knobs().nocolgroup ? tn : (tn + ngroups - 1) / ngroups;
#define TILE_LAUNCH(EPI)                                                                                                              \
    if (sp) hipLaunchKernelGGL((gemm_f16x3_dma_kernel<EPI, true, WM, WN, AST, WST, 4, WPS, TERMS, kPersist>), dim3(grid), dim3(64 * WM * WN), 0, \
                               s, a, tm, tn, (int)nb, wg_step, col_group);                                                            \
    else hipLaunchKernelGGL((gemm_f16x3_dma_kernel<EPI, false, WM, WN, AST, WST, 4, WPS, TERMS, kPersist>), dim3(grid), dim3(64 * WM * WN), 0,  \
                            s, a, tm, tn, (int)nb, wg_step, col_group);
    switch (a.epilogue) {
        case kEpiNone: TILE_LAUNCH(kEpiNone) break;
        case kEpiGelu: TILE_LAUNCH(kEpiGelu) break;
        case kEpiResidual: TILE_LAUNCH(kEpiResidual) break;
        case kEpiQkvScatter: TILE_LAUNCH(kEpiQkvScatter) break;
        default: return hipErrorInvalidValue;
    }
#undef TILE_LAUNCH
    return hipGetLastError();
}

hipError_t launch_gemm_split(const GemmSplitArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % SBK != 0) return hipErrorInvalidValue;
    if ((a.lda | a.ldw | a.sA1 | a.sA2) & 7) return hipErrorInvalidValue;  // 16-byte staging of 8 halves
    if ((a.N | a.ldc | a.sC1 | a.sC2) & 3) return hipErrorInvalidValue;
    if (a.epilogue == kEpiResidual && ((!a.R && !(a.Rhi && a.Rlo)) || (a.ldr & (a.Rhi ? 7 : 3)))) return hipErrorInvalidValue;
    const bool split = a.Chi != nullptr;
    if (split ? (a.Clo == nullptr) : (a.C == nullptr)) return hipErrorInvalidValue;
    if (a.epilogue == kEpiQkvScatter &&
        (!split || a.N != kQkv || a.nb1 * a.nb2 != 1 || a.qkv_stride < (long)a.M * kHidden))
        return hipErrorInvalidValue;
    if (a.epilogue == kEpiPosConv) {
        // grouped positional conv: N = 48 outputs per group -> 512 x 64 tile (8 x 1 waves, 3 of 4 column sub-tiles computed)
        if (a.N != kPosCg || !a.C || !a.R || !a.sin_table || a.T <= 0 || split) return hipErrorInvalidValue;
        const int tm = (a.M + 511) / 512;
        const long nb = (long)tm * a.nb1 * a.nb2;
        if (nb <= 0 || nb > 0x7fffffffL) return hipErrorInvalidValue;
        // One or two short clips: 16 groups x B workgroups would each walk K = 6144 in 192 dependent k-tiles (200 us, the longest
        // kernel of a 2 ms forward).  Split-K over the taps: the slice becomes the inner half of z1 (GemmSplitArgs::z1_inner),
        // fp32 partial sums [clip][slice][group][frame][48], then splitk_reduce_posconv_kernel applies the epilogue.
        if (a.splitk_ws && nb <= 64 && a.z1_inner == 1 && !gemm_knobs().nosplitk) {
            // a FIXED slice count: a clip's result must not depend on how many neighbours share its batch (the summation order
            // is part of the result), so within this regime every batch size takes the same eight slices of 24 k-tiles
            const int ks = 8;
            if (a.K % (ks * SBK) == 0 && (size_t)ks * a.nb1 * a.nb2 * a.M * kPosCg * sizeof(float) <= kSplitKBytes) {
                GemmSplitArgs b = a;
                const int kslice = a.K / ks;
                b.splitk_ws = nullptr;
                b.bias = nullptr; b.R = nullptr; b.out_scale = 1.0f; b.sin_table = nullptr; b.frames = nullptr;
                b.C = a.splitk_ws; b.ldc = kPosCg;
                b.nb1 = a.nb1 * ks; b.z1_inner = ks;
                b.sA1i = kslice; b.sW1i = kslice;
                b.sC1 = (long)a.nb2 * a.M * kPosCg; b.sC2 = (long)a.M * kPosCg;
                b.K = kslice;
                b.epilogue = kEpiNone;
                const long nbp = (long)tm * b.nb1 * b.nb2;
                if (a.terms == 2)
                    hipLaunchKernelGGL((gemm_f16x3_dma_kernel<kEpiNone, false, 8, 1, 2, 2, 3, 0, 2>), dim3((unsigned)nbp), dim3(512), 0, s, b, tm, 1,
                                       (int)nbp, 0, 1);
                else
                    hipLaunchKernelGGL((gemm_f16x3_dma_kernel<kEpiNone, false, 8, 1, 2, 2, 3>), dim3((unsigned)nbp), dim3(512), 0, s, b, tm, 1,
                                       (int)nbp, 0, 1);
                hipError_t err = hipGetLastError();
                if (err != hipSuccess) return err;
                const long n4 = (long)a.nb1 * a.nb2 * a.M * (kPosCg / 4);
                const unsigned blocks = (unsigned)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
                hipLaunchKernelGGL(splitk_reduce_posconv_kernel, dim3(blocks), dim3(256), 0, s, a, ks);
                return hipGetLastError();
            }
        }
        // the layout the resident-A kernel is written for (launch_group_major_split): lda = 48, K = 128 taps x 48, z1 = clip, z2 = group
        const bool resident = !gemm_knobs().posconv_generic && a.z1_inner == 1 && a.lda == kPosCg && a.K == kPosK * kPosCg && a.nb2 == kPosGroups &&
                              a.nb1 <= 65535 && a.sA2 == (long)(a.M + kPosK) * kPosCg && ((2 * a.ldw) & 15) == 0;
        if (resident) {
            // 512- or 256-frame tiles: a tile of the small form costs kSmallTile of a large one (measured, tools/posconv_ab.py: the
            // weights' k-tile stream and the barriers do not shrink with the rows), so it wins where it saves more than that in
            // idle rows -- T <= 256, 513 ... 768, ...; never at T = 1 499 (3 x 1.0 against 6 x kSmallTile)
            constexpr double kSmallTile = 0.62;
            const int t512 = (a.M + 511) / 512, t256 = (a.M + 255) / 256;
            int ni = (t256 * kSmallTile < t512) ? 2 : 4;
            if (gemm_knobs().posconv_ni) ni = gemm_knobs().posconv_ni;  // A/B knob: LOCO_POSCONV_NI=2|4
            const dim3 grid((unsigned)(ni == 2 ? t256 : t512), kPosGroups, (unsigned)a.nb1);
            if (ni == 2) {
                if (a.terms == 2) hipLaunchKernelGGL((pos_conv_resident_kernel<2, 2>), grid, dim3(512), 0, s, a);
                else hipLaunchKernelGGL((pos_conv_resident_kernel<3, 2>), grid, dim3(512), 0, s, a);
            } else {
                if (a.terms == 2) hipLaunchKernelGGL((pos_conv_resident_kernel<2, 4>), grid, dim3(512), 0, s, a);
                else hipLaunchKernelGGL((pos_conv_resident_kernel<3, 4>), grid, dim3(512), 0, s, a);
            }
            return hipGetLastError();
        }
        if (a.terms == 2)
            hipLaunchKernelGGL((gemm_f16x3_dma_kernel<kEpiPosConv, false, 8, 1, 2, 2, 3, 0, 2>), dim3((unsigned)nb), dim3(512), 0, s, a, tm, 1,
                               (int)nb, 0, 1);
        else
            hipLaunchKernelGGL((gemm_f16x3_dma_kernel<kEpiPosConv, false, 8, 1, 2, 2, 3>), dim3((unsigned)nb), dim3(512), 0, s, a, tm, 1,
                               (int)nb, 0, 1);
        return hipGetLastError();
    }
    // Split-K for grids that cannot fill the chip (one 5 s utterance: M = 249 -> 12 workgroups for the FFN's second GEMM,
    // each walking K = 3072 in 96 dependent steps that are bound by HBM latency, not bandwidth): the K range is cut into ks
    // slices computed as a batch dimension of the same kernel (fp32 partial sums in a workspace), then summed in a fixed
    // order -- bitwise reproducible -- by a reduction kernel that applies the epilogue.
    if (a.ktaps < 1 || a.ktaps > 3 || (!a.kchan && a.ktaps > 1 && a.K % (a.ktaps * 2 * SBK) != 0) || a.K / SBK >= 32768) return hipErrorInvalidValue;
    if (a.splitk_ws && a.nb1 * a.nb2 == 1 && a.M <= kSplitKMaxM && !gemm_knobs().nosplitk) {
        const int bm = a.M >= 1024 ? 256 : 128;  // the tile the dispatch below picks for this M (N tile 128)
        const int tm = (a.M + bm - 1) / bm, tn = (a.N + 127) / 128;
        int ks = 256 / (tm * tn);
        if (ks > a.K / 256) ks = a.K / 256;  // >= 8 k-steps per slice
        while (ks > 1 && a.K % (ks * SBK) != 0) --ks;
        if (ks >= 2) {
            GemmSplitArgs b = a;
            const int kslice = a.K / ks;
            b.splitk_ws = nullptr;
            b.bias = nullptr; b.R = nullptr; b.Chi = nullptr; b.Clo = nullptr; b.out_scale = 1.0f; b.range_slot = nullptr;
            b.C = a.splitk_ws; b.ldc = a.N;
            b.nb1 = 1; b.nb2 = ks; b.sA1 = 0; b.sC1 = 0;
            b.sA2 = kslice; b.sW2 = kslice; b.sC2 = (long)a.M * a.N; b.sBias2 = 0;
            if (a.ktaps > 1) {  // channel-block-major k walk: a slice is a run of k-tiles of that walk, not a contiguous piece of an A row
                b.sA2 = 0;
                b.kt_per_z2 = kslice / SBK;
                b.kchan = a.kchan ? a.kchan : a.K / a.ktaps;
            }
            b.K = kslice;
            b.epilogue = kEpiNone;
            hipError_t err = launch_gemm_split(b, s);
            if (err != hipSuccess) return err;
            const long n4 = (long)a.M * (a.N / 4);
            const unsigned blocks = (unsigned)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
#define RED_LAUNCH(EPI)                                                                                                  \
            if (split || EPI == kEpiQkvScatter) hipLaunchKernelGGL((splitk_reduce_kernel<EPI, true>), dim3(blocks), dim3(256), 0, s, a, ks); \
            else hipLaunchKernelGGL((splitk_reduce_kernel<EPI, false>), dim3(blocks), dim3(256), 0, s, a, ks);
            switch (a.epilogue) {
                case kEpiNone: RED_LAUNCH(kEpiNone) break;
                case kEpiGelu: RED_LAUNCH(kEpiGelu) break;
                case kEpiResidual: RED_LAUNCH(kEpiResidual) break;
                case kEpiQkvScatter:
                    hipLaunchKernelGGL(splitk_reduce_qkv_kernel, dim3((unsigned)((a.M + 63) / 64), kQkv / 64), dim3(256), 0, s, a, ks);
                    break;
                default: return hipErrorInvalidValue;
            }
#undef RED_LAUNCH
            return hipGetLastError();
        }
    }
    // Tile choice (tools/gemm_split_bench.py, MI355X).  LOCO_GEMM_TILE=<1..8> forces one form (gemm_knobs: a bench A/Bs the forms
    // inside one process -- same device, same clocks -- by changing the environment and calling loco_debug_reload_gemm_knobs).
    int tile = gemm_knobs().tile;
    if (tile == 0) {
        const long rows_total = (long)a.M * a.nb1 * a.nb2;
        if (rows_total >= 1024 && a.K <= 128) {
            // K <= 128 (the relative-position table as a GEMM: two k-tiles, then 128 KiB of stores per tile) is epilogue-bound and runs
            // best as two small workgroups per CU (128x128 / 4 waves / 64 KiB: -6 %)
            tile = 4;
        } else if (rows_total >= 1024) {
            // Tile form by a cost model (round 4; tools/gemm_split_bench.py [--pack] --tiles=0,1,2,6): a launch costs whole ROUNDS of 256
            // workgroups, so  cost(form) = ceil(tiles / 256) x bm x bn / eff(form)  with the per-FLOP efficiencies measured on this part
            // -- 256x256 / 16 waves 1.0, 192x256 / 12 waves 0.91 (more L2 -> LDS bytes per FLOP, 3 waves per SIMD), 256x128 / 8 waves
            // 0.80, 128x128 / 4 waves 0.55 -- and tiles counted per batch entry (a conv layer of 64 clips x 799 frames wastes the last
            // row tile of EVERY clip).  At 30 s x 32 (M = 47 968) it reproduces the choices measured there in round 2: 256x256 for QKV /
            // FFN1 / the conv layers, 192x256 for the N = 768 GEMMs (750 tiles = 2.93 rounds of a tile 3/4 the size against 2.2 rounds
            // paid as 3).  Rounds 2-3 applied it only between those two forms and fell back to 256x128 whenever fewer than 256 tiles of
            // 256x256 existed, and to 128x128 whenever ONE batch entry had fewer than 1024 rows: at the shapes of a pack of utterances
            // (M = 16 000: 189 tiles of 256x256 for N = 768; conv layers 4-6 with 199-799 frames per clip) that cost 30-65 % on out-proj,
            // FFN2, the feature projection and the small conv layers -- 219 -> 306, 277 -> 397, 242 -> 302, 201 -> 331 TFLOP/s.
            // Under the two half-batch schedule the other stream's workgroups fill a partly filled round, so rounds are not rounded up.
            struct Form { int id, bm, bn; double eff; bool n256; };
            static const Form kForms[] = {{1, 256, 256, 1.00, true}, {6, 192, 256, 0.91, true}, {2, 256, 128, 0.80, false}, {5, 128, 128, 0.55, false}};
            double best = 0.0;
            for (const Form& f : kForms) {
                if (f.n256 && a.N % 256 != 0) continue;
                if (f.id == 6 && gemm_knobs().no192) continue;
                const double tiles = (double)((a.M + f.bm - 1) / f.bm) * ((a.N + f.bn - 1) / f.bn) * a.nb1 * a.nb2;
                const double rounds = a.co_scheduled ? tiles / 256.0 : (double)(((long)tiles + 255) / 256);
                const double cost = rounds * f.bm * f.bn / f.eff;
                if (tile == 0 || cost < best) { tile = f.id; best = cost; }
            }
            if (tile == 2 && gemm_knobs().narrow) tile = gemm_knobs().narrow;  // A/B knob for the GEMMs that would take the 256x128 form
        } else {
            tile = 5;  // small problems (short clips without a split-K workspace, the text branch, tests): 128 x 128, 4 waves, 3 stages
        }
    }
    // ring depths: 256x256 -> three A slots + two W slots = 160 KiB; 192x256 -> 3 + 2 = 136 KiB; 256x128 and the one-per-CU
    // 128x128 -> 3 + 3 (five slots per side were tried for the latter: tile 8); the two-per-CU forms 2 + 2
    if (a.terms == 2) {  // precision mode "f16x2": the weights' lo plane is neither streamed nor multiplied
        switch (tile) {
            case 1: return launch_tile<4, 4, 3, 2, 0, 2>(a, s);
            case 2: return launch_tile<4, 2, 3, 3, 0, 2>(a, s);
            case 4: return launch_tile<2, 2, 2, 2, 2, 2>(a, s);
            case 6: return launch_tile<3, 4, 3, 2, 0, 2>(a, s);
            case 3:
            case 5: return launch_tile<2, 2, 3, 3, 0, 2>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (tile) {
        case 1: return launch_tile<4, 4, 3, 2, 0>(a, s);
        case 2: return launch_tile<4, 2, 3, 3, 0>(a, s);
        case 3: return launch_tile<3, 2, 2, 2, 3>(a, s);
        case 4: return launch_tile<2, 2, 2, 2, 2>(a, s);
        case 5: return launch_tile<2, 2, 3, 3, 0>(a, s);
        case 8: return launch_tile<2, 2, 5, 5, 0>(a, s);  // A/B: the small-M form with five slots per side (160 KiB): no faster --
                                                          // its few workgroups are not waiting for their DMA stream
        case 6: return launch_tile<3, 4, 3, 2, 0>(a, s);
        case 7: return launch_tile<4, 4, 2, 2, 0>(a, s);  // A/B: the 256x256 form with two slots per side (128 KiB)
        default: return hipErrorInvalidValue;
    }
}

// x -> (hi, lo) fp16 planes, n % 4 == 0
__global__ void split_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo, long n4, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i] * scale;
        h4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = (_Float16)v[e];
            b[e] = (_Float16)(v[e] - (float)a[e]);
        }
        reinterpret_cast<h4*>(hi)[i] = a;
        reinterpret_cast<h4*>(lo)[i] = b;
    }
}

// x [B,T,768] -> group-major hi/lo planes [B][16][T+128][48] with 64 zero frames of halo on both sides
__global__ void group_major_split_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int T,
                                         long total4, float* __restrict__ range_slot, const int32_t* __restrict__ rows_clip) {
    const int rows = T + kPosK;
    float amax = 0.f;
    const unsigned seen = range_peek(range_slot);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (kPosCg / 4));
        long rest = i / (kPosCg / 4);
        const int row = (int)(rest % rows);
        rest /= rows;
        const int g = (int)(rest % kPosGroups);
        const int b = (int)(rest / kPosGroups);
        const int t = row - kPosK / 2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < (rows_clip ? rows_clip[b] : T)) v = *reinterpret_cast<const f32x4*>(x + ((long)b * T + t) * kHidden + g * kPosCg + 4 * c4);
        h4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            amax = fmaxf(amax, fabsf(v[e]));
            a[e] = (_Float16)v[e];
            c[e] = (_Float16)(v[e] - (float)a[e]);
        }
        reinterpret_cast<h4*>(hi)[i] = a;
        reinterpret_cast<h4*>(lo)[i] = c;
    }
    range_commit_block(range_slot, amax, seen);
}

hipError_t launch_group_major_split(const float* x, void* hi, void* lo, int B, int T, hipStream_t s, float* range_slot,
                                    const int32_t* rows_clip) {
    if (B <= 0 || T <= 0) return hipErrorInvalidValue;
    const long total4 = (long)B * kPosGroups * (T + kPosK) * (kPosCg / 4);
    hipLaunchKernelGGL(group_major_split_kernel, dim3(2048), dim3(256), 0, s, x, (_Float16*)hi, (_Float16*)lo, T, total4, range_slot, rows_clip);
    return hipGetLastError();
}

// [g][tap][o][i] -> [g][o][tap][i]
__global__ void pos_w_for_gemm_kernel(const float* __restrict__ wf, float* __restrict__ out) {
    const long total = (long)kHidden * kPosCg * kPosK;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx % kPosCg);
        long rest = idx / kPosCg;
        const int tap = (int)(rest % kPosK);
        rest /= kPosK;
        const int o = (int)(rest % kPosCg);
        const int g = (int)(rest / kPosCg);
        out[idx] = wf[(((long)g * kPosK + tap) * kPosCg + o) * kPosCg + i];
    }
}

hipError_t launch_pos_w_for_gemm(const float* wf, float* out, hipStream_t s) {
    hipLaunchKernelGGL(pos_w_for_gemm_kernel, dim3(1024), dim3(256), 0, s, wf, out);
    return hipGetLastError();
}

hipError_t launch_split_f16(const float* x, void* hi, void* lo, long n, hipStream_t s, float scale) {
    if (n <= 0 || (n & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(split_f16_kernel, dim3(2048), dim3(256), 0, s, x, (_Float16*)hi, (_Float16*)lo, n / 4, scale);
    return hipGetLastError();
}

}  // namespace loco

#ifdef LOCO_GEMM_STAMPS
extern "C" int loco_debug_set_gemm_stamps(void* buf) {  // diagnostic build only: 8 x u64 per workgroup of the next launches
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(loco::g_gemm_stamps), &buf, sizeof(buf));
}
#endif

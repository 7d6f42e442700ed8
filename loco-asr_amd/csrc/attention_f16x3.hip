// Self-attention core on the split-precision fp16 x3 MFMA (precision mode f16x3) -- same algorithm, layout tricks and
// relative-position handling as attention_f32.hip (flash-style online softmax, swapped product S^T = K Q^T so the
// query sits on the lane, per-row constant bias for clipped tiles, coalesced + transposed band for the diagonal),
// with both matrix products issued as three v_mfma_f32_32x32x16_f16 per fp32-class product:
//
//   S^T  += K_lo Q_hi^T + K_hi Q_lo^T + K_hi Q_hi^T          (q, k arrive as fp16 hi/lo planes from the QKV GEMM epilogue)
//   O^T  += V_lo^T P_hi + V_hi^T P_lo + V_hi^T P_hi           (P = exp2(..) is split in registers; V arrives ROW-major
//                                                              like q and k and is transposed by the LDS read)
//
// 48 MFMAs x 32 cycles per 64-key tile and wave instead of 128 x 64 cycles: 5.3x fewer matrix-pipe cycles; softmax
// statistics, the running (m, l) and the O accumulators stay fp32.
//
// P as the next MFMA's B operand without touching LDS: the C/D fragment of S^T holds, in lane-half h, register 8s+j,
// key 16s + 8(j>>2) + 4h + (j&3) of a 32-key sub-tile; the 32x32x16 B operand wants k = 8h + j from that lane half, so
// registers 8s..8s+7 ARE k-step s up to a permutation of k: lane half h holds keys 16s + 4h + {0..3} and 16s + 8 + 4h + {0..3}.
// The matching A fragment -- V[those 8 keys][d = the lane's row] -- is gathered from the row-major V tile by two
// ds_read_b64_tr_b16 (a 4-key x 16-d block per 16 lanes, delivered column-major: element q of lane i = key q, d = i), so V needs
// no transposed copy in memory.  (Rounds 1-2 had the QKV epilogue write V^T planes with the permutation baked in: 2-byte stores a row
// pitch apart, 0.76 ms per step at 30 s x 32, a memset of the pad columns per forward and a reduction kernel of its own.)
//
// LDS: two-deep rings of K and V tiles (hi + lo planes, 64 keys x 128 bytes, unpadded: filled by LDS-DMA, 16-byte pieces
// XOR-swizzled on the source address so that ds_read_b128 (K) and ds_read_b64_tr_b16 (V) are conflict-free) = 64 KiB + the per-wave bias transpose
// scratch = 72.5 KiB, two workgroups per CU.  The loop is software-pipelined INSIDE each wave (QK of tile t+1 against the
// softmax of tile t, see the main loop): the chip is power-limited here, so what that buys is fewer stalls per joule, not
// a higher matrix-pipe duty cycle at the nominal clock.
#include <cstdlib>

#include "loco_kernels.h"

namespace loco {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int AX_BQ = 128, AX_BK = 64;
constexpr int AX_PL = AX_BK * kHeadDim;  // halves per plane of a tile (64 x 64, unpadded: XOR-swizzled 16-byte pieces)
constexpr int AX_STG = 2 * AX_PL;        // hi + lo plane

typedef __attribute__((address_space(1))) const void* ax_gptr_t;
typedef __attribute__((address_space(3))) void* ax_lptr_t;

// LOCO_ATTN_HACK (timing-only diagnostic builds, WRONG results; tools/ab/build_variant.sh): 1 = every table block multiplies pe_k block 0
// (its fragments stay in the L1: what do the 80 KiB of pe_k planes per wave cost?), 2 = the table is not stored, 3 = the main loop
// issues only every other K / V DMA piece (half the L2 -> LDS bytes and DMA instructions per tile at unchanged MFMA / exp work: an
// UPPER BOUND on what a workgroup of 8 waves x 32 queries sharing one K / V ring could gain -- round 4 pricing, profiles/r04_attention_8wave.txt).
#ifndef LOCO_ATTN_HACK
#define LOCO_ATTN_HACK 0
#endif
#define AX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define AX_PIN() __builtin_amdgcn_sched_barrier(0)

// Diagnostic build only (-DLOCO_ATTN_STAMPS, tools/attn_stamps.py): wave 0 of every workgroup sums, per segment of the loop body,
// the s_memtime cycles it spent there (DMA issue + rescale / block 1 / blocks 2-3 / bookkeeping + band / wait + barrier).
#ifdef LOCO_ATTN_STAMPS
__device__ unsigned long long* g_attn_stamps = nullptr;
#define AX_STAMP(k_)                                                                                                     \
    {                                                                                                                    \
        unsigned long long t_;                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        seg[k_] += t_ - tlast;                                                                                           \
        tlast = t_;                                                                                                      \
    }
#else
#define AX_STAMP(k_) {}
#endif

// p = exp2(s log2e + dsh) for a register pair; hi = fp16(p) by one cvt_pk, lo = fp16(p - hi) by one mixed-precision FMA per
// element (fp32 p, fp16 hi: the difference is exact, so lo is rounded once, like the host-side split).
__device__ __forceinline__ void ax_split_pair(const f32x2 pv, unsigned& hi, unsigned& lo) {
    // The s_nop is load-bearing: pv usually comes straight from v_exp_f32, and gfx950 needs one wait state between a
    // transcendental result and a VALU reader.  hipcc pads its own instructions but does not look inside inline asm
    // (seen as wrong results in the lanes the quarter-rate unit finishes last).
    asm("s_nop 0\n\tv_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(pv.x), "v"(pv.y));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(pv.x), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(pv.y), "v"(hi));
}

// Diagnostic build only (-DLOCO_ATTN_DEBUG, tools/attn_nan_probe.py): every lane records, per key tile, a bit mask of the
// quantities that were not finite at the end of that tile's iteration and the values of the bookkeeping scalars.
#ifdef LOCO_ATTN_DEBUG
__device__ float* g_attn_dbg = nullptr;  // [workgroup][thread][tile < 8][8 floats]
#define AX_DBG(tile_, code_, S0_, S1_)                                                                                   \
    if (g_attn_dbg && (tile_) < 8) {                                                                                     \
        float* d_ = g_attn_dbg + (((long)blockIdx.x * 256 + tid) * 8 + (tile_)) * 8;                                     \
        unsigned mask_ = 0;                                                                                              \
        float smax_ = -INFINITY, smin_ = INFINITY;                                                                       \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                                 \
            if (!(fabsf(S0_[e]) <= 3e38f) && S0_[e] != -INFINITY) mask_ |= 1u;                                           \
            if (!(fabsf(S1_[e]) <= 3e38f) && S1_[e] != -INFINITY) mask_ |= 1u;                                           \
            smax_ = fmaxf(smax_, fmaxf(S0_[e], S1_[e])); smin_ = fminf(smin_, fminf(S0_[e], S1_[e]));                    \
            if (!(fabsf(o0[e]) <= 3e38f) || !(fabsf(o1[e]) <= 3e38f)) mask_ |= 2u;                                       \
        }                                                                                                                \
        if (!(fabsf(l_run) <= 3e38f)) mask_ |= 4u;                                                                       \
        if (!(fabsf(alpha) <= 3e38f)) mask_ |= 8u;                                                                       \
        if (!(fabsf(dsh) <= 3e38f)) mask_ |= 16u;                                                                        \
        d_[0] = __uint_as_float(mask_ | ((code_) << 16)); d_[1] = m_run; d_[2] = alpha; d_[3] = dsh; d_[4] = l_run;      \
        d_[5] = smax_; d_[6] = smin_; d_[7] = o0[0];                                                                     \
    }
#else
#define AX_DBG(tile_, code_, S0_, S1_) {}
#endif

// two transposed 64-bit LDS reads (4 keys each) -> the 8-half A fragment of one 16-key step
typedef __fp16 ax_tr4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h8 ax_join_tr(ax_tr4_t a, ax_tr4_t b) {
    const h4 x = __builtin_bit_cast(h4, a), y = __builtin_bit_cast(h4, b);
    return h8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
}

// TABLE: the relative-position table Qp[b, head, i, 0..319] = q_scaled[i] . pe_k^T (HF modeling:432-441, 939-945) is computed HERE,
// by the wave that owns query i, instead of by a GEMM launch in front of this kernel: ten 32x32 blocks of Qp^T = pe_k Q^T on the
// Q fragments already in registers (120 MFMAs per wave: what the table GEMM spent), written to the same [B,12,T,320] buffer and
// read back by the band tiles exactly as before -- the SAME WAVE reads what it wrote, so no other workgroup is involved.  What
// moves is the cost: 884 MB of fp32 stores per launch at 30 s x 32 used to be an HBM-bound kernel of its own (2.9 ms per step, nothing
// else on the chip); issued from here they drain under the MFMA / exp work of 2 048 other waves.
template <bool OUT_SPLIT, bool TABLE, bool LONG_SEQ>
__global__ __launch_bounds__(256, 2) void attention_f16x3_kernel(const _Float16* __restrict__ qhi, const _Float16* __restrict__ qlo,
                                                                 const _Float16* __restrict__ khi, const _Float16* __restrict__ klo,
                                                                 const _Float16* __restrict__ vhi, const _Float16* __restrict__ vlo,
                                                                 float* qp, const int32_t* __restrict__ frames,
                                                                 _Float16* __restrict__ ctx_hi, _Float16* __restrict__ ctx_lo,
                                                                 float* __restrict__ ctx, int T, int nqb,
                                                                 const _Float16* __restrict__ pe_hi, const _Float16* __restrict__ pe_lo,
                                                                 float pe_scale) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[4 * AX_STG];  // K ring (2 tiles), V^T ring (2 tiles)
    __shared__ __attribute__((aligned(16))) float bias_stage[4][32 * 17];

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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: kept in an SGPR (the DMA asm takes "s" operands)
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
    float* qprow = qp + (((long)b * kHeads + head) * T + iqc) * kRelN;

    // LDS-DMA descriptors: one wave instruction writes 1 KiB = 8 rows x 8 pieces of 16 bytes; lane -> row lane/8, stored
    // position lane%8, which holds source piece position ^ swz(row), swz(row) = {row bit 4, row bit 3, row bit 1}.
    // ds_read_b128 serves a wave in four fixed groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) over 64
    // banks: with one 128-byte row per lane the 16 lanes of a group must cover both row parities x 8 distinct pieces,
    // which is exactly what those three row bits separate (checked with SQ_LDS_BANK_CONFLICT = 0).
    const int drow = lane >> 3, dpos = lane & 7;
    // Eight DMA instructions per wave move one K tile and one V^T tile (hi + lo planes).  The loop issues them ONE AT A TIME between
    // the MFMAs of block 1 (AX_DMA_PIECE): back to back at the top of the iteration they held the wave for ~1 000 of its ~5 200
    // cycles per key tile with no MFMA of its own in flight (tools/attn_stamps.py).  Scalar-base form (global_load_lds_dwordx4 voffset, sbase; M0 = LDS address of the piece): the base
    // = plane + tile origin is wave-uniform, the lane offset fits 32 bits.  The tile index is clamped to the last tile instead of
    // branching: the surplus DMAs of the last two iterations re-fetch it into a slot whose tile has been consumed, and the
    // vmcnt(0) that ends every iteration retires them.
    const unsigned lds0 = (unsigned)(unsigned long)(ax_lptr_t)lds;
    // column part of the lane offset (the row part depends on the clamp at T).  K pieces are swizzled for its ds_read_b128 fragments
    // (above); V pieces by {row bit 1} << 2: a transposed read takes, per 32 lanes, 4 consecutive keys x one 64-byte half of their
    // rows -- keys q and q + 2 fall on the same bank half (128-byte rows), so the swizzle sends them to different 64-byte halves.
    unsigned kvc[2], vvc;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int swz = (((2 * wave + i) & 3) << 1) | ((drow >> 1) & 1);
        kvc[i] = 2u * (unsigned)(head * kHeadDim + 8 * (dpos ^ swz));
    }
    vvc = 2u * (unsigned)(head * kHeadDim + 8 * (dpos ^ (((drow >> 1) & 1) << 2)));
    const char* const kbase_h = reinterpret_cast<const char*>(khi + (long)b * T * kHidden);
    const char* const kbase_l = reinterpret_cast<const char*>(klo + (long)b * T * kHidden);
    const char* const vbase_h = reinterpret_cast<const char*>(vhi + (long)b * T * kHidden);
    const char* const vbase_l = reinterpret_cast<const char*>(vlo + (long)b * T * kHidden);
#define AX_DMA16(base_, voff_, ldsb_) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsb_), "v"(voff_), "s"(base_) : "memory")
#define AX_DMA_PIECE(pc_, tk_, tv_, kslot_, vslot_)                                                                          \
    {                                                                                                                        \
        const int i_ = ((pc_) >> 1) & 1, pl_ = (pc_) & 1;                                                                    \
        const bool isk_ = (pc_) < 4;                                                                                         \
        const int tt_ = isk_ ? (tk_) : (tv_);                                                                                \
        const int rmax_ = T - 1 - tt_ * AX_BK; /* rows of the tile past the last key read key T-1 (masked later) */          \
        int row_ = 8 * (2 * wave + i_) + drow;                                                                               \
        row_ = row_ < rmax_ ? row_ : rmax_;                                                                                  \
        const unsigned vo_ = (unsigned)row_ * (2u * kHidden) + (isk_ ? kvc[i_] : vvc);                                       \
        const unsigned d_ = lds0 + 2u * (unsigned)((isk_ ? (kslot_) : 2 + (vslot_)) * AX_STG + pl_ * AX_PL + 8 * (2 * wave + i_) * kHeadDim); \
        AX_DMA16((isk_ ? (pl_ ? kbase_l : kbase_h) : (pl_ ? vbase_l : vbase_h)) + (long)tt_ * (AX_BK * kHidden * 2), vo_, d_); \
    }

    // fragment addresses (halves, within a tile): row 32 x + r, piece (2 y + h) ^ swz(r); x = key sub-tile (K) or d half
    // (V^T), y = k-step (K) or 16-key group (V^T: the keys of a group are stored in the order the P operand wants, see
    // loco_kernels.h, so one 16-byte piece per lane half is a whole A fragment)
    int fo[4];
    {
        const int swz = (((r >> 3) & 3) << 1) | ((r >> 1) & 1);
#pragma unroll
        for (int y = 0; y < 4; ++y) fo[y] = r * kHeadDim + 8 * ((2 * y + h) ^ swz);
    }
#define AX_KF(kb_, st_, ks_, pl_) (*reinterpret_cast<const h8*>((kb_) + (pl_) * AX_PL + (st_) * 32 * kHeadDim + fo[ks_]))
    // V^T fragment of the 16-key step c_ (keys 16 c_ .. + 15 of the tile), d half dt_: lane (group G = lane / 16: h = G / 2, column half
    // g = G % 2; q = (lane / 4) % 4, p = lane % 4) supplies the address of key 16 c_ + 4 h + 8 u + q, d = 32 dt_ + 16 g + 4 p .. + 3 and
    // receives, for u = 0, 1, keys 16 c_ + 4 h + 8 u + {0..3} at d = 32 dt_ + (lane % 32): exactly elements 4 u .. 4 u + 3 of the A operand
    // whose k = 8 h + j pairs with P's register order.  vfo[dt_] holds the lane part (the swizzle bit follows q, so dt_ cannot be an immediate).
    int vfo[2];
    {
        const int q = (lane >> 2) & 3, pp = lane & 3, g = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            vfo[dt] = (4 * h + q) * kHeadDim + 8 * ((4 * (dt ^ ((q >> 1) & 1))) + 2 * g + (pp >> 1)) + 4 * (pp & 1);
    }
    typedef __fp16 ax_tr4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    typedef __attribute__((address_space(3))) ax_tr4* ax_tr4_lptr;
#define AX_VTR(vb_, dt_, c_, u_, pl_) \
    __builtin_amdgcn_ds_read_tr16_b64_v4f16((ax_tr4_lptr)((vb_) + (pl_) * AX_PL + (16 * (c_) + 8 * (u_)) * kHeadDim + vfo[dt_]))
#define AX_VF(vb_, dt_, c_, pl_) ax_join_tr(AX_VTR(vb_, dt_, c_, 0, pl_), AX_VTR(vb_, dt_, c_, 1, pl_))

    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    float alpha = 0.f, dsh = 0.f;  // of the tile whose scores sit in (sa0, sa1) / (sb0, sb1) at the top of an iteration

    // ---- relative-position bias, key mask, row max and the online-softmax bookkeeping for one tile's raw scores.
    //      mx_ is the max of the raw scores (valid for tiles with a constant bias and no mask; recomputed otherwise).
    // The diagonal band (key tiles with |i - j| < 160 for some pair).  The table is read COALESCED -- lane (lq, lj) loads
    // Qp[query 4u + lq][key lj of a 16-key group]: 16 consecutive keys are 16 consecutive floats -- and transposed to the
    // accumulator layout (query on the lane) through a per-wave LDS scratch.  The 32 loads of tile t+1 are issued at the TOP of
    // iteration t and land in registers while its MFMAs run; only the LDS transposition remains in the bookkeeping.  (Loaded
    // right where they were needed, four dependent rounds of load -> LDS -> read cost ~2 900 cycles per band tile, 14 % of the
    // kernel at T = 1499; per-lane 16-byte gathers straight into the accumulator layout need no LDS but doubled the L2
    // requests and made the launch 30 % slower: tools/attn_stamps.py, tools/attn_bench.py.)
    float bandv[2][2][8];
#define AX_IS_BAND(tt_) (iw0 - ((tt_) * AX_BK + AX_BK - 1) < kRelMax - 1 && iw0 + 31 - (tt_) * AX_BK > -kRelMax)
#define AX_BAND_LOAD(tt_)                                                                                              \
    if (AX_IS_BAND(tt_)) {                                                                                             \
        const float* qpb = qp + ((long)b * kHeads + head) * T * kRelN;                                                 \
        const int lj = lane & 15, lq = lane >> 4;                                                                      \
        _Pragma("unroll") for (int st_ = 0; st_ < 2; ++st_)                                                            \
            _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                   \
                const int j = (tt_) * AX_BK + 32 * st_ + 16 * half + lj;                                               \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                        \
                    const int i = iw0 + 4 * u + lq;                                                                    \
                    int rel = i - j;                                                                                   \
                    rel = rel < -kRelMax ? -kRelMax : (rel > kRelMax - 1 ? kRelMax - 1 : rel);                         \
                    bandv[st_][half][u] = qpb[(long)(i < T ? i : T - 1) * kRelN + rel + kRelMax];                      \
                }                                                                                                      \
            }                                                                                                          \
    }
// The scratch holds 32 rows (queries) of 16 floats (keys); row r starts at 16 r + r / 2 (527 floats in all, rows never overlap).
// ds_write_b32 / ds_read_b32 are served 32 lanes at a time over 32 banks (MI355X_MICROARCH.md, LDS): a write group is two
// neighbouring rows x 16 keys -- the rows of a pair start 16 banks apart -- and a read group is one column of all 32 rows, whose starts
// 16 (r % 2) + r / 2 are the 32 different banks.  (Rows padded to 17 floats, rounds 1-3, were conflict-free for the reads but put key 15
// of the second row of every write group on the bank of key 0 of the first: SQ_LDS_BANK_CONFLICT = 6.6e6 per launch at 30 s x 32, two
// extra cycles for each of these stores -- hidden behind the store's own register transfer, so this is a counter put right, not time.)
#define AX_BAND_ADD(S_, st_)                                                                                           \
    _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                           \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) sc[66 * u + band_wr] = bandv[st_][half][u];                      \
        __builtin_amdgcn_wave_barrier();                                                                               \
        _Pragma("unroll") for (int e8 = 0; e8 < 8; ++e8)                                                               \
            S_[8 * half + e8] += sc[band_rd + (e8 & 3) + 8 * (e8 >> 2)];                                               \
        __builtin_amdgcn_wave_barrier();                                                                               \
    }
#define AX_FINISH_TILE(S0_, S1_, tt_, mx_)                                                                             \
    {                                                                                                                  \
        const int j0 = (tt_) * AX_BK;                                                                                  \
        const int dmin = iw0 - (j0 + AX_BK - 1);                                                                       \
        const int dmax = iw0 + 31 - j0;                                                                                \
        float cb = 0.f;                                                                                                \
        bool redo = false;                                                                                             \
        if (dmin >= kRelMax - 1) {                                                                                     \
            cb = c_past;                                                                                               \
        } else if (dmax <= -kRelMax) {                                                                                 \
            cb = c_future;                                                                                             \
        } else {                                                                                                       \
            float* sc = bias_stage[wave];                                                                              \
            const int lj = lane & 15, lq = lane >> 4;                                                                  \
            const int band_wr = 16 * lq + (lq >> 1) + lj, band_rd = 16 * r + (r >> 1) + 4 * h;                         \
            AX_BAND_ADD(S0_, 0)                                                                                        \
            AX_BAND_ADD(S1_, 1)                                                                                        \
            redo = true;                                                                                               \
        }                                                                                                              \
        if (j0 + AX_BK > nvalid) {                                                                                     \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                           \
                const int j = j0 + (e & 3) + 8 * (e >> 2) + 4 * h;                                                     \
                S0_[e] = j < nvalid ? S0_[e] : -INFINITY;                                                              \
                S1_[e] = j + 32 < nvalid ? S1_[e] : -INFINITY;                                                         \
            }                                                                                                          \
            redo = true;                                                                                               \
        }                                                                                                              \
        float mxv = (mx_);                                                                                             \
        if (redo) {                                                                                                    \
            mxv = S0_[0];                                                                                              \
            _Pragma("unroll") for (int e = 1; e < 16; ++e) mxv = fmaxf(mxv, S0_[e]);                                   \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) mxv = fmaxf(mxv, S1_[e]);                                   \
        }                                                                                                              \
        {   /* the other lane half holds the other 32 keys of this query: one permlane32 swap, no LDS permute.       \
               INLINE ASM ON TWO DISTINCT REGISTERS, deliberately: written as __builtin_amdgcn_permlane32_swap(mu, mu) hipcc    \
               (ROCm 7.2) folds fmaxf(result[0], result[1]) to result[0] -- it treats the two results of a swap of equal    \
               inputs as equal -- and every row then took the maximum of the LOWER lane half's 32 keys only.  Softmax is  \
               shift-invariant, so nothing showed until a key of the upper half beat that maximum by more than ~11 nats:  \
               P = exp(s - m) then leaves fp16, its hi plane is inf, lo = -inf and the row's context is NaN (found by the  \
               second golden weight family, whose outlier tokens do exactly that; tests/test_gpu_ops.py pins it).       \
               v_permlane32_swap a, b exchanges a[32..63] with b[0..31]: afterwards a holds the lower half's value in    \
               both halves and b the upper half's.  One wait state between the VALU write of the inputs and the swap     \
               (the compiler pads its own instructions, not inline asm). */                                            \
            float ma_ = mxv, mb_ = mxv;                                                                                \
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ma_), "+v"(mb_));                \
            mxv = fmaxf(ma_, mb_) + cb;                                                                                \
        }                                                                                                              \
        const float m_new = fmaxf(m_run, mxv);                                                                         \
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);                                                      \
        m_run = m_new;                                                                                                 \
        dsh = (cb - m_new) * kLog2e;                                                                                   \
    }

#ifdef LOCO_ATTN_STAMPS
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, tlast, tstart;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstart) :: "memory");
    tlast = tstart;
#endif
    // ---- prologue: K(0), V(0), K(1) in flight; S(0) = K(0) Q^T and its bookkeeping
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) AX_DMA_PIECE(pc, 0, 0, 0, 0)
    float c_past, c_future;  // Qp[i][319] (i - j >= 159) and Qp[i][0] (i - j <= -160): the bias of every key beyond the band
    if (TABLE) {
        // Qp^T block blk = pe_k[32 blk .. 32 blk + 31] Q^T: A = pe_k rows (fragment of lane (r, h) at k-step ks: pe_k[32 blk + r][16 ks + 8 h ..
        // + 7]), B = the Q fragments.  acc[e] = Qp[iq][32 blk + (e & 3) + 8 (e >> 2) + 4 h]: four consecutive table entries per e >> 2 ->
        // one 16-byte store.  The weight planes carry pe_k * 2^k (loco_api.hip, make_split): pe_scale = 2^-k restores it, exactly.
        //
        // pe_k reaches the matrix pipe THROUGH LDS, once per workgroup: 64 rows (two blocks, hi + lo planes = 16 KiB) have exactly the
        // shape of a K tile, so a chunk is DMA'd like one -- same swizzle, same fragment reads -- into the two ring slots the first
        // iteration does not need yet (K slot 1, V^T slot 1).  Loaded per wave straight from global memory (the first form of this
        // prologue) every wave pulled the whole 80 KiB through the L2: 1.5 GB per launch at 30 s x 32 beside 3.5 GB of K / V^T, and
        // 21 000 of a workgroup's 173 000 cycles (tools/attn_stamps.py with the timing-only build LOCO_ATTN_HACK=1).  Two rounds of
        // five blocks; a round's accumulators stay in registers until the NEXT round's DMAs have been issued, so that each wait
        // sees table stores and DMAs together (vmcnt counts both; they may complete out of order, hence vmcnt(0)).
        //
        // Only the blocks some key can read are formed (round 4).  The band tiles of this wave read column clip(i - j) + 160 for its 32
        // queries i and the keys j below the end of the last key tile: a range of at most ntiles * 64 + 31 columns around the diagonal.
        // At T = 1 499 that is the whole table for all but the first and last query blocks; at the 150 ... 300 frames of an utterance
        // (the packed forward's regime, where this prologue was 59 % of the workgroup's cycles: every workgroup forms, stores and
        // reads back its 160 KB at the same time) it is 5 - 8 of the 10 blocks.  The skipped blocks' MFMAs and stores are gone; the
        // pe_k planes still pass through LDS for the whole workgroup (other waves need other blocks).  c_future / c_past (columns 0 /
        // 319) are needed exactly when a key tile is clipped on that side, and then the range above reaches that end of the table.
        int bmin, bmax;
        {
            int lo = iw0 - (ntiles * AX_BK - 1), hi = iw0 + 31;
            lo = lo < -kRelMax ? -kRelMax : lo;
            hi = hi > kRelMax - 1 ? kRelMax - 1 : hi;
            bmin = (lo + kRelMax) >> 5;
            bmax = (hi + kRelMax) >> 5;
        }
        float cf = 0.f, cp = 0.f;
        unsigned pev[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int swz = (((2 * wave + i) & 3) << 1) | ((drow >> 1) & 1);
            pev[i] = (unsigned)(8 * (2 * wave + i) + drow) * (2u * kHeadDim) + 16u * (unsigned)(dpos ^ swz);
        }
        // a fifth block per round sits in the per-wave bias scratch (8.5 KiB, idle until the first band tile): 32 rows, wave w moves rows 8 w ..
        const unsigned bs0 = (unsigned)(unsigned long)(ax_lptr_t)&bias_stage[0][0];
        const unsigned pev1 = (unsigned)(8 * wave + drow) * (2u * kHeadDim) + 16u * (unsigned)(dpos ^ (((wave & 3) << 1) | ((drow >> 1) & 1)));
        constexpr int kBsPlane = 32 * kHeadDim;  // halves per plane of the single block
        // blocks b0_, b0_ + 1 -> ring slot slot_ (pe_k rows 32 b0_ .. + 63: the shape and swizzle of a K tile)
#define AX_PE_DMA2(b0_, slot_)                                                                                                     \
    _Pragma("unroll") for (int pc_ = 0; pc_ < 4; ++pc_) {                                                                          \
        const int i_ = (pc_ >> 1) & 1, pl_ = pc_ & 1;                                                                              \
        const unsigned d_ = lds0 + 2u * (unsigned)((slot_) * AX_STG + pl_ * AX_PL + 8 * (2 * wave + i_) * kHeadDim);               \
        AX_DMA16(reinterpret_cast<const char*>(pl_ ? pe_lo : pe_hi) + (long)(b0_) * (32 * kHeadDim * 2), pev[i_], d_);             \
    }
#define AX_PE_DMA1(b0_)                                                                                                            \
    _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_) {                                                                          \
        const unsigned d_ = bs0 + 2u * (unsigned)(pl_ * kBsPlane + 8 * wave * kHeadDim);                                           \
        AX_DMA16(reinterpret_cast<const char*>(pl_ ? pe_lo : pe_hi) + (long)(b0_) * (32 * kHeadDim * 2), pev1, d_);                \
    }
        constexpr int kBlocksPerRound = kRelN / 64;  // 5
        static_assert(2 * kBlocksPerRound * 32 == kRelN && sizeof(bias_stage) >= 2 * kBsPlane * sizeof(_Float16), "table rounds");
#define AX_PE_ROUND_DMA(rd_)                                                                      \
    AX_PE_DMA2(LOCO_ATTN_HACK == 1 ? 0 : kBlocksPerRound * (rd_), 1)                               \
    AX_PE_DMA2(LOCO_ATTN_HACK == 1 ? 0 : kBlocksPerRound * (rd_) + 2, 3)                           \
    AX_PE_DMA1(LOCO_ATTN_HACK == 1 ? 0 : kBlocksPerRound * (rd_) + 4)
        AX_PE_ROUND_DMA(0)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // this round's blocks have landed (and K(0) / V^T(0), and the previous round's table stores)
            f32x16 acc[kBlocksPerRound];
#pragma unroll
            for (int bb = 0; bb < kBlocksPerRound; ++bb) {
                const _Float16* pb = bb == 4 ? reinterpret_cast<const _Float16*>(&bias_stage[0][0]) : lds + ((bb >> 1) ? 3 : 1) * AX_STG;
                const int plane = bb == 4 ? kBsPlane : AX_PL;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[bb][e] = 0.f;
                const int blk_ = kBlocksPerRound * rd + bb;
                if (blk_ < bmin || blk_ > bmax) continue;  // wave-uniform: no key of this wave's band reads these 32 columns
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const h8 fh = *reinterpret_cast<const h8*>(pb + (bb & 1) * 32 * kHeadDim + fo[ks]);
                    const h8 fl = *reinterpret_cast<const h8*>(pb + plane + (bb & 1) * 32 * kHeadDim + fo[ks]);
                    acc[bb] = AX_MFMA(fl, qh[ks], acc[bb]);
                    acc[bb] = AX_MFMA(fh, ql[ks], acc[bb]);
                    acc[bb] = AX_MFMA(fh, qh[ks], acc[bb]);
                }
            }
            __syncthreads();  // every wave has read the three buffers: they may be refilled
            if (rd == 0) {
                AX_PE_ROUND_DMA(1)
            } else {
                const int t1 = ntiles > 1 ? 1 : 0;
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) AX_DMA_PIECE(pc, t1, 0, 1, 0)  // K(1): its slot is free now
            }
#pragma unroll
            for (int bb = 0; bb < kBlocksPerRound; ++bb) {
                const int blk = kBlocksPerRound * rd + bb;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[bb][e] *= pe_scale;
                if (blk == 0) cf = acc[bb][0];                    // table column 0: lane half 0
                if (blk == kRelN / 32 - 1) cp = acc[bb][15];      // table column 319 = 288 + 3 + 24 + 4: lane half 1
                if (iq < T && blk >= bmin && blk <= bmax && (LOCO_ATTN_HACK != 2 || T < 0)) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        *reinterpret_cast<float4*>(qprow + 32 * blk + 8 * g4 + 4 * h) =
                            make_float4(acc[bb][4 * g4], acc[bb][4 * g4 + 1], acc[bb][4 * g4 + 2], acc[bb][4 * g4 + 3]);
                }
            }
        }
#undef AX_PE_ROUND_DMA
#undef AX_PE_DMA2
#undef AX_PE_DMA1
        // both lane halves of a query need both constants: one swap each (after it the first register holds the lower half's
        // value in both halves, the second the upper half's -- see the row maximum below for why this is inline asm)
        float cf2 = cf, cp2 = cp;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(cf), "+v"(cf2));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(cp2), "+v"(cp));
        c_future = cf;  // lower half's
        c_past = cp;    // upper half's
        // the band tiles of THIS wave read these rows back (other lanes of it): the stores must have reached L2 first -- the
        // vmcnt(0) right below, which also retires K(1)'s DMAs
    } else {
        const int t1 = ntiles > 1 ? 1 : 0;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) AX_DMA_PIECE(pc, t1, 0, 1, 0)
        c_past = qprow[kRelN - 1];
        c_future = qprow[0];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 sa0, sa1, sb0, sb1;
    {
#pragma unroll
        for (int e = 0; e < 16; ++e) { sa0[e] = 0.f; sa1[e] = 0.f; }
        const _Float16* kb = lds;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h8 kh0 = AX_KF(kb, 0, ks, 0), kl0 = AX_KF(kb, 0, ks, 1), kh1 = AX_KF(kb, 1, ks, 0), kl1 = AX_KF(kb, 1, ks, 1);
            sa0 = AX_MFMA(kl0, qh[ks], sa0); sa1 = AX_MFMA(kl1, qh[ks], sa1);
            sa0 = AX_MFMA(kh0, ql[ks], sa0); sa1 = AX_MFMA(kh1, ql[ks], sa1);
            sa0 = AX_MFMA(kh0, qh[ks], sa0); sa1 = AX_MFMA(kh1, qh[ks], sa1);
        }
        float mx = sa0[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sa0[e]);
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sa1[e]);
        AX_BAND_LOAD(0)
        AX_FINISH_TILE(sa0, sa1, 0, mx)
    }
    __syncthreads();  // every wave has read K(0) before K(2) may land on it
    AX_STAMP(5)

    // ---- main loop, software-pipelined inside each wave so that the matrix pipe and the vector ALU always have
    //      independent work from the SAME wave (two waves per SIMD cannot be relied on to fall into anti-phase):
    //        block 1:  S(t+1) = K(t+1) Q^T   (24 MFMAs)  ||  exp2 + hi/lo split of S(t) keys 0..31, exp2 of keys 32..63
    //        block 2:  O += V(t)[0..31] P     (12 MFMAs)  ||  hi/lo split of keys 32..63
    //        block 3:  O += V(t)[32..63] P    (12 MFMAs)  ||  raw row max of S(t+1)
    //      K(t+2) and V(t+1) stream into the rings by LDS-DMA meanwhile.  SC = scores of tile t (consumed), SN = tile t+1.
#define AX_ITER(t_, SC0, SC1, SN0, SN1)                                                                                \
    {                                                                                                                  \
        const int tq = (t_);                                                                                           \
        const int tkn = tq + 2 < ntiles ? tq + 2 : ntiles - 1, tvn = tq + 1 < ntiles ? tq + 1 : ntiles - 1;             \
        if (tq + 1 < ntiles) { AX_BAND_LOAD(tq + 1) }                                                                  \
        const _Float16* kb = lds + ((tq + 1) & 1) * AX_STG;                                                            \
        const _Float16* vb = lds + (2 + (tq & 1)) * AX_STG;                                                            \
        h8 kf[2][4]; /* [buffer][hi0, lo0, hi1, lo1] */                                                                \
        kf[0][0] = AX_KF(kb, 0, 0, 0); kf[0][1] = AX_KF(kb, 0, 0, 1);                                                  \
        kf[0][2] = AX_KF(kb, 1, 0, 0); kf[0][3] = AX_KF(kb, 1, 0, 1);                                                  \
        const f32x2 al2 = {alpha, alpha}, k2 = {kLog2e, kLog2e}, d2 = {dsh, dsh};                                      \
        f32x2 ps2 = {0.f, 0.f};                                                                                        \
        /* LONG_SEQ (T >= 4 096, chosen by the launcher): once the running maximum has settled, alpha is exactly 1 in every lane for    \
           most tiles and the 32 multiplications are skipped -- bit-identical; interleaved A/B (tools/attn_long_ab.py): +0.0 % at       \
           T = 1 499, -0.3 % at 4 000, -0.8 % at 8 192, -1.6 % at 29 999.  Short sequences keep the instantiation without the test. */   \
        if (!LONG_SEQ || __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {                                            \
            _Pragma("unroll") for (int e = 0; e < 16; e += 2) {                                                        \
                f32x2 a_ = {o0[e], o0[e + 1]}, c_ = {o1[e], o1[e + 1]};                                                \
                a_ *= al2; c_ *= al2;                                                                                  \
                o0[e] = a_.x; o0[e + 1] = a_.y; o1[e] = c_.x; o1[e + 1] = c_.y;                                        \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) { SN0[e] = 0.f; SN1[e] = 0.f; }                                 \
        u32x4 ph0[2], pl0[2], ph1[2], pl1[2];                                                                          \
        AX_PIN();                                                                                                      \
        AX_STAMP(0)                                                                                                    \
        /* ---- block 1 */                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                             \
            if (ks + 1 < 4) {                                                                                          \
                kf[(ks + 1) & 1][0] = AX_KF(kb, 0, ks + 1, 0); kf[(ks + 1) & 1][1] = AX_KF(kb, 0, ks + 1, 1);          \
                kf[(ks + 1) & 1][2] = AX_KF(kb, 1, ks + 1, 0); kf[(ks + 1) & 1][3] = AX_KF(kb, 1, ks + 1, 1);          \
            }                                                                                                          \
            _Pragma("unroll") for (int m = 0; m < 6; ++m) {                                                            \
                const int st = m & 1, wh = m >> 1;                                                                     \
                const h8 a_ = kf[ks & 1][2 * st + (wh == 0 ? 1 : 0)];                                                  \
                const h8 b_ = wh == 1 ? ql[ks] : qh[ks];                                                               \
                if (st == 0) SN0 = AX_MFMA(a_, b_, SN0); else SN1 = AX_MFMA(a_, b_, SN1);                              \
                const int n = 6 * ks + m, q_ = n / 3;                                                                  \
                if (n % 3 == 0) {                                                                                      \
                    const int e0 = 2 * q_;                                                                             \
                    const f32x2 sx = {SC0[e0], SC0[e0 + 1]};                                                           \
                    const f32x2 ax = __builtin_elementwise_fma(sx, k2, d2);                                            \
                    const f32x2 pv = {__builtin_amdgcn_exp2f(ax.x), __builtin_amdgcn_exp2f(ax.y)};                     \
                    ps2 += pv;                                                                                         \
                    asm volatile("" : "+v"(ps2));                                                                      \
                    unsigned hi_, lo_;                                                                                 \
                    ax_split_pair(pv, hi_, lo_);                                                                       \
                    ph0[q_ >> 2][q_ & 3] = hi_; pl0[q_ >> 2][q_ & 3] = lo_;                                            \
                } else if (n % 3 == 1) {                                                                               \
                    const int e0 = 2 * q_;                                                                             \
                    const f32x2 sx = {SC1[e0], SC1[e0 + 1]};                                                           \
                    const f32x2 ax = __builtin_elementwise_fma(sx, k2, d2);                                            \
                    SC1[e0] = __builtin_amdgcn_exp2f(ax.x); SC1[e0 + 1] = __builtin_amdgcn_exp2f(ax.y);                \
                }                                                                                                      \
                if (n % 3 == 1 && (LOCO_ATTN_HACK != 3 || ((n / 3) & 1) == 0)) { /* K(t+2) into the slot of K(t), V(t+1) into the slot of V(t-1): both consumed */ \
                    AX_PIN();                                                                                          \
                    AX_DMA_PIECE(n / 3, tkn, tvn, tq & 1, (tq + 1) & 1)                                                \
                }                                                                                                      \
                AX_PIN();                                                                                              \
            }                                                                                                          \
        }                                                                                                              \
        AX_STAMP(1)                                                                                                    \
        /* ---- blocks 2 and 3: 8 groups (st, s2, dt) of 3 MFMAs; V^T fragments one group ahead */                     \
        h8 vf[2][2]; /* [buffer][hi, lo] */                                                                            \
        vf[0][0] = AX_VF(vb, 0, 0, 0); vf[0][1] = AX_VF(vb, 0, 0, 1);                                                  \
        float mxr = -INFINITY;                                                                                         \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                                \
            const int st = g >> 2, s2 = (g >> 1) & 1, dt = g & 1;                                                      \
            if (g + 1 < 8) {                                                                                           \
                const int c1 = (g + 1) >> 1, dt1 = (g + 1) & 1;                                                        \
                vf[(g + 1) & 1][0] = AX_VF(vb, dt1, c1, 0); vf[(g + 1) & 1][1] = AX_VF(vb, dt1, c1, 1);                \
            }                                                                                                          \
            const h8 vh = vf[g & 1][0], vl = vf[g & 1][1];                                                             \
            const h8 ph = __builtin_bit_cast(h8, st == 0 ? ph0[s2] : ph1[s2]);                                         \
            const h8 pl = __builtin_bit_cast(h8, st == 0 ? pl0[s2] : pl1[s2]);                                         \
            _Pragma("unroll") for (int m = 0; m < 3; ++m) {                                                            \
                const h8 a_ = m == 0 ? vl : vh;                                                                        \
                const h8 b_ = m == 1 ? pl : ph;                                                                        \
                if (dt == 0) o0 = AX_MFMA(a_, b_, o0); else o1 = AX_MFMA(a_, b_, o1);                                  \
                const int n = 3 * g + m;                                                                               \
                if (n < 8) { /* block 2: split keys 32..63 (already exponentiated in place) */                         \
                    const f32x2 pv = {SC1[2 * n], SC1[2 * n + 1]};                                                     \
                    ps2 += pv;                                                                                         \
                    asm volatile("" : "+v"(ps2));                                                                      \
                    unsigned hi_, lo_;                                                                                 \
                    ax_split_pair(pv, hi_, lo_);                                                                       \
                    ph1[n >> 2][n & 3] = hi_; pl1[n >> 2][n & 3] = lo_;                                                \
                } else if (n >= 12 && n < 20) { /* block 3: raw max of the next tile, 4 values per slot */             \
                    const int e0 = 4 * (n - 12);                                                                       \
                    if (e0 < 16) mxr = fmaxf(fmaxf(fmaxf(mxr, SN0[e0]), fmaxf(SN0[e0 + 1], SN0[e0 + 2])), SN0[e0 + 3]); \
                    else mxr = fmaxf(fmaxf(fmaxf(mxr, SN1[e0 - 16]), fmaxf(SN1[e0 - 15], SN1[e0 - 14])), SN1[e0 - 13]); \
                    asm volatile("" : "+v"(mxr));                                                                      \
                }                                                                                                      \
                AX_PIN();                                                                                              \
            }                                                                                                          \
        }                                                                                                              \
        l_run = l_run * alpha + (ps2.x + ps2.y);                                                                       \
        AX_STAMP(2)                                                                                                    \
        if (tq + 1 < ntiles) { AX_FINISH_TILE(SN0, SN1, tq + 1, mxr) }                                                 \
        AX_DBG(tq, 1u, SN0, SN1)                                                                                       \
        AX_STAMP(3)                                                                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
        __syncthreads();                                                                                               \
        AX_STAMP(4)                                                                                                    \
    }

    for (int t = 0; t < ntiles; t += 2) {
        AX_ITER(t, sa0, sa1, sb0, sb1)
        if (t + 1 >= ntiles) break;
        AX_ITER(t + 1, sb0, sb1, sa0, sa1)
    }
#undef AX_ITER
#undef AX_FINISH_TILE
#undef AX_BAND_LOAD
#undef AX_BAND_ADD
#undef AX_IS_BAND
#undef AX_DMA_PIECE
#undef AX_DMA16
#undef AX_KF
#undef AX_VF
#undef AX_VTR

#ifdef LOCO_ATTN_STAMPS
    if (g_attn_stamps && tid == 0) {
        unsigned long long* o_ = g_attn_stamps + 8l * blockIdx.x;
        for (int k_ = 0; k_ < 6; ++k_) o_[k_] = seg[k_];
        o_[6] = (unsigned long long)ntiles;
        o_[7] = tlast - tstart;
    }
#endif
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
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                unsigned h0, l0, h1, l1;
                split_f16_2pairs(a[0], a[1], a[2], a[3], h0, l0, h1, l1);
                *reinterpret_cast<u32x2*>(ctx_hi + obase + 8 * g4) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(ctx_lo + obase + 8 * g4) = u32x2{l0, l1};
                split_f16_2pairs(c[0], c[1], c[2], c[3], h0, l0, h1, l1);
                *reinterpret_cast<u32x2*>(ctx_hi + obase + 32 + 8 * g4) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(ctx_lo + obase + 32 + 8 * g4) = u32x2{l0, l1};
            } else {
                *reinterpret_cast<float4*>(ctx + obase + 8 * g4) = make_float4(a[0], a[1], a[2], a[3]);
                *reinterpret_cast<float4*>(ctx + obase + 32 + 8 * g4) = make_float4(c[0], c[1], c[2], c[3]);
            }
        }
    }
}

// A/B and test knob: LOCO_ATTN_LONG=0 / 1 forces the instantiation without / with the long-sequence rescale skip (read once, and again
// on loco_debug_reload_gemm_knobs); unset = chosen by T.
static int read_attn_long_knob() {
    const char* v = getenv("LOCO_ATTN_LONG");
    return v ? (atoi(v) != 0) : -1;
}
static int& attn_long_knob() {
    static int k = read_attn_long_knob();
    return k;
}
void reload_attention_knobs() { attn_long_knob() = read_attn_long_knob(); }

hipError_t launch_attention_f16x3(const _Float16* qhi, const _Float16* qlo, const _Float16* khi, const _Float16* klo,
                                  const _Float16* vhi, const _Float16* vlo, const float* qp, const int32_t* frames,
                                  _Float16* ctx_hi, _Float16* ctx_lo, float* ctx, int B, int T, hipStream_t s,
                                  const _Float16* pe_hi, const _Float16* pe_lo, float pe_scale) {
    if (B <= 0 || T <= 0 || B > 65535) return hipErrorInvalidValue;
    const int nqb = (T + AX_BQ - 1) / AX_BQ;
    const long nblk = (long)nqb * kHeads * B;
    if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblk);
    float* qpw = const_cast<float*>(qp);  // written only in the pe_hi != nullptr form (the table is then an output / scratch of this launch)
    if ((pe_hi == nullptr) != (pe_lo == nullptr)) return hipErrorInvalidValue;
    const int forced = attn_long_knob();
    const bool long_seq = forced < 0 ? T >= 4096 : forced != 0;
#define AX_LAUNCH1(SPLIT_, TABLE_, LONG_)                                                                                              \
    hipLaunchKernelGGL((attention_f16x3_kernel<SPLIT_, TABLE_, LONG_>), grid, dim3(256), 0, s, qhi, qlo, khi, klo, vhi, vlo, qpw, frames, ctx_hi, \
                       ctx_lo, ctx, T, nqb, pe_hi, pe_lo, pe_scale)
#define AX_LAUNCH(SPLIT_, TABLE_)                    \
    {                                                \
        if (long_seq) AX_LAUNCH1(SPLIT_, TABLE_, true); \
        else AX_LAUNCH1(SPLIT_, TABLE_, false);      \
    }
    if (ctx_hi && pe_hi) AX_LAUNCH(true, true)
    else if (ctx_hi) AX_LAUNCH(true, false)
    else if (pe_hi) AX_LAUNCH(false, true)
    else AX_LAUNCH(false, false)
#undef AX_LAUNCH
#undef AX_LAUNCH1
    return hipGetLastError();
}

}  // namespace loco

#ifdef LOCO_ATTN_DEBUG
extern "C" int loco_debug_set_attn_dbg(void* buf) {  // diagnostic build only
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(loco::g_attn_dbg), &buf, sizeof(buf));
}
#endif
#ifdef LOCO_ATTN_STAMPS
extern "C" int loco_debug_set_attn_stamps(void* buf) {  // diagnostic build only: 8 x u64 per workgroup of the next launches
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(loco::g_attn_stamps), &buf, sizeof(buf));
}
#endif

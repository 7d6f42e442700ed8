// C ABI (include/loco_asr.h) and the host-side orchestration of the encoder forward.
//
// One forward = SpeechT5EncoderWithSpeechPrenet.forward in eval mode (HF modeling_speecht5.py:1339-1358):
//   prenet (HF :534-566): conv0+GroupNorm+GELU -> 6 x (conv as GEMM + GELU) -> LayerNorm(512) -> Linear(512,768)
//                         -> + GELU(pos-conv) + sinusoid
//   encoder (HF :1234-1322): LayerNorm(768) -> 12 x [ fused QKV GEMM, Qp GEMM, flash attention, out-proj GEMM(+x),
//                            LayerNorm, FFN1 GEMM(+GELU), FFN2 GEMM(+h), LayerNorm ]
// Everything is enqueued on the caller's stream from a caller-owned workspace; no allocation, host
// synchronisation or thread is used inside loco_forward (the one exception, growing the sinusoid table
// past its reserved rows, mirrors HF's own auto-grow at modeling:331-333 and is done before any launch).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/loco_asr.h"
#include "loco_kernels.h"

using namespace loco;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(LOCO_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct Tensor {
    float* d = nullptr;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

const int kConvK[7] = {10, 3, 3, 3, 3, 2, 2};
const int kConvS[7] = {5, 2, 2, 2, 2, 2, 2};

struct SplitW {  // fp16 hi/lo planes of one GEMM weight (precision mode f16x3), holding W * 2^k; inv_scale = 2^-k
    _Float16* hi = nullptr;
    _Float16* lo = nullptr;
    float inv_scale = 1.0f;
};

struct LayerW {
    float* wqkv = nullptr;  // [2304,768], q rows pre-scaled by 1/8
    float* bqkv = nullptr;  // [2304]
    SplitW sqkv, so, s1, s2;
};

// Profiling buckets, named after the kernel that runs in them (the two precision modes have their own attention and
// positional-conv buckets: the f16x3 positional conv IS a gemm_f16x3_dma_kernel launch, but keeps a bucket of its own because
// its shape -- N = 48, K = 6144, halo layout -- has little in common with the projection GEMMs).
enum KernelId { K_GEMM = 0, K_ATTN, K_LN, K_CONV0, K_POSCONV, K_FRAMES, K_COPY, K_GEMM_SPLIT, K_ATTN_SPLIT, K_POSCONV_SPLIT, K_QP, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"gemm_f32",     "attention_f32", "layernorm",       "conv0_gn_gelu",
                                           "pos_conv_f32", "frame_counts",  "copy",            "gemm_f16x3",
                                           "attention_f16x3", "pos_conv_f16x3_gemm", "qp_table_gemm_f16x3"};

struct ProfRec {
    hipEvent_t a, b;
    int kid;
    double flops, bytes;
};

// Status block of ONE forward (include/loco_asr.h, loco_status_bytes): host memory, caller-owned for loco_forward_async, the
// handle's own pinned block for loco_forward.  The host part is filled while the forward is enqueued; `words` is the target of the
// device-to-host copy that follows the forward on its stream.
constexpr uint32_t kStatusMagic = 0x53434f4cu;  // "LOCS"
constexpr int kMaxPackClips = 512;              // clips per loco_forward_packed
struct StatusBlock {
    uint32_t magic;
    int32_t precision;                        // arithmetic mode of the forward this block describes
    int32_t used;                             // stages filled
    int32_t layer[kRangeMaxStages];
    const char* names[kRangeMaxStages];       // static strings of this library
    char msg[384];                            // non-empty: a range verdict known on the host (weights outside the planes' range)
    float words[kRangeMaxStages * kRangeShards];
    // loco_forward_packed: per clip, the conv-layer-0 and the encoder frame counts of its OWN reference batch's padded length
    // ([0, B): conv0 frames, [B, 2B): encoder frames, [2B, 3B): valid frames when the caller gave valid_len instead of a mask).
    // Staged here because the block is host memory the caller keeps alive (and pinned) until the stream has completed the forward:
    // the source of the host-to-device copy that opens the forward.
    int32_t clip_tab[3 * kMaxPackClips];
};
// device side of the same: the first bytes of every workspace
constexpr size_t kStatusDevBytes = (sizeof(float) * kRangeMaxStages * kRangeShards + 255) & ~size_t(255);

// Per-call state of one loco_forward*: everything the enqueue mutates lives here, not in the handle, so that forwards of one
// handle may be enqueued from several host threads and be in flight together (different streams, workspaces and status blocks).
struct Call {
    int precision = 1;
    float* splitk = nullptr;     // split-K workspace of the (half-)batch being enqueued (null for large problems)
    bool dual = false;           // inside the two half-batch schedule (GemmSplitArgs::co_scheduled)
    float* range_dev = nullptr;  // [kRangeMaxStages][kRangeShards] in the workspace, zeroed at the start of the forward
    StatusBlock* st = nullptr;
};

}  // namespace

struct loco_encoder {
    loco_config cfg;
    int device = 0;
    std::map<std::string, Tensor> raw;  // as loaded, HF names
    std::map<std::string, std::vector<int64_t>> expected;
    bool finalized = false;
    // prepared
    float* conv_w[7] = {nullptr};  // [512, k*512] tap-major (layer 0 stays [512,10])
    float* pos_w = nullptr;        // [16][128][48][48]
    SplitW conv_s[7];              // split copies of conv_w[1..6]
    SplitW proj_s;                 // feature projection
    SplitW pe_s;                   // relative-position table pe_k [320,64]
    SplitW posg_s;                 // positional conv weight as the GEMM's W: [16][48][128*48]
    int precision = 1;             // 0 = exact fp32 MFMA, 1 = fp16 x3 split MFMA (default), 2 = f16x2 (weights rounded to fp16; opt-in)
    std::vector<LayerW> layers;
    // published with release stores (table first, then its row count), read with acquire loads: a forward on another host thread
    // that sees the new row count also sees the new, longer table (ensure_sin_rows; loco_forward_async allows concurrent callers)
    std::atomic<float*> sin_tab{nullptr};
    std::atomic<int> sin_rows{0};
    bool sin_user = false;
    std::mutex sin_mu;            // growth of the sinusoid table (ensure_sin_rows)
    std::vector<float*> retired;  // tables replaced while forwards may still have been reading them: freed at loco_destroy
    // taps
    float *tap_conv = nullptr, *tap_proj = nullptr, *tap_prenet = nullptr;
    // text front end (SpeechT5TextEncoderPrenet): optional; a handle may carry the speech prenet, the text prenet or both
    float* text_embed = nullptr;  // [text_vocab, 768]
    int text_vocab = 0;
    float* text_alpha = nullptr;  // [1]
    float* text_pe = nullptr;     // [text_pe_rows, 768]
    int text_pe_rows = 0;
    bool speech_ready = false;    // set by loco_finalize_weights when the speech prenet weights were supplied
    // concurrency inside one forward: a batch may run as two half-batches on two streams (loco_set_streams).  The side stream and
    // its events are created on first use under side_mu; forwards in flight together serialise their second halves on it.
    int streams = 2;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::mutex side_mu;
    // range tracking of the fp16-plane activations (loco_kernels.h, range_commit): one status word x 8 shards per stage, in the
    // workspace of each forward; `own` is the status block of the forwards enqueued through loco_forward (pinned host memory)
    StatusBlock* own = nullptr;
    float* absmax_dev = nullptr;  // one float, weight preparation
    bool debug_nonfinite = false;                 // LOCO_DEBUG_NONFINITE=1 at loco_create (dbg_check)
    unsigned long long* debug_counter = nullptr;  // two words, device
    int range_policy = 1;         // loco_forward_checked: 1 = re-run out-of-range batches on the exact-fp32 kernels, 0 = report
    std::string range_static;     // non-empty: a weight-determined plane tensor (a LayerNorm output) leaves the range
    // profiling
    bool profiling = false;
    int profile_only = -1;  // >= 0: only launches of this kernel bucket are bracketed (loco_set_profiling_filter)
    std::vector<ProfRec> recs;
    size_t recs_used = 0;
    loco_kernel_stat stats[K_COUNT];
};

namespace {

void build_expected(loco_encoder* e) {
    auto& x = e->expected;
    const int64_t H = e->cfg.hidden, C = e->cfg.conv_dim, F = e->cfg.ffn;
    const std::string p = "prenet.";
    x[p + "masked_spec_embed"] = {H};
    for (int i = 0; i < 7; ++i)
        x[p + "feature_encoder.conv_layers." + std::to_string(i) + ".conv.weight"] = {C, i == 0 ? 1 : C, kConvK[i]};
    x[p + "feature_encoder.conv_layers.0.layer_norm.weight"] = {C};
    x[p + "feature_encoder.conv_layers.0.layer_norm.bias"] = {C};
    x[p + "feature_projection.layer_norm.weight"] = {C};
    x[p + "feature_projection.layer_norm.bias"] = {C};
    x[p + "feature_projection.projection.weight"] = {H, C};
    x[p + "feature_projection.projection.bias"] = {H};
    x[p + "pos_conv_embed.conv.bias"] = {H};
    x[p + "pos_conv_embed.conv.parametrizations.weight.original0"] = {1, 1, e->cfg.pos_conv_kernel};
    x[p + "pos_conv_embed.conv.parametrizations.weight.original1"] = {H, H / e->cfg.pos_conv_groups, e->cfg.pos_conv_kernel};
    const std::string w = "wrapped_encoder.";
    x[w + "layer_norm.weight"] = {H};
    x[w + "layer_norm.bias"] = {H};
    x[w + "embed_positions.pe_k.weight"] = {2 * e->cfg.rel_max, H / e->cfg.heads};
    for (int l = 0; l < e->cfg.layers; ++l) {
        const std::string b = w + "layers." + std::to_string(l) + ".";
        for (const char* pr : {"q_proj", "k_proj", "v_proj", "out_proj"}) {
            x[b + "attention." + pr + ".weight"] = {H, H};
            x[b + "attention." + pr + ".bias"] = {H};
        }
        for (const char* ln : {"layer_norm", "final_layer_norm"}) {
            x[b + ln + ".weight"] = {H};
            x[b + ln + ".bias"] = {H};
        }
        x[b + "feed_forward.intermediate_dense.weight"] = {F, H};
        x[b + "feed_forward.intermediate_dense.bias"] = {F};
        x[b + "feed_forward.output_dense.weight"] = {H, F};
        x[b + "feed_forward.output_dense.bias"] = {H};
    }
}

bool optional_key(const std::string& k) { return k == "prenet.masked_spec_embed"; }

std::string canonical_key(const char* key) {
    std::string k(key);
    const std::string a = "pos_conv_embed.conv.weight_g", b = "pos_conv_embed.conv.weight_v";
    size_t pos;
    if ((pos = k.find(a)) != std::string::npos && pos + a.size() == k.size())
        k.replace(pos, a.size(), "pos_conv_embed.conv.parametrizations.weight.original0");
    else if ((pos = k.find(b)) != std::string::npos && pos + b.size() == k.size())
        k.replace(pos, b.size(), "pos_conv_embed.conv.parametrizations.weight.original1");
    return k;
}

const float* W(const loco_encoder* e, const std::string& k) { return e->raw.at(k).d; }

size_t align_up(size_t n) { return (n + 255) & ~size_t(255); }

struct Plan {
    int B;
    long L;
    long Tc[7];  // conv output lengths
    long T, M;
    size_t off_frames, off_clip, off_c0scratch, off_a, off_b, off_x0, off_x1, off_tmp, off_ctx, off_qkv, off_qp, off_ffn, off_xs0, off_xs1,
        off_splitk, total;
    bool splitk;
};

void carve_plan(const loco_encoder* e, Plan& p);

bool make_plan(const loco_encoder* e, int B, long L, Plan& p) {
    p.B = B;
    p.L = L;
    long n = L;
    for (int i = 0; i < 7; ++i) {
        n = conv_out_len(n, kConvK[i], kConvS[i]);
        p.Tc[i] = n;
    }
    if (B <= 0 || n <= 0) return false;
    p.T = n;
    p.M = (long)B * n;
    carve_plan(e, p);
    return true;
}

// the text front end enters the encoder with T tokens per sequence: no waveform, no conv buffers
bool make_plan_tokens(const loco_encoder* e, int B, long T, Plan& p) {
    p.B = B;
    p.L = 0;
    for (int i = 0; i < 7; ++i) p.Tc[i] = 0;
    if (B <= 0 || T <= 0) return false;
    p.T = T;
    p.M = (long)B * T;
    carve_plan(e, p);
    return true;
}

void carve_plan(const loco_encoder* e, Plan& p) {
    const int B = p.B;
    const size_t f = sizeof(float);
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o += align_up(bytes);
        return at;
    };
    p.off_frames = take((size_t)B * sizeof(int32_t));
    p.off_clip = take((size_t)3 * B * sizeof(int32_t));  // packed forward: per-clip conv0 / encoder frame counts (StatusBlock::clip_tab)
    p.off_c0scratch = take(conv0_scratch_bytes(B));
    p.off_a = take((size_t)B * p.Tc[0] * kConvDim * f);
    p.off_b = take((size_t)B * p.Tc[1] * kConvDim * f);
    p.off_x0 = take((size_t)p.M * kHidden * f);
    p.off_x1 = take((size_t)p.M * kHidden * f);
    p.off_tmp = take((size_t)p.M * kHidden * f);
    p.off_ctx = take((size_t)p.M * kHidden * f);
    // fp32 [M,2304], or (precision f16x3) q, k and v as fp16 hi|lo planes [M,768] each: the same bytes
    p.off_qkv = take((size_t)3 * p.M * kHidden * f);
    p.off_qp = take((size_t)p.M * kHeads * kRelN * f);
    // FFN intermediate [M,3072]; doubles as scratch for the group-major positional-conv operand [B,16,T+128,48] x 2 planes
    const size_t ffn_elems = (size_t)p.M * e->cfg.ffn, posg_elems = (size_t)B * (p.T + kPosK) * kHidden;
    p.off_ffn = take((ffn_elems > posg_elems ? ffn_elems : posg_elems) * f);
    p.off_xs0 = take((size_t)p.M * kHidden * f);  // fp16 hi|lo planes of x0 / x1 (precision f16x3)
    p.off_xs1 = take((size_t)p.M * kHidden * f);
    p.splitk = p.M <= kSplitKMaxM;  // small problems: fp32 partial sums of the split-K GEMM path (gemm_f16x3.hip)
    p.off_splitk = take(p.splitk ? kSplitKBytes : 0);
    p.total = o;
}

// ---- profiling brackets -----------------------------------------------------------------------------
struct Bracket {
    loco_encoder* e;
    hipStream_t s;
    ProfRec* rec = nullptr;
    Bracket(loco_encoder* enc, hipStream_t st, int kid, double flops, double bytes) : e(enc), s(st) {
        if (!e->profiling || (e->profile_only >= 0 && e->profile_only != kid)) return;
        if (e->recs_used == e->recs.size()) {
            ProfRec r{};
            if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
            e->recs.push_back(r);
        }
        rec = &e->recs[e->recs_used++];
        rec->kid = kid;
        rec->flops = flops;
        rec->bytes = bytes;
        (void)hipEventRecord(rec->a, s);
    }
    ~Bracket() {
        if (rec) (void)hipEventRecord(rec->b, s);
    }
};

int run_gemm(loco_encoder* e, hipStream_t s, const float* A, long lda, const float* Wt, long ldw, const float* bias,
             const float* R, long ldr, float* C, long ldc, int M, int N, int K, int epi, int nb1 = 1, int nb2 = 1,
             long sA1 = 0, long sA2 = 0, long sC1 = 0, long sC2 = 0) {
    GemmArgs a{A, Wt, bias, R, C, M, N, K, lda, ldw, ldc, ldr, nb1, nb2, sA1, sA2, sC1, sC2, epi};
    const double nb = (double)nb1 * nb2;
    const double flops = 2.0 * M * (double)N * K * nb;
    const double bytes = 4.0 * (nb * ((double)M * K + (double)M * N * (epi == kEpiResidual ? 2 : 1)) + (double)N * K);
    Bracket br(e, s, K_GEMM, flops, bytes);
    HIP_TRY(launch_gemm(a, s));
    return LOCO_OK;
}

int run_ln(loco_encoder* e, hipStream_t s, const float* x, const float* g, const float* b, float* y, long rows, int dim,
           _Float16* yhi = nullptr, _Float16* ylo = nullptr, float* nonfinite_slot = nullptr) {
    Bracket br(e, s, K_LN, 8.0 * rows * dim, (y && yhi ? 12.0 : 8.0) * rows * dim);
    HIP_TRY(launch_layernorm(x, g, b, y, rows, dim, e->cfg.ln_eps, s, yhi, ylo, nonfinite_slot));
    return LOCO_OK;
}

// split-precision GEMM: A and W as fp16 hi/lo planes; output fp32 (C) or planes (Chi/Clo)
int run_gemm_split(loco_encoder* e, const Call& c, hipStream_t s, const _Float16* Ahi, const _Float16* Alo, long lda, const SplitW& Wt, long ldw,
                   const float* bias, const float* R, long ldr, float* C, _Float16* Chi, _Float16* Clo, long ldc, int M, int N, int K,
                   int epi, int nb1 = 1, long sA1 = 0, long sC1 = 0, int nb2 = 1, long sA2 = 0, long sC2 = 0,
                   const GemmSplitArgs* scatter = nullptr, const _Float16* Rhi = nullptr, const _Float16* Rlo = nullptr,
                   float* range_slot = nullptr, int kid = K_GEMM_SPLIT, int ktaps = 1) {
    GemmSplitArgs a{Ahi, Alo, Wt.hi, Wt.lo, bias, R, C, Chi, Clo, M, N, K, lda, ldw, ldc, ldr, nb1, nb2, sA1, sA2, sC1, sC2, epi};
    a.Rhi = Rhi;
    a.Rlo = Rlo;
    a.out_scale = Wt.inv_scale;
    a.range_slot = range_slot;
    a.co_scheduled = c.dual;
    a.ktaps = ktaps;
    a.terms = (c.precision == 2 && kid != K_QP) ? 2 : 3;  // the relative-position table keeps all three terms (K = 64: it costs nothing)
    if (scatter) {
        a.qkv_stride = scatter->qkv_stride;
        a.T = scatter->T;
    }
    a.splitk_ws = c.splitk;
    const double nb = (double)nb1 * nb2;
    const double flops = 2.0 * M * (double)N * K * nb;
    const double bytes = 4.0 * (nb * ((double)M * K + (double)M * N * (epi == kEpiResidual ? 2 : 1)) + (double)N * K);
    Bracket br(e, s, kid, flops, bytes);
    HIP_TRY(launch_gemm_split(a, s));
    return LOCO_OK;
}

// Weight planes: W * 2^k with k such that max|W| lands in [2^13, 2^14) -- the largest element then sits two binades under
// fp16's maximum and elements down to 2^-27 of it still have a normal fp16 lo part, whatever the tensor's absolute level
// (a checkpoint with weights of 1e-6 or of 1e+3 is represented exactly as well as one with weights of 0.03).  The GEMM
// multiplies its accumulator by inv_scale = 2^-k (exact).  One tiny reduction + one host read per tensor, at load time only.
int make_split(loco_encoder* e, SplitW& w, const float* src, size_t n, hipStream_t s) {
    if (!w.hi) HIP_TRY(hipMalloc(&w.hi, n * sizeof(_Float16)));
    if (!w.lo) HIP_TRY(hipMalloc(&w.lo, n * sizeof(_Float16)));
    HIP_TRY(hipMemsetAsync(e->absmax_dev, 0, sizeof(float), s));
    HIP_TRY(launch_absmax(src, (long)n, e->absmax_dev, s));
    float amax = 0.f;
    HIP_TRY(hipMemcpyAsync(&amax, e->absmax_dev, sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    int k = 0;
    if (amax > 0.f && amax < INFINITY) {
        int ex = 0;
        (void)frexpf(amax, &ex);  // amax = f * 2^ex, f in [0.5, 1)  ->  amax * 2^(14 - ex) in [2^13, 2^14)
        k = 14 - ex;
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
    }
    w.inv_scale = ldexpf(1.0f, -k);
    HIP_TRY(launch_split_f16(src, w.hi, w.lo, (long)n, s, ldexpf(1.0f, k)));
    return LOCO_OK;
}

int absmax_host(loco_encoder* e, const float* src, size_t n, hipStream_t s, float& out) {
    HIP_TRY(hipMemsetAsync(e->absmax_dev, 0, sizeof(float), s));
    HIP_TRY(launch_absmax(src, (long)n, e->absmax_dev, s));
    HIP_TRY(hipMemcpyAsync(&out, e->absmax_dev, sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LOCO_OK;
}

constexpr float kRangeHi = 65504.0f;   // fp16 maximum: hi = fp16(x) is inf from 65520 on
constexpr float kRangeLo = 0.015625f;  // 2^-6: a plane tensor whose LARGEST element is below this has lost fp32-class accuracy

// Plane tensors whose range follows from the weights: y = x_hat * gamma + beta with sum(x_hat^2) <= D gives
// |y| <= sqrt(D) max|gamma| + max|beta| (LayerNorm over D channels), and a unit-variance x_hat puts the tensor's largest element near max(max|gamma|, max|beta|).  The attention
// context is a convex combination of V rows, tracked with q|k|v.  Checked once per weight load; nothing is measured at run time.
int static_range_check(loco_encoder* e, const std::string& ln_prefix, int dim, hipStream_t s) {
    float g = 0.f, b = 0.f;
    int rc = absmax_host(e, W(e, ln_prefix + "weight"), (size_t)dim, s, g);
    if (!rc) rc = absmax_host(e, W(e, ln_prefix + "bias"), (size_t)dim, s, b);
    if (rc) return rc;
    const float hi = sqrtf((float)dim) * g + b, lo = fmaxf(g, b);
    if (e->range_static.empty() && (!(hi < kRangeHi) || lo < kRangeLo)) {
        char msg[320];
        snprintf(msg, sizeof msg, "activation range: the output of '%s*' (max|weight| = %.6g, max|bias| = %.6g) is %s the range precision mode "
                 "f16x3 represents to fp32 class (%g <= max|x| < %g)", ln_prefix.c_str(), (double)g, (double)b,
                 !(hi < kRangeHi) ? "not bounded inside" : "below", (double)kRangeLo, (double)kRangeHi);
        e->range_static = msg;
    }
    return LOCO_OK;
}

bool nl_is_zero(const loco_encoder* e) { return e->cfg.layers == 0; }

// Diagnostics, off unless the environment had LOCO_DEBUG_NONFINITE=1 when the handle was created: after each stage of the f16x3
// forward, count the non-finite elements of what it wrote (host synchronisation per stage!) and name the FIRST stage that produced
// any on stderr.  The range words cannot see NaNs (fmaxf drops them); this can.
int dbg_check(loco_encoder* e, hipStream_t s, const char* stage, int layer, const void* p, size_t n, bool half, long row_len) {
    if (!e->debug_nonfinite || !p) return LOCO_OK;
    unsigned long long h[2] = {0, ~0ull};
    HIP_TRY(hipMemcpyAsync(e->debug_counter, h, sizeof h, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_count_nonfinite(p, (long)n, half, e->debug_counter, s));
    HIP_TRY(hipMemcpyAsync(h, e->debug_counter, sizeof h, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h[0])
        fprintf(stderr, "[loco debug] non-finite values: %llu of %zu elements written by '%s' (layer %d); first at element %llu = row %llu, column %llu\n",
                h[0], n, stage, layer, h[1], row_len ? h[1] / row_len : 0ull, row_len ? h[1] % row_len : 0ull);
    return LOCO_OK;
}

int run_copy(loco_encoder* e, hipStream_t s, float* dst, const float* src, size_t n) {
    Bracket br(e, s, K_COPY, 0.0, 8.0 * n);
    HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return LOCO_OK;
}

// On success *tab is a table of at least `rows` rows: the snapshot this forward uses from here on, whatever other threads publish later.
int ensure_sin_rows(loco_encoder* e, int rows, hipStream_t s, const float** tab) {
    if (e->sin_rows.load(std::memory_order_acquire) >= rows) {
        // the count is stored AFTER its table: a table loaded now is that one or a later, longer one
        *tab = e->sin_tab.load(std::memory_order_acquire);
        return LOCO_OK;
    }
    // grow (HF does the same on demand, modeling:331-333); happens once per new maximum length.  Safe with other forwards of the handle
    // in flight: the new table is complete before it is published (this is the one place a forward may block its host thread), and
    // the old one is RETIRED, not freed -- forwards already enqueued on other streams keep reading it -- until loco_destroy.
    std::lock_guard<std::mutex> lock(e->sin_mu);
    if (e->sin_rows.load(std::memory_order_relaxed) >= rows) {
        *tab = e->sin_tab.load(std::memory_order_relaxed);
        return LOCO_OK;
    }
    int want = rows < 4002 ? 4002 : rows + 2;
    float* nt = nullptr;
    HIP_TRY(hipMalloc(&nt, (size_t)want * kHidden * sizeof(float)));
    hipError_t err = launch_sinusoid_table(nt, want, s);
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    if (err != hipSuccess) {
        (void)hipFree(nt);
        return fail(LOCO_E_HIP, "sinusoid table: %s", hipGetErrorString(err));
    }
    if (float* old = e->sin_tab.load(std::memory_order_relaxed)) e->retired.push_back(old);
    e->sin_tab.store(nt, std::memory_order_release);
    e->sin_rows.store(want, std::memory_order_release);
    *tab = nt;
    e->sin_user = false;
    return LOCO_OK;
}


struct Bufs {
    int32_t* frames;
    const int32_t* frames_or_null;
    float *bufA, *bufB, *x0, *x1, *tmp, *ctx, *qkv, *qp, *ffn;
    _Float16 *xs0, *xs1;
    char* c0scratch;
    // packed forward (null otherwise): per clip, the conv-layer-0 frame count and the encoder frame count of its own reference batch
    const int32_t* t0_clip = nullptr;
    const int32_t* rows_clip = nullptr;
    const float* sin_tab = nullptr;  // the sinusoid table this forward reads (a snapshot: ensure_sin_rows)
};

// ---- precision 0: every contraction on the exact-fp32 MFMA ------------------------------------------------------
// skip_prenet: x0 already holds the encoder's input hidden states (the text front end wrote them)
int forward_f32(loco_encoder* e, const Plan& p, const float* wav, float* out, float* const* hidden_states, const Bufs& bf,
                hipStream_t s, bool skip_prenet = false) {
    const int B = p.B;
    const long L = p.L;
    const int T = (int)p.T;
    const long M = p.M;
    int rc;
    float *bufA = bf.bufA, *bufB = bf.bufB, *x0 = bf.x0, *x1 = bf.x1, *tmp = bf.tmp, *ctx = bf.ctx, *qkv = bf.qkv, *qp = bf.qp,
          *ffn = bf.ffn;
    const int32_t* frames_or_null = bf.frames_or_null;
    const std::string pn = "prenet.", we = "wrapped_encoder.";
    if (!skip_prenet) {
    // ---- feature encoder (HF :484-494)
    {
        const double outb = 4.0 * B * (double)p.Tc[0] * kConvDim;
        Bracket br(e, s, K_CONV0, 2.0 * 10 * B * (double)p.Tc[0] * kConvDim, outb + 8.0 * B * (double)L);
        HIP_TRY(launch_conv0_gn_gelu(wav, B, L, e->conv_w[0], W(e, pn + "feature_encoder.conv_layers.0.layer_norm.weight"),
                                     W(e, pn + "feature_encoder.conv_layers.0.layer_norm.bias"), bufA,
                                     bf.c0scratch, e->cfg.ln_eps, s, nullptr, nullptr, nullptr, bf.t0_clip));
    }
    float* cin = bufA;
    float* cout = bufB;
    for (int i = 1; i < 7; ++i) {
        const long Tin = p.Tc[i - 1], Tout = p.Tc[i];
        rc = run_gemm(e, s, cin, (long)kConvS[i] * kConvDim, e->conv_w[i], (long)kConvK[i] * kConvDim, nullptr, nullptr, 0,
                      cout, kConvDim, (int)Tout, kConvDim, kConvK[i] * kConvDim, kEpiGelu, B, 1, Tin * kConvDim, 0,
                      Tout * kConvDim, 0);
        if (rc) return rc;
        float* t = cin;
        cin = cout;
        cout = t;
    }
    float* feats = cin;  // [M,512]
    if (e->tap_conv && (rc = run_copy(e, s, e->tap_conv, feats, (size_t)M * kConvDim))) return rc;

    // ---- feature projection (HF :498-510): LayerNorm(512) in place, then Linear(512,768)
    if ((rc = run_ln(e, s, feats, W(e, pn + "feature_projection.layer_norm.weight"),
                     W(e, pn + "feature_projection.layer_norm.bias"), feats, M, kConvDim)))
        return rc;
    if ((rc = run_gemm(e, s, feats, kConvDim, W(e, pn + "feature_projection.projection.weight"), kConvDim,
                       W(e, pn + "feature_projection.projection.bias"), nullptr, 0, x1, kHidden, (int)M, kHidden, kConvDim,
                       kEpiNone)))
        return rc;
    if (e->tap_proj && (rc = run_copy(e, s, e->tap_proj, x1, (size_t)M * kHidden))) return rc;

    // ---- positional conv + sinusoid (HF :555-564)
    {
        Bracket br(e, s, K_POSCONV, 2.0 * M * (double)kHidden * kPosCg * kPosK, 8.0 * M * kHidden);
        HIP_TRY(launch_pos_conv(x1, e->pos_w, W(e, pn + "pos_conv_embed.conv.bias"), bf.sin_tab, frames_or_null, x0, B, T, s, bf.rows_clip));
    }
    if (e->tap_prenet && (rc = run_copy(e, s, e->tap_prenet, x0, (size_t)M * kHidden))) return rc;
    }  // !skip_prenet

    // ---- encoder (HF :1276-1304)
    if ((rc = run_ln(e, s, x0, W(e, we + "layer_norm.weight"), W(e, we + "layer_norm.bias"), x0, M, kHidden))) return rc;
    const float* pe_k = W(e, we + "embed_positions.pe_k.weight");
    const int nl = e->cfg.layers;
    for (int l = 0; l < nl; ++l) {
        if (hidden_states && hidden_states[l] && (rc = run_copy(e, s, hidden_states[l], x0, (size_t)M * kHidden))) return rc;
        const std::string b = we + "layers." + std::to_string(l) + ".";
        const LayerW& lw = e->layers[l];
        // fused q|k|v projection, q pre-scaled (HF :891,911-914)
        if ((rc = run_gemm(e, s, x0, kHidden, lw.wqkv, kHidden, lw.bqkv, nullptr, 0, qkv, kQkv, (int)M, kQkv, kHidden, kEpiNone)))
            return rc;
        // Qp[b,h] = q_scaled[b,:,h,:] pe_k^T  -> [B,12,T,320]
        if ((rc = run_gemm(e, s, qkv, kQkv, pe_k, kHeadDim, nullptr, nullptr, 0, qp, kRelN, T, kRelN, kHeadDim, kEpiNone, B,
                           kHeads, (long)T * kQkv, kHeadDim, (long)kHeads * T * kRelN, (long)T * kRelN)))
            return rc;
        {
            const double tt = (double)T * T;
            Bracket br(e, s, K_ATTN, 4.0 * B * kHeads * tt * kHeadDim, 4.0 * (M * (double)(kQkv + kHidden) + M * (double)kHeads * kRelN));
            HIP_TRY(launch_attention(qkv, qp, frames_or_null, ctx, B, T, s));
        }
        // out_proj + residual (HF :984,1056), LayerNorm (HF :1058)
        if ((rc = run_gemm(e, s, ctx, kHidden, W(e, b + "attention.out_proj.weight"), kHidden, W(e, b + "attention.out_proj.bias"),
                           x0, kHidden, tmp, kHidden, (int)M, kHidden, kHidden, kEpiResidual)))
            return rc;
        if ((rc = run_ln(e, s, tmp, W(e, b + "layer_norm.weight"), W(e, b + "layer_norm.bias"), x1, M, kHidden))) return rc;
        // FFN (HF :1003-1010) + residual + final LayerNorm (HF :1059-1060)
        if ((rc = run_gemm(e, s, x1, kHidden, W(e, b + "feed_forward.intermediate_dense.weight"), kHidden,
                           W(e, b + "feed_forward.intermediate_dense.bias"), nullptr, 0, ffn, e->cfg.ffn, (int)M, e->cfg.ffn,
                           kHidden, kEpiGelu)))
            return rc;
        if ((rc = run_gemm(e, s, ffn, e->cfg.ffn, W(e, b + "feed_forward.output_dense.weight"), e->cfg.ffn,
                           W(e, b + "feed_forward.output_dense.bias"), x1, kHidden, tmp, kHidden, (int)M, kHidden, e->cfg.ffn,
                           kEpiResidual)))
            return rc;
        float* dst = (l == nl - 1) ? out : x0;
        if ((rc = run_ln(e, s, tmp, W(e, b + "final_layer_norm.weight"), W(e, b + "final_layer_norm.bias"), dst, M, kHidden)))
            return rc;
    }
    if (nl == 0 && (rc = run_copy(e, s, out, x0, (size_t)M * kHidden))) return rc;
    if (hidden_states && hidden_states[nl] && (rc = run_copy(e, s, hidden_states[nl], out, (size_t)M * kHidden))) return rc;
    return LOCO_OK;
}

// ---- precision 1: every contraction (conv layers 1-6, feature projection, positional conv, QKV / out / FFN projections, the
// Qp table, QK^T and PV) on the fp16 x3 split MFMA; every producer writes the fp16 hi/lo planes its consumer needs, so no
// separate conversion pass exists.  conv0 (10 taps, VALU), residuals, GroupNorm / LayerNorm statistics and softmax stay fp32.
int forward_f16x3(loco_encoder* e, Call& c, const Plan& p, const float* wav, float* out, float* const* hidden_states, const Bufs& bf,
                  hipStream_t s, bool skip_prenet = false) {
    const int B = p.B;
    const long L = p.L;
    const int T = (int)p.T;
    const long M = p.M;
    int rc;
    float *x0 = bf.x0, *x1 = bf.x1, *tmp = bf.tmp, *qp = bf.qp;
    const int32_t* frames_or_null = bf.frames_or_null;
    const std::string pn = "prenet.", we = "wrapped_encoder.";
    // q / k / v as fp16 hi|lo planes [M,768] carved from the qkv region (attention clamps the key rows of its last tile to T - 1
    // and multiplies them by P = 0: nothing to pad or to zero)
    _Float16* qshi = reinterpret_cast<_Float16*>(bf.qkv);
    _Float16* qslo = qshi + (size_t)M * kHidden;
    // planes in the order q_hi, q_lo, k_hi, k_lo, v_hi, v_lo: the k and v pairs sit 2 M 768 halves behind the pair before them
    _Float16* const kshi = qslo + (size_t)M * kHidden;
    _Float16* const kslo = kshi + (size_t)M * kHidden;
    _Float16* const vshi = kslo + (size_t)M * kHidden;
    _Float16* const vslo = vshi + (size_t)M * kHidden;
    GemmSplitArgs scat{};
    scat.qkv_stride = (long)2 * M * kHidden;
    scat.T = T;
    // a fp32 buffer of n elements holds the two fp16 planes of n elements back to back
    auto planes = [](float* base, size_t n, _Float16*& hi, _Float16*& lo) {
        hi = reinterpret_cast<_Float16*>(base);
        lo = hi + n;
    };
    // range tracking: one status word (x 8 shards) per tensor that is stored as fp16 planes and whose range does not follow from
    // the weights alone (loco_kernels.h), in launch order; the two halves of a dual-stream forward walk the same sequence and
    // share the words (a maximum does not care who contributes)
    int nslot = 0;
    auto slot = [&](const char* name, int layer = -1) -> float* {
        if (!c.range_dev || !c.st || nslot >= kFiniteStage) return nullptr;
        c.st->names[nslot] = name;
        c.st->layer[nslot] = layer;
        return c.range_dev + (size_t)kRangeShards * nslot++;
    };
    static const char* const kConvNames[7] = {"feature_encoder.conv_layers.0 (GroupNorm + GELU)", "feature_encoder.conv_layers.1",
                                              "feature_encoder.conv_layers.2", "feature_encoder.conv_layers.3",
                                              "feature_encoder.conv_layers.4", "feature_encoder.conv_layers.5",
                                              "feature_encoder.conv_layers.6"};

    if (!skip_prenet) {
    // ---- feature encoder: conv0 writes planes, conv1-5 planes -> planes, conv6 planes -> fp32 (LayerNorm input)
    _Float16 *ihi, *ilo, *ohi, *olo;
    planes(bf.bufA, (size_t)B * p.Tc[0] * kConvDim, ihi, ilo);
    {
        const double outb = 4.0 * B * (double)p.Tc[0] * kConvDim;
        Bracket br(e, s, K_CONV0, 2.0 * 10 * B * (double)p.Tc[0] * kConvDim, outb + 8.0 * B * (double)L);
        HIP_TRY(launch_conv0_gn_gelu(wav, B, L, e->conv_w[0], W(e, pn + "feature_encoder.conv_layers.0.layer_norm.weight"),
                                     W(e, pn + "feature_encoder.conv_layers.0.layer_norm.bias"), nullptr, bf.c0scratch,
                                     e->cfg.ln_eps, s, ihi, ilo, slot(kConvNames[0]), bf.t0_clip));
    }
    float* cin = bf.bufA;
    float* cout = bf.bufB;
    for (int i = 1; i < 7; ++i) {
        const long Tin = p.Tc[i - 1], Tout = p.Tc[i];
        planes(cin, (size_t)B * Tin * kConvDim, ihi, ilo);
        planes(cout, (size_t)B * Tout * kConvDim, ohi, olo);
        const bool last = i == 6;
        rc = run_gemm_split(e, c, s, ihi, ilo, (long)kConvS[i] * kConvDim, e->conv_s[i], (long)kConvK[i] * kConvDim, nullptr, nullptr, 0,
                            last ? cout : nullptr, last ? nullptr : ohi, last ? nullptr : olo, kConvDim, (int)Tout, kConvDim,
                            kConvK[i] * kConvDim, kEpiGelu, B, Tin * kConvDim, Tout * kConvDim, 1, 0, 0, nullptr, nullptr, nullptr,
                            last ? nullptr : slot(kConvNames[i]), K_GEMM_SPLIT, kConvK[i]);
        if (rc) return rc;
        float* t = cin;
        cin = cout;
        cout = t;
    }
    float* feats = cin;  // [M,512] fp32
    if (e->tap_conv && (rc = run_copy(e, s, e->tap_conv, feats, (size_t)M * kConvDim))) return rc;

    // ---- feature projection: LayerNorm(512) -> planes (in the idle conv buffer) -> Linear(512,768) -> x1 fp32
    planes(cout, (size_t)M * kConvDim, ohi, olo);
    if ((rc = run_ln(e, s, feats, W(e, pn + "feature_projection.layer_norm.weight"), W(e, pn + "feature_projection.layer_norm.bias"),
                     nullptr, M, kConvDim, ohi, olo)))
        return rc;
    if ((rc = run_gemm_split(e, c, s, ohi, olo, kConvDim, e->proj_s, kConvDim, W(e, pn + "feature_projection.projection.bias"), nullptr, 0,
                             x1, nullptr, nullptr, kHidden, (int)M, kHidden, kConvDim, kEpiNone)))
        return rc;
    if (e->tap_proj && (rc = run_copy(e, s, e->tap_proj, x1, (size_t)M * kHidden))) return rc;

    // ---- positional conv + sinusoid as ONE split GEMM per (clip, group): x1 is re-laid group-major with a 64-frame zero
    // halo, so that output frame t reads the contiguous run of 128 taps x 48 channels (lda = 48, K = 6144); the epilogue
    // fuses bias, GELU, the residual x1 and the sinusoidal-position gather (HF :389-397,555-564)
    {
        _Float16* ghi = reinterpret_cast<_Float16*>(bf.ffn);  // scratch: the FFN buffer is idle until the first layer
        _Float16* glo = ghi + (size_t)B * kPosGroups * (T + kPosK) * kPosCg;
        {
            Bracket br(e, s, K_COPY, 0.0, 8.0 * M * kHidden);
            HIP_TRY(launch_group_major_split(x1, ghi, glo, B, T, s, slot("feature_projection (input of pos_conv_embed)"), bf.rows_clip));
        }
        GemmSplitArgs a{};
        a.Ahi = ghi; a.Alo = glo; a.Whi = e->posg_s.hi; a.Wlo = e->posg_s.lo;
        a.bias = W(e, pn + "pos_conv_embed.conv.bias");
        a.R = x1; a.C = x0;
        a.M = T; a.N = kPosCg; a.K = kPosK * kPosCg;
        a.lda = kPosCg; a.ldw = (long)kPosK * kPosCg; a.ldc = kHidden; a.ldr = kHidden;
        a.nb1 = B; a.nb2 = kPosGroups;
        a.sA1 = (long)kPosGroups * (T + kPosK) * kPosCg; a.sA2 = (long)(T + kPosK) * kPosCg;
        a.sC1 = (long)T * kHidden; a.sC2 = kPosCg;
        a.sW2 = (long)kPosCg * kPosK * kPosCg; a.sBias2 = kPosCg;
        a.epilogue = kEpiPosConv;
        a.out_scale = e->posg_s.inv_scale;
        a.terms = c.precision == 2 ? 2 : 3;
        a.sin_table = bf.sin_tab; a.frames = frames_or_null; a.T = T;
        a.splitk_ws = c.splitk;  // one or two short clips: split-K over the taps (launch_gemm_split)
        Bracket br(e, s, K_POSCONV_SPLIT, 2.0 * M * (double)kHidden * kPosCg * kPosK, 8.0 * M * kHidden);
        HIP_TRY(launch_gemm_split(a, s));
    }
    if (e->tap_prenet && (rc = run_copy(e, s, e->tap_prenet, x0, (size_t)M * kHidden))) return rc;
    }  // !skip_prenet

    // ---- encoder
    _Float16 *x0hi = bf.xs0, *x0lo = bf.xs0 + (size_t)M * kHidden;
    _Float16 *x1hi = bf.xs1, *x1lo = bf.xs1 + (size_t)M * kHidden;
    _Float16 *chi, *clo, *fhi, *flo;
    planes(bf.ctx, (size_t)M * kHidden, chi, clo);
    planes(bf.ffn, (size_t)M * e->cfg.ffn, fhi, flo);
    // Between layers the residual stream exists only as its fp16 hi/lo planes (22 bits): they are what the next GEMM reads
    // as its A operand anyway, and the residual adds reconstruct hi + lo exactly.  An fp32 copy is written only where
    // somebody reads one: hidden-state outputs, the zero-layer configuration, and the final output.
    float* x0f = (hidden_states || nl_is_zero(e)) ? x0 : nullptr;
    if ((rc = run_ln(e, s, x0, W(e, we + "layer_norm.weight"), W(e, we + "layer_norm.bias"), x0f, M, kHidden, x0hi, x0lo))) return rc;
    const int nl = e->cfg.layers;
    for (int l = 0; l < nl; ++l) {
        if (hidden_states && hidden_states[l] && (rc = run_copy(e, s, hidden_states[l], x0, (size_t)M * kHidden))) return rc;
        const std::string b = we + "layers." + std::to_string(l) + ".";
        const LayerW& lw = e->layers[l];
        // fused q|k|v projection -> q, k and v as fp16 hi/lo planes [M,768] (the layout attention_f16x3 reads; it transposes V itself)
        if ((rc = run_gemm_split(e, c, s, x0hi, x0lo, kHidden, lw.sqkv, kHidden, lw.bqkv, nullptr, 0, nullptr, qshi, qslo, kHidden, (int)M,
                                 kQkv, kHidden, kEpiQkvScatter, 1, 0, 0, 1, 0, 0, &scat, nullptr, nullptr,
                                 slot("attention q|k|v projections", l))))
            return rc;
        if (e->debug_nonfinite) {
            dbg_check(e, s, "x0 planes hi (layer input)", l, x0hi, (size_t)M * kHidden, true, kHidden);
            dbg_check(e, s, "x0 planes lo (layer input)", l, x0lo, (size_t)M * kHidden, true, kHidden);
            dbg_check(e, s, "q planes hi", l, qshi, (size_t)M * kHidden, true, kHidden);
            dbg_check(e, s, "q planes lo", l, qslo, (size_t)M * kHidden, true, kHidden);
            dbg_check(e, s, "k planes hi", l, kshi, (size_t)M * kHidden, true, kHidden);
            dbg_check(e, s, "k planes lo", l, kslo, (size_t)M * kHidden, true, kHidden);
            dbg_check(e, s, "v planes hi", l, vshi, (size_t)M * kHidden, true, kHidden);
            dbg_check(e, s, "v planes lo", l, vslo, (size_t)M * kHidden, true, kHidden);
        }
        // Qp[b,h] = q_scaled[b,:,h,:] pe_k^T -> fp32 [B,12,T,320] is computed INSIDE the attention kernel (attention_f16x3.hip,
        // TABLE form): `qp` is scratch of that launch; no table GEMM runs any more.
        {
            const double tt = (double)T * T;
            // algorithmic FLOPs: QK^T + PV (4 T^2 64 per head) + the compact relative-position table (2 T 320 64 per head);
            // algorithmic bytes: q|k|v in, context out -- the table is now an internal scratch of the launch, not compulsory traffic
            Bracket br(e, s, K_ATTN_SPLIT, 4.0 * B * kHeads * tt * kHeadDim + 2.0 * M * (double)kHeads * kRelN * kHeadDim,
                       4.0 * (M * (double)(kQkv + kHidden)));
            HIP_TRY(launch_attention_f16x3(qshi, qslo, kshi, kslo, vshi, vslo, qp, frames_or_null, chi, clo, nullptr, B, T,
                                           s, e->pe_s.hi, e->pe_s.lo, e->pe_s.inv_scale));
        }
        if (e->debug_nonfinite) {
            dbg_check(e, s, "attention context planes hi", l, chi, (size_t)M * kHidden, true, kHidden);
            dbg_check(e, s, "attention context planes lo", l, clo, (size_t)M * kHidden, true, kHidden);
        }
        if ((rc = run_gemm_split(e, c, s, chi, clo, kHidden, lw.so, kHidden, W(e, b + "attention.out_proj.bias"), nullptr, kHidden, tmp, nullptr,
                                 nullptr, kHidden, (int)M, kHidden, kHidden, kEpiResidual, 1, 0, 0, 1, 0, 0, nullptr, x0hi, x0lo)))
            return rc;
        dbg_check(e, s, "out_proj + residual (fp32)", l, tmp, (size_t)M * kHidden, false, kHidden);
        if ((rc = run_ln(e, s, tmp, W(e, b + "layer_norm.weight"), W(e, b + "layer_norm.bias"), nullptr, M, kHidden, x1hi, x1lo))) return rc;
        if (e->debug_nonfinite) {
            dbg_check(e, s, "layer_norm planes hi", l, x1hi, (size_t)M * kHidden, true, kHidden);
            dbg_check(e, s, "layer_norm planes lo", l, x1lo, (size_t)M * kHidden, true, kHidden);
        }
        if ((rc = run_gemm_split(e, c, s, x1hi, x1lo, kHidden, lw.s1, kHidden, W(e, b + "feed_forward.intermediate_dense.bias"), nullptr, 0,
                                 nullptr, fhi, flo, e->cfg.ffn, (int)M, e->cfg.ffn, kHidden, kEpiGelu, 1, 0, 0, 1, 0, 0, nullptr, nullptr,
                                 nullptr, slot("feed_forward intermediate (GELU)", l))))
            return rc;
        if (e->debug_nonfinite) {
            dbg_check(e, s, "feed_forward intermediate planes hi", l, fhi, (size_t)M * e->cfg.ffn, true, e->cfg.ffn);
            dbg_check(e, s, "feed_forward intermediate planes lo", l, flo, (size_t)M * e->cfg.ffn, true, e->cfg.ffn);
        }
        if ((rc = run_gemm_split(e, c, s, fhi, flo, e->cfg.ffn, lw.s2, e->cfg.ffn, W(e, b + "feed_forward.output_dense.bias"), nullptr, kHidden,
                                 tmp, nullptr, nullptr, kHidden, (int)M, kHidden, e->cfg.ffn, kEpiResidual, 1, 0, 0, 1, 0, 0, nullptr, x1hi,
                                 x1lo)))
            return rc;
        dbg_check(e, s, "feed_forward output + residual (fp32)", l, tmp, (size_t)M * kHidden, false, kHidden);
        const bool last = l == nl - 1;
        if ((rc = run_ln(e, s, tmp, W(e, b + "final_layer_norm.weight"), W(e, b + "final_layer_norm.bias"),
                         last ? out : (hidden_states ? x0 : nullptr), M, kHidden, last ? nullptr : x0hi, last ? nullptr : x0lo,
                         last && c.range_dev ? c.range_dev + (size_t)kRangeShards * kFiniteStage : nullptr)))
            return rc;
    }
    if (nl == 0 && (rc = run_copy(e, s, out, x0, (size_t)M * kHidden))) return rc;
    if (hidden_states && hidden_states[nl] && (rc = run_copy(e, s, hidden_states[nl], out, (size_t)M * kHidden))) return rc;
    if (c.st) c.st->used = nslot;
    return LOCO_OK;
}

// The status words travel to pinned host memory behind the forward, on its stream: readable once the stream has got there.
int range_begin(loco_encoder* e, Call& c, hipStream_t s) {
    StatusBlock* st = c.st;
    st->magic = kStatusMagic;
    st->precision = c.precision;
    st->used = 0;
    st->msg[0] = 0;
    if (c.precision >= 1 && !e->range_static.empty()) snprintf(st->msg, sizeof st->msg, "%s", e->range_static.c_str());
    if (c.precision >= 1) HIP_TRY(hipMemsetAsync(c.range_dev, 0, sizeof(float) * kRangeMaxStages * kRangeShards, s));
    return LOCO_OK;
}
int range_end(const Call& c, hipStream_t s) {
    if (c.precision >= 1 && c.st->used > 0)  // all of them (3 KiB): the tracked stages and the reserved finite-check word
        HIP_TRY(hipMemcpyAsync(c.st->words, c.range_dev, sizeof(float) * kRangeShards * kRangeMaxStages, hipMemcpyDeviceToHost, s));
    return LOCO_OK;
}

}  // namespace

extern "C" {

int loco_abi_version(void) { return LOCO_ABI_VERSION; }

const char* loco_last_error(void) { return g_err.c_str(); }

void loco_default_config(loco_config* c) {
    if (!c) return;
    c->struct_size = (int32_t)sizeof(loco_config);
    c->hidden = kHidden;
    c->heads = kHeads;
    c->ffn = kFfn;
    c->layers = 12;
    c->conv_dim = kConvDim;
    c->pos_conv_kernel = kPosK;
    c->pos_conv_groups = kPosGroups;
    c->rel_max = kRelMax;
    c->ln_eps = 1e-5f;
}

loco_encoder* loco_create(const loco_config* cfg) {
    loco_config c;
    loco_default_config(&c);
    if (cfg) {
        if (cfg->struct_size != (int32_t)sizeof(loco_config)) {
            fail(LOCO_E_INVALID, "loco_config.struct_size %d != %zu", cfg->struct_size, sizeof(loco_config));
            return nullptr;
        }
        c = *cfg;
    }
    if (c.hidden != kHidden || c.heads != kHeads || c.ffn != kFfn || c.conv_dim != kConvDim ||
        c.pos_conv_kernel != kPosK || c.pos_conv_groups != kPosGroups || c.rel_max != kRelMax || c.layers < 0 ||
        c.layers > 64) {
        fail(LOCO_E_INVALID, "unsupported configuration: kernels are specialised for SpeechT5-base (768/12/3072/512/128/16/160)");
        return nullptr;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        fail(LOCO_E_HIP, "hipGetDevice failed: no usable HIP device");
        return nullptr;
    }
    loco_encoder* e = new loco_encoder();
    e->cfg = c;
    e->device = dev;
    if (hipHostMalloc(reinterpret_cast<void**>(&e->own), sizeof(StatusBlock), hipHostMallocDefault) != hipSuccess ||
        hipMalloc(&e->absmax_dev, sizeof(float)) != hipSuccess) {
        fail(LOCO_E_HIP, "loco_create: could not allocate the range-status block");
        if (e->own) (void)hipHostFree(e->own);
        delete e;
        return nullptr;
    }
    memset(e->own, 0, sizeof(StatusBlock));
    e->own->magic = kStatusMagic;
    if (const char* dbg = getenv("LOCO_DEBUG_NONFINITE")) {
        e->debug_nonfinite = dbg[0] == '1' && hipMalloc(&e->debug_counter, 2 * sizeof(unsigned long long)) == hipSuccess;
    }
    e->layers.resize(c.layers);
    build_expected(e);
    for (int i = 0; i < K_COUNT; ++i) {
        memset(&e->stats[i], 0, sizeof(loco_kernel_stat));
        snprintf(e->stats[i].name, sizeof e->stats[i].name, "%s", kKernelNames[i]);
    }
    return e;
}

void loco_destroy(loco_encoder* e) {
    if (!e) return;
    for (auto& kv : e->raw) (void)hipFree(kv.second.d);
    for (int i = 1; i < 7; ++i) (void)hipFree(e->conv_w[i]);
    (void)hipFree(e->pos_w);
    auto free_split = [](SplitW& w) {
        (void)hipFree(w.hi);
        (void)hipFree(w.lo);
    };
    for (int i = 1; i < 7; ++i) free_split(e->conv_s[i]);
    free_split(e->proj_s);
    free_split(e->pe_s);
    free_split(e->posg_s);
    for (auto& l : e->layers) {
        (void)hipFree(l.wqkv);
        (void)hipFree(l.bqkv);
        free_split(l.sqkv);
        free_split(l.so);
        free_split(l.s1);
        free_split(l.s2);
    }
    (void)hipFree(e->absmax_dev);
    (void)hipFree(e->debug_counter);
    if (e->own) (void)hipHostFree(e->own);
    (void)hipFree(e->sin_tab.load());
    for (float* t : e->retired) (void)hipFree(t);
    (void)hipFree(e->text_embed);
    (void)hipFree(e->text_alpha);
    (void)hipFree(e->text_pe);
    if (e->side) {
        (void)hipStreamDestroy(e->side);
        (void)hipEventDestroy(e->ev_fork);
        (void)hipEventDestroy(e->ev_join);
    }
    for (auto& r : e->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    delete e;
}

int loco_set_weight(loco_encoder* e, const char* key, const float* data, const int64_t* shape, int ndim) {
    if (!e || !key || !data || !shape || ndim < 1 || ndim > 4) return fail(LOCO_E_INVALID, "loco_set_weight: null/invalid argument");
    const std::string k = canonical_key(key);
    std::vector<int64_t> shp(shape, shape + ndim);
    int64_t n = 1;
    for (auto s : shp) n *= s;
    if (k == "prenet.pos_sinusoidal_embed.weights") {
        if (ndim != 2 || shp[1] != kHidden || shp[0] < 3) return fail(LOCO_E_INVALID, "%s: expected [rows,768]", key);
        float* d = nullptr;
        HIP_TRY(hipMalloc(&d, (size_t)n * sizeof(float)));
        HIP_TRY(hipMemcpy(d, data, (size_t)n * sizeof(float), hipMemcpyDefault));
        HIP_TRY(hipDeviceSynchronize());
        {   // a weight load: the caller guarantees no forward is in flight (as for every loco_set_weight)
            std::lock_guard<std::mutex> lock(e->sin_mu);
            (void)hipFree(e->sin_tab.load(std::memory_order_relaxed));
            e->sin_rows.store(0, std::memory_order_release);
            e->sin_tab.store(d, std::memory_order_release);
            e->sin_rows.store((int)shp[0], std::memory_order_release);
        }
        e->sin_user = true;
        return LOCO_OK;
    }
    if (k.rfind("text_prenet.", 0) == 0) {
        // SpeechT5TextEncoderPrenet.state_dict(): embed_tokens.weight [V,768], encode_positions.alpha [] (pass as [1]);
        // encode_positions.pe [1,rows,768] or [rows,768] is a non-persistent buffer in current HF, a key in 4.30's pickles
        float** slot = nullptr;
        if (k == "text_prenet.embed_tokens.weight") {
            if (ndim != 2 || shp[1] != kHidden || shp[0] < 1) return fail(LOCO_E_INVALID, "%s: expected [vocab,768]", key);
            slot = &e->text_embed;
            e->text_vocab = (int)shp[0];
        } else if (k == "text_prenet.encode_positions.alpha") {
            if (n != 1) return fail(LOCO_E_INVALID, "%s: expected one element", key);
            slot = &e->text_alpha;
        } else if (k == "text_prenet.encode_positions.pe") {
            if (!((ndim == 2 && shp[1] == kHidden) || (ndim == 3 && shp[0] == 1 && shp[2] == kHidden)))
                return fail(LOCO_E_INVALID, "%s: expected [rows,768] or [1,rows,768]", key);
            slot = &e->text_pe;
            e->text_pe_rows = (int)(n / kHidden);
        } else {
            return fail(LOCO_E_INVALID, "unexpected key in state_dict: %s", key);
        }
        float* d = nullptr;
        HIP_TRY(hipMalloc(&d, (size_t)n * sizeof(float)));
        HIP_TRY(hipMemcpy(d, data, (size_t)n * sizeof(float), hipMemcpyDefault));
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(*slot);
        *slot = d;
        return LOCO_OK;
    }
    auto it = e->expected.find(k);
    if (it == e->expected.end()) return fail(LOCO_E_INVALID, "unexpected key in state_dict: %s", key);
    if (it->second != shp) {
        std::string got, want;
        for (auto s : shp) got += std::to_string(s) + ",";
        for (auto s : it->second) want += std::to_string(s) + ",";
        return fail(LOCO_E_INVALID, "size mismatch for %s: got [%s] expected [%s]", key, got.c_str(), want.c_str());
    }
    Tensor& t = e->raw[k];
    if (!t.d) HIP_TRY(hipMalloc(&t.d, (size_t)n * sizeof(float)));
    t.shape = shp;
    HIP_TRY(hipMemcpy(t.d, data, (size_t)n * sizeof(float), hipMemcpyDefault));
    HIP_TRY(hipStreamSynchronize(nullptr));  // device-to-device copies may return early; the caller may free `data` now
    e->finalized = false;
    return LOCO_OK;
}

int loco_missing_weights(const loco_encoder* e, char* buf, size_t buflen) {
    if (!e) return fail(LOCO_E_INVALID, "null encoder");
    int missing = 0;
    std::string names;
    bool any_speech = false;
    for (auto& kv : e->raw) any_speech = any_speech || kv.first.rfind("prenet.", 0) == 0;
    const bool text_only = e->text_embed && !any_speech;  // a text-encoder handle: the speech prenet is not required
    for (auto& kv : e->expected) {
        if (optional_key(kv.first) || e->raw.count(kv.first)) continue;
        if (text_only && kv.first.rfind("prenet.", 0) == 0) continue;
        ++missing;
        if (!names.empty()) names += ",";
        names += kv.first;
    }
    if (e->text_embed && !e->text_alpha) {
        ++missing;
        names += (names.empty() ? "" : ",") + std::string("text_prenet.encode_positions.alpha");
    }
    if (buf && buflen) snprintf(buf, buflen, "%s", names.c_str());
    return missing;
}

int loco_finalize_weights(loco_encoder* e, void* stream) {
    if (!e) return fail(LOCO_E_INVALID, "null encoder");
    hipStream_t s = (hipStream_t)stream;
    char names[256];
    const int miss = loco_missing_weights(e, names, sizeof names);
    if (miss) return fail(LOCO_E_STATE, "%d weights missing: %s", miss, names);
    const std::string p = "prenet.", w = "wrapped_encoder.";
    const bool speech = e->raw.count(p + "feature_encoder.conv_layers.0.conv.weight") != 0;
    e->speech_ready = false;
    if (e->text_embed && !e->text_pe) {  // C callers without HF's table: 1024 rows from the library's own generator
        HIP_TRY(hipMalloc(&e->text_pe, (size_t)1024 * kHidden * sizeof(float)));
        HIP_TRY(launch_text_pe_table(e->text_pe, 1024, s));
        e->text_pe_rows = 1024;
    }
    // conv layers 1..6: [512,512,k] -> [512, k*512]
    for (int i = 1; speech && i < 7; ++i) {
        const size_t n = (size_t)kConvDim * kConvDim * kConvK[i];
        if (!e->conv_w[i]) HIP_TRY(hipMalloc(&e->conv_w[i], n * sizeof(float)));
        HIP_TRY(launch_relayout_conv_weight(W(e, p + "feature_encoder.conv_layers." + std::to_string(i) + ".conv.weight"),
                                            e->conv_w[i], kConvDim, kConvDim, kConvK[i], s));
    }
    if (speech) {
        e->conv_w[0] = e->raw.at(p + "feature_encoder.conv_layers.0.conv.weight").d;
        // positional conv: fold weight-norm, lay out [group][tap][o][i]
        if (!e->pos_w) HIP_TRY(hipMalloc(&e->pos_w, (size_t)kHidden * kPosCg * kPosK * sizeof(float)));
        HIP_TRY(launch_fold_pos_conv(W(e, p + "pos_conv_embed.conv.parametrizations.weight.original0"),
                                     W(e, p + "pos_conv_embed.conv.parametrizations.weight.original1"), e->pos_w, s));
    }
    // fused QKV with the 1/8 query scaling folded in: (x Wq^T + bq)/8 == x (Wq/8)^T + bq/8 exactly (power of two)
    const size_t hh = (size_t)kHidden * kHidden;
    for (int l = 0; l < e->cfg.layers; ++l) {
        LayerW& lw = e->layers[l];
        const std::string b = w + "layers." + std::to_string(l) + ".attention.";
        if (!lw.wqkv) HIP_TRY(hipMalloc(&lw.wqkv, 3 * hh * sizeof(float)));
        if (!lw.bqkv) HIP_TRY(hipMalloc(&lw.bqkv, 3 * kHidden * sizeof(float)));
        HIP_TRY(launch_scale_copy(W(e, b + "q_proj.weight"), lw.wqkv, hh, 0.125f, s));
        HIP_TRY(launch_scale_copy(W(e, b + "k_proj.weight"), lw.wqkv + hh, hh, 1.0f, s));
        HIP_TRY(launch_scale_copy(W(e, b + "v_proj.weight"), lw.wqkv + 2 * hh, hh, 1.0f, s));
        HIP_TRY(launch_scale_copy(W(e, b + "q_proj.bias"), lw.bqkv, kHidden, 0.125f, s));
        HIP_TRY(launch_scale_copy(W(e, b + "k_proj.bias"), lw.bqkv + kHidden, kHidden, 1.0f, s));
        HIP_TRY(launch_scale_copy(W(e, b + "v_proj.bias"), lw.bqkv + 2 * kHidden, kHidden, 1.0f, s));
    }
    // fp16 hi/lo planes of every GEMM weight for precision mode f16x3 (378 MB; built unconditionally so that the
    // mode can be switched per forward)
    int rc = LOCO_OK;
    for (int i = 1; speech && i < 7 && !rc; ++i) {  // conv planes in the channel-block-major k order the GEMM walks (GemmSplitArgs::ktaps)
        const size_t n = (size_t)kConvDim * kConvDim * kConvK[i];
        float* tmpw = nullptr;
        HIP_TRY(hipMalloc(&tmpw, n * sizeof(float)));
        const hipError_t he = launch_permute_conv_k(e->conv_w[i], tmpw, kConvDim, kConvK[i], kConvDim, s);
        if (he == hipSuccess) rc = make_split(e, e->conv_s[i], tmpw, n, s);
        HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(tmpw);
        if (he != hipSuccess) return fail(LOCO_E_HIP, "permute_conv_k: %s", hipGetErrorString(he));
    }
    if (speech && !rc) rc = make_split(e, e->proj_s, W(e, p + "feature_projection.projection.weight"), (size_t)kHidden * kConvDim, s);
    if (!rc) rc = make_split(e, e->pe_s, W(e, w + "embed_positions.pe_k.weight"), (size_t)kRelN * kHeadDim, s);
    if (speech && !rc) {  // positional conv weight re-laid [g][o][tap*48+i] for the conv-as-GEMM form, then split
        float* tmpw = nullptr;
        const size_t n = (size_t)kHidden * kPosCg * kPosK;
        HIP_TRY(hipMalloc(&tmpw, n * sizeof(float)));
        hipError_t he = launch_pos_w_for_gemm(e->pos_w, tmpw, s);
        if (he == hipSuccess) rc = make_split(e, e->posg_s, tmpw, n, s);
        HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(tmpw);
        if (he != hipSuccess) return fail(LOCO_E_HIP, "pos_w_for_gemm: %s", hipGetErrorString(he));
    }
    for (int l = 0; l < e->cfg.layers && !rc; ++l) {
        LayerW& lw = e->layers[l];
        const std::string b = w + "layers." + std::to_string(l) + ".";
        rc = make_split(e, lw.sqkv, lw.wqkv, 3 * hh, s);
        if (!rc) rc = make_split(e, lw.so, W(e, b + "attention.out_proj.weight"), hh, s);
        if (!rc) rc = make_split(e, lw.s1, W(e, b + "feed_forward.intermediate_dense.weight"), (size_t)e->cfg.ffn * kHidden, s);
        if (!rc) rc = make_split(e, lw.s2, W(e, b + "feed_forward.output_dense.weight"), (size_t)e->cfg.ffn * kHidden, s);
    }
    if (rc) return rc;
    if (speech) {
        const float* unused_tab = nullptr;
        rc = ensure_sin_rows(e, 4002, s, &unused_tab);
        if (rc) return rc;
    }
    // weight-determined activation ranges of precision mode f16x3 (static_range_check above)
    // (conv0's GroupNorm + GELU output is not among them: its bound sqrt(frames per clip) * max|gamma| + max|beta| says nothing
    // useful for 10-minute clips -- 1385 * max|gamma| -- so conv0_apply_kernel folds max|x| of what it writes like the GEMMs do)
    e->range_static.clear();
    if (speech) rc = static_range_check(e, p + "feature_projection.layer_norm.", kConvDim, s);
    if (!rc) rc = static_range_check(e, w + "layer_norm.", kHidden, s);
    for (int l = 0; l < e->cfg.layers && !rc; ++l) {
        const std::string b = w + "layers." + std::to_string(l) + ".";
        rc = static_range_check(e, b + "layer_norm.", kHidden, s);
        if (!rc && l + 1 < e->cfg.layers) rc = static_range_check(e, b + "final_layer_norm.", kHidden, s);  // the last one is written in fp32
    }
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    e->speech_ready = speech;
    e->finalized = true;
    return LOCO_OK;
}

int64_t loco_output_frames(int64_t n) {
    for (int i = 0; i < 7; ++i) {
        const int64_t d = n - kConvK[i];
        n = (d >= 0 ? d / kConvS[i] : -((-d + kConvS[i] - 1) / kConvS[i])) + 1;
    }
    return n;
}

namespace {
// Two half-batches on two streams: clips are independent, so the halves give the same bits as one pass, and the tail of
// every kernel of one half (the last, partly filled round of workgroups: up to 12 % of the N = 768 GEMMs) is filled by
// the other half's kernels.  Measured gain (tools/two_stream_probe.py): +2 % at 32 x 30 s, +5..8 % at 16-32 clips of 2.5-15 s;
// halves of fewer than ~1000 frames (8 x 5 s) lose 4 %, so those stay on one stream.
bool split_batch(const loco_encoder* e, int B, long L, Plan& p0, Plan& p1) {
    if (e->streams < 2 || B < 2) return false;
    const int B0 = (B + 1) / 2;
    if (!make_plan(e, B0, L, p0) || !make_plan(e, B - B0, L, p1)) return false;
    return p1.M >= 1024;
}
}  // namespace

size_t loco_workspace_bytes(const loco_encoder* e, int32_t B, int64_t L) {
    Plan p, p0, p1;
    if (!e || !make_plan(e, B, L, p)) return 0;
    if (split_batch(e, B, L, p0, p1)) return kStatusDevBytes + std::max(p.total, p0.total + p1.total);
    return kStatusDevBytes + p.total;
}

int loco_set_streams(loco_encoder* e, int n) {
    if (!e || (n != 1 && n != 2)) return fail(LOCO_E_INVALID, "loco_set_streams: n must be 1 or 2");
    e->streams = n;
    return LOCO_OK;
}

const char* loco_precision_name(int mode) {
    static const char* const kNames[] = {"f32", "f16x3", "f16x2"};
    return (mode >= 0 && mode < 3) ? kNames[mode] : nullptr;
}

int loco_set_precision(loco_encoder* e, int mode) {
    if (!e || !loco_precision_name(mode)) return fail(LOCO_E_INVALID, "loco_set_precision: mode must be 0 (f32), 1 (f16x3) or 2 (f16x2)");
    e->precision = mode;
    return LOCO_OK;
}

int loco_get_precision(const loco_encoder* e) { return e ? e->precision : LOCO_E_INVALID; }

int loco_set_taps(loco_encoder* e, float* conv_stack, float* feature_projection, float* prenet) {
    if (!e) return fail(LOCO_E_INVALID, "null encoder");
    e->tap_conv = conv_stack;
    e->tap_proj = feature_projection;
    e->tap_prenet = prenet;
    return LOCO_OK;
}

namespace {
// clip_tab: null, or (loco_forward_packed) the host table {conv0 frames [B], encoder frames [B]} of THIS (half-)batch's clips
int forward_one(loco_encoder* e, Call& c, const Plan& p, const float* wav, const int32_t* mask, int B, long L, float* out, int32_t* out_frames,
                float* const* hidden_states, char* ws, hipStream_t s, const int32_t* clip_t0 = nullptr, const int32_t* clip_rows = nullptr,
                const int32_t* clip_valid = nullptr, const float* sin_tab = nullptr) {
    int32_t* frames = out_frames ? out_frames : reinterpret_cast<int32_t*>(ws + p.off_frames);
    float* bufA = reinterpret_cast<float*>(ws + p.off_a);
    float* bufB = reinterpret_cast<float*>(ws + p.off_b);
    float* x0 = reinterpret_cast<float*>(ws + p.off_x0);
    float* x1 = reinterpret_cast<float*>(ws + p.off_x1);
    float* tmp = reinterpret_cast<float*>(ws + p.off_tmp);
    float* ctx = reinterpret_cast<float*>(ws + p.off_ctx);
    float* qkv = reinterpret_cast<float*>(ws + p.off_qkv);
    float* qp = reinterpret_cast<float*>(ws + p.off_qp);
    float* ffn = reinterpret_cast<float*>(ws + p.off_ffn);

    // ---- valid frame counts (HF :569-598); a packed forward that was given valid_len has them from the host (below)
    if (!clip_valid) {
        Bracket br(e, s, K_FRAMES, 0.0, mask ? 4.0 * B * (double)L : 0.0);
        HIP_TRY(launch_frame_counts(mask, B, L, frames, s));
    }
    const int32_t* frames_or_null = mask ? frames : nullptr;

    struct Bufs bufs{frames, frames_or_null, bufA, bufB, x0, x1, tmp, ctx, qkv, qp, ffn, reinterpret_cast<_Float16*>(ws + p.off_xs0),
                     reinterpret_cast<_Float16*>(ws + p.off_xs1), ws + p.off_c0scratch};
    bufs.sin_tab = sin_tab;
    if (clip_t0) {
        // packed forward: the per-clip tables follow the stream into the workspace; without a mask every sample of a clip's own
        // reference batch counts, so its valid frames are that batch's frames (and a key mask is needed whatever the caller passed:
        // clips of shorter batches end before the pack does)
        int32_t* tab = reinterpret_cast<int32_t*>(ws + p.off_clip);
        HIP_TRY(hipMemcpyAsync(tab, clip_t0, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(tab + B, clip_rows, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s));
        if (clip_valid) HIP_TRY(hipMemcpyAsync(frames, clip_valid, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s));
        else if (!mask) HIP_TRY(hipMemcpyAsync(frames, tab + B, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        bufs.frames_or_null = frames;
        bufs.t0_clip = tab;
        bufs.rows_clip = tab + B;
    }
    c.splitk = p.splitk ? reinterpret_cast<float*>(ws + p.off_splitk) : nullptr;
    return c.precision >= 1 ? forward_f16x3(e, c, p, wav, out, hidden_states, bufs, s) : forward_f32(e, p, wav, out, hidden_states, bufs, s);
}

// One forward in arithmetic mode `precision`, its range words in the first bytes of the workspace, its status in `st`.  Nothing
// the enqueue mutates is shared between calls except the lazily created side stream (side_mu), the profiling records (profiling
// is a single-caller diagnostic mode) and the sinusoid table when a clip longer than any before makes it grow.
int forward_impl(loco_encoder* e, int precision, StatusBlock* st, const float* wav, const int32_t* mask, int32_t B, int64_t L, float* out,
                 int32_t* out_frames, float* const* hidden_states, void* workspace, size_t workspace_bytes, void* stream,
                 const int64_t* pad_len = nullptr, const int64_t* valid_len = nullptr) {
    if (!e || !wav || !out || !workspace || !st) return fail(LOCO_E_INVALID, "loco_forward: null argument");
    if (!e->finalized) return fail(LOCO_E_STATE, "loco_forward: call loco_finalize_weights first");
    if (!e->speech_ready) return fail(LOCO_E_STATE, "loco_forward: this encoder was loaded without the speech prenet weights");
    Plan p, p0, p1;
    if (!make_plan(e, B, L, p))
        return fail(LOCO_E_INVALID, "loco_forward: batch %d x %lld samples gives no output frame (need >= 400 samples)", B, (long long)L);
    if (B > 65535) return fail(LOCO_E_INVALID, "loco_forward: batch %d > 65535", B);
    if (p.M > 0x7fffffffL / 8) return fail(LOCO_E_INVALID, "loco_forward: B*T = %ld frames is too large", p.M);
    // per-kernel timing, hidden-state and tap outputs keep the single in-order pass
    const bool dual = !e->profiling && !hidden_states && !e->tap_conv && !e->tap_proj && !e->tap_prenet && split_batch(e, B, L, p0, p1);
    const size_t need = kStatusDevBytes + (dual ? p0.total + p1.total : p.total);
    if (workspace_bytes < need)
        return fail(LOCO_E_WORKSPACE, "loco_forward: workspace %zu < required %zu bytes", workspace_bytes, need);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) || (reinterpret_cast<uintptr_t>(wav) & 3))
        return fail(LOCO_E_INVALID, "loco_forward: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float* sin_tab = nullptr;  // this forward's snapshot of the sinusoid table
    int rc = ensure_sin_rows(e, (int)p.T + 2, s, &sin_tab);
    if (rc) return rc;
    const int32_t *tab_t0 = nullptr, *tab_rows = nullptr, *tab_valid = nullptr;
    if (pad_len) {  // loco_forward_packed: validate, then derive the two per-clip frame counts into the caller's status block
        if (B > kMaxPackClips) return fail(LOCO_E_INVALID, "loco_forward_packed: %d clips > %d", B, kMaxPackClips);
        for (int b = 0; b < B; ++b) {
            if (pad_len[b] > L || loco_output_frames(pad_len[b]) < 1)
                return fail(LOCO_E_INVALID, "loco_forward_packed: pad_len[%d] = %lld must lie in [400, L = %lld]", b, (long long)pad_len[b],
                            (long long)L);
            st->clip_tab[b] = (int32_t)conv_out_len(pad_len[b], kConvK[0], kConvS[0]);
            st->clip_tab[B + b] = (int32_t)loco_output_frames(pad_len[b]);
            if (valid_len) {
                if (valid_len[b] < 0 || valid_len[b] > pad_len[b])
                    return fail(LOCO_E_INVALID, "loco_forward_packed: valid_len[%d] = %lld must lie in [0, pad_len = %lld]", b,
                                (long long)valid_len[b], (long long)pad_len[b]);
                st->clip_tab[2 * B + b] = (int32_t)loco_output_frames(valid_len[b]);  // as frames_from_counts_kernel: HF's floor division
            }
        }
        tab_t0 = st->clip_tab;
        tab_rows = st->clip_tab + B;
        if (valid_len) tab_valid = st->clip_tab + 2 * B;
    }
    Call c;
    c.precision = precision;
    c.st = st;
    c.range_dev = reinterpret_cast<float*>(workspace);
    char* ws = reinterpret_cast<char*>(workspace) + kStatusDevBytes;
    if ((rc = range_begin(e, c, s))) return rc;
    if (!dual) {
        rc = forward_one(e, c, p, wav, mask, B, L, out, out_frames, hidden_states, ws, s, tab_t0, tab_rows, tab_valid, sin_tab);
        if (rc) return rc;
        return range_end(c, s);
    }

    // The side stream is one per handle: forwards in flight together take turns on it (the lock covers the enqueue only).
    std::lock_guard<std::mutex> lock(e->side_mu);
    if (!e->side) {
        HIP_TRY(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    }
    const int B0 = (B + 1) / 2, B1 = B - B0;
    HIP_TRY(hipEventRecord(e->ev_fork, s));  // the side stream starts after everything already queued on the caller's stream
    HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork, 0));
    // From here on the side stream may hold work that reads the caller's buffers and the workspace: whatever fails below, the
    // caller's stream is joined to it before this function returns, so that "stream idle" still means "workspace free".
    c.dual = true;
    rc = forward_one(e, c, p0, wav, mask, B0, L, out, out_frames, nullptr, ws, s, tab_t0, tab_rows, tab_valid, sin_tab);
    int rc1 = LOCO_OK;
    std::string first_error;
    if (rc) first_error = g_err;
    else rc1 = forward_one(e, c, p1, wav + (size_t)B0 * L, mask ? mask + (size_t)B0 * L : nullptr, B1, L, out + (size_t)B0 * p.T * kHidden,
                           out_frames ? out_frames + B0 : nullptr, nullptr, ws + p0.total, e->side, tab_t0 ? tab_t0 + B0 : nullptr,
                           tab_rows ? tab_rows + B0 : nullptr, tab_valid ? tab_valid + B0 : nullptr, sin_tab);
    c.dual = false;
    if (!rc && rc1) first_error = g_err;
    const hipError_t j1 = hipEventRecord(e->ev_join, e->side);  // ... and the caller's stream continues after both halves
    const hipError_t j2 = j1 == hipSuccess ? hipStreamWaitEvent(s, e->ev_join, 0) : j1;
    if (j2 != hipSuccess) {
        // the join itself failed: block until the side stream has drained rather than hand back a workspace in use
        (void)hipStreamSynchronize(e->side);
        if (!rc && !rc1) return fail(LOCO_E_HIP, "loco_forward: joining the second stream failed: %s", hipGetErrorString(j2));
    }
    if (rc || rc1) {
        g_err = first_error;
        return rc ? rc : rc1;
    }
    return range_end(c, s);
}

const StatusBlock* as_status(const void* status) {
    const StatusBlock* st = reinterpret_cast<const StatusBlock*>(status);
    return (st && st->magic == kStatusMagic) ? st : nullptr;
}

float stage_amax(const StatusBlock* st, int i) {
    float amax = 0.f;
    for (int k = 0; k < kRangeShards; ++k) amax = fmaxf(amax, st->words[(size_t)i * kRangeShards + k]);
    return amax;
}

int status_check(const StatusBlock* st, char* buf, size_t buflen) {
    if (buf && buflen) buf[0] = 0;
    if (st->precision == 0) return LOCO_OK;  // the exact-fp32 mode stores no fp16 planes
    if (st->msg[0]) {
        if (buf && buflen) snprintf(buf, buflen, "%s", st->msg);
        return fail(LOCO_E_RANGE, "%s; use the exact-fp32 kernels for this model (loco_set_precision(enc, 0) / loco_forward_checked)", st->msg);
    }
    for (int i = 0; i < st->used; ++i) {
        const float amax = stage_amax(st, i);
        const bool over = !(amax < kRangeHi), under = amax < kRangeLo;
        if (!over && !under) continue;
        char where[160];
        if (st->layer[i] >= 0) snprintf(where, sizeof where, "wrapped_encoder.layers.%d %s", st->layer[i], st->names[i]);
        else snprintf(where, sizeof where, "%s", st->names[i]);
        char msg[400];
        snprintf(msg, sizeof msg,
                 "activation range: max|x| = %.6g of '%s' is %s the range precision mode f16x3 represents to fp32 class "
                 "(%g <= max|x| < %g); use the exact-fp32 kernels for this input (loco_set_precision(enc, 0) / loco_forward_checked)",
                 (double)amax, where, over ? "above" : "below", (double)kRangeLo, (double)kRangeHi);
        if (buf && buflen) snprintf(buf, buflen, "%s", msg);
        return fail(LOCO_E_RANGE, "%s", msg);
    }
    // the finite check of the last LayerNorm: the range words above are maxima taken with fmaxf, which a NaN never enters -- an
    // inf / NaN born inside a stage (not by leaving the range of a tracked plane tensor) is caught where everything ends up
    if (st->used > 0 && stage_amax(st, kFiniteStage) > 0.f) {
        const char* msg = "activation range: last_hidden_state holds non-finite values (inf / NaN) although every tracked plane tensor "
                          "stayed inside the range precision mode f16x3 represents; use the exact-fp32 kernels for this input "
                          "(loco_set_precision(enc, 0) / loco_forward_checked)";
        if (buf && buflen) snprintf(buf, buflen, "%s", msg);
        return fail(LOCO_E_RANGE, "%s", msg);
    }
    return LOCO_OK;
}

int status_range(const StatusBlock* st, int32_t stage, float* amax, int32_t* layer, char* name, size_t namelen) {
    if (stage < 0 || stage >= st->used) return st->used;  // not an error: lets a caller enumerate 0 .. n-1
    if (amax) *amax = stage_amax(st, stage);
    if (layer) *layer = st->layer[stage];
    if (name && namelen) snprintf(name, namelen, "%s", st->names[stage]);
    return st->used;
}
}  // namespace

int loco_forward(loco_encoder* e, const float* wav, const int32_t* mask, int32_t B, int64_t L, float* out,
                 int32_t* out_frames, float* const* hidden_states, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e) return fail(LOCO_E_INVALID, "loco_forward: null argument");
    return forward_impl(e, e->precision, e->own, wav, mask, B, L, out, out_frames, hidden_states, workspace, workspace_bytes, stream);
}

// ---- forwards in flight: one status block per forward (include/loco_asr.h) ---------------------------------------------------
size_t loco_status_bytes(void) { return sizeof(StatusBlock); }

int loco_forward_async(loco_encoder* e, int precision, const float* wav, const int32_t* mask, int32_t B, int64_t L, float* out,
                       int32_t* out_frames, float* const* hidden_states, void* workspace, size_t workspace_bytes, void* stream,
                       void* status) {
    if (!e || !status) return fail(LOCO_E_INVALID, "loco_forward_async: null argument");
    if (precision != -1 && !loco_precision_name(precision))
        return fail(LOCO_E_INVALID, "loco_forward_async: precision must be -1 (the handle's mode) or one of " LOCO_PRECISION_MODES);
    if (reinterpret_cast<uintptr_t>(status) & 7) return fail(LOCO_E_INVALID, "loco_forward_async: the status block must be 8-byte aligned");
    StatusBlock* st = reinterpret_cast<StatusBlock*>(status);
    st->magic = 0;  // not a valid status until the enqueue has succeeded
    const int rc = forward_impl(e, precision < 0 ? e->precision : precision, st, wav, mask, B, L, out, out_frames, hidden_states, workspace,
                                workspace_bytes, stream);
    if (rc) st->magic = 0;
    return rc;
}

int loco_forward_packed(loco_encoder* e, int precision, const float* wav, const int32_t* mask, const int64_t* valid_len, int32_t B, int64_t L,
                        const int64_t* pad_len, float* out, int32_t* out_frames, float* const* hidden_states, void* workspace,
                        size_t workspace_bytes, void* stream, void* status) {
    if (!e || !status || !pad_len) return fail(LOCO_E_INVALID, "loco_forward_packed: null argument");
    if (mask && valid_len) return fail(LOCO_E_INVALID, "loco_forward_packed: give attention_mask or valid_len, not both");
    if (precision != -1 && !loco_precision_name(precision))
        return fail(LOCO_E_INVALID, "loco_forward_packed: precision must be -1 (the handle's mode) or one of " LOCO_PRECISION_MODES);
    if (reinterpret_cast<uintptr_t>(status) & 7) return fail(LOCO_E_INVALID, "loco_forward_packed: the status block must be 8-byte aligned");
    StatusBlock* st = reinterpret_cast<StatusBlock*>(status);
    st->magic = 0;
    const int rc = forward_impl(e, precision < 0 ? e->precision : precision, st, wav, mask, B, L, out, out_frames, hidden_states, workspace,
                                workspace_bytes, stream, pad_len, valid_len);
    if (rc) st->magic = 0;
    return rc;
}

int loco_max_pack_clips(void) { return kMaxPackClips; }

int loco_status_check(const void* status, char* buf, size_t buflen) {
    const StatusBlock* st = as_status(status);
    if (!st) return fail(LOCO_E_INVALID, "loco_status_check: not a status block filled by loco_forward_async");
    return status_check(st, buf, buflen);
}

int loco_status_range(const void* status, int32_t stage, float* amax, int32_t* layer, char* name, size_t namelen) {
    const StatusBlock* st = as_status(status);
    if (!st) return fail(LOCO_E_INVALID, "loco_status_range: not a status block filled by loco_forward_async");
    return status_range(st, stage, amax, layer, name, namelen);
}

// ---- range status of the last forward enqueued through loco_forward / loco_forward_text (include/loco_asr.h) ------------------
int loco_forward_status(loco_encoder* e, char* buf, size_t buflen) {
    if (!e) return fail(LOCO_E_INVALID, "null encoder");
    return status_check(e->own, buf, buflen);
}

int loco_forward_range(const loco_encoder* e, int32_t stage, float* amax, int32_t* layer, char* name, size_t namelen) {
    if (!e) return fail(LOCO_E_INVALID, "null encoder");
    return status_range(e->own, stage, amax, layer, name, namelen);
}

int loco_set_range_policy(loco_encoder* e, int policy) {
    if (!e || (policy != 0 && policy != 1)) return fail(LOCO_E_INVALID, "loco_set_range_policy: policy must be 0 (report) or 1 (re-run in fp32)");
    e->range_policy = policy;
    return LOCO_OK;
}

int loco_forward_checked(loco_encoder* e, const float* wav, const int32_t* mask, int32_t B, int64_t L, float* out, int32_t* out_frames,
                         float* const* hidden_states, void* workspace, size_t workspace_bytes, void* stream, int32_t* used_fp32) {
    if (used_fp32) *used_fp32 = 0;
    int rc = loco_forward(e, wav, mask, B, L, out, out_frames, hidden_states, workspace, workspace_bytes, stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    rc = loco_forward_status(e, nullptr, 0);
    if (rc != LOCO_E_RANGE || e->range_policy == 0) return rc;
    // out of the fp16 planes' range: the same batch again on the exact-fp32 MFMA kernels of this library
    rc = forward_impl(e, 0, e->own, wav, mask, B, L, out, out_frames, hidden_states, workspace, workspace_bytes, stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (used_fp32) *used_fp32 = 1;
    return LOCO_OK;
}

// ---- text front end ---------------------------------------------------------------------------------------
size_t loco_text_workspace_bytes(const loco_encoder* e, int32_t B, int32_t T) {
    Plan p;
    if (!e || !make_plan_tokens(e, B, T, p)) return 0;
    return kStatusDevBytes + p.total;
}

int loco_text_max_positions(const loco_encoder* e) { return e ? e->text_pe_rows : 0; }

namespace {
int forward_text_impl(loco_encoder* e, int precision, StatusBlock* st, const int32_t* input_ids, const int32_t* attention_mask, int32_t B, int32_t T,
                      float* out, int32_t* out_frames, float* const* hidden_states, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e || !input_ids || !out || !workspace || !st) return fail(LOCO_E_INVALID, "loco_forward_text: null argument");
    if (!e->finalized) return fail(LOCO_E_STATE, "loco_forward_text: call loco_finalize_weights first");
    if (!e->text_embed || !e->text_alpha || !e->text_pe)
        return fail(LOCO_E_STATE, "loco_forward_text: this encoder was loaded without the text prenet weights");
    Plan p;
    if (!make_plan_tokens(e, B, T, p)) return fail(LOCO_E_INVALID, "loco_forward_text: batch %d x %d tokens is empty", B, T);
    if (B > 65535) return fail(LOCO_E_INVALID, "loco_forward_text: batch %d > 65535", B);
    if (T > e->text_pe_rows)
        return fail(LOCO_E_INVALID, "loco_forward_text: %d tokens exceed the positional table (%d rows: max_text_positions)", T,
                    e->text_pe_rows);
    if (workspace_bytes < kStatusDevBytes + p.total)
        return fail(LOCO_E_WORKSPACE, "loco_forward_text: workspace %zu < required %zu bytes", workspace_bytes, kStatusDevBytes + p.total);
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(LOCO_E_INVALID, "loco_forward_text: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    Call c;
    c.precision = precision;
    c.st = st;
    c.range_dev = reinterpret_cast<float*>(workspace);
    char* ws = reinterpret_cast<char*>(workspace) + kStatusDevBytes;
    int32_t* frames = out_frames ? out_frames : reinterpret_cast<int32_t*>(ws + p.off_frames);
    float* x0 = reinterpret_cast<float*>(ws + p.off_x0);
    int rcb = range_begin(e, c, s);
    if (rcb) return rcb;
    {
        Bracket br(e, s, K_FRAMES, 0.0, attention_mask ? 4.0 * B * (double)T : 0.0);
        HIP_TRY(launch_token_counts(attention_mask, B, T, frames, s));
    }
    {   // embed_tokens + alpha * pe (HF SpeechT5TextEncoderPrenet.forward)
        Bracket br(e, s, K_COPY, 0.0, 12.0 * p.M * kHidden);
        HIP_TRY(launch_text_prenet(input_ids, e->text_embed, e->text_vocab, e->text_alpha, e->text_pe, B, T, x0, s));
    }
    struct Bufs bufs{frames, attention_mask ? frames : nullptr, reinterpret_cast<float*>(ws + p.off_a), reinterpret_cast<float*>(ws + p.off_b),
                     x0, reinterpret_cast<float*>(ws + p.off_x1), reinterpret_cast<float*>(ws + p.off_tmp),
                     reinterpret_cast<float*>(ws + p.off_ctx), reinterpret_cast<float*>(ws + p.off_qkv),
                     reinterpret_cast<float*>(ws + p.off_qp), reinterpret_cast<float*>(ws + p.off_ffn),
                     reinterpret_cast<_Float16*>(ws + p.off_xs0), reinterpret_cast<_Float16*>(ws + p.off_xs1), ws + p.off_c0scratch};
    c.splitk = p.splitk ? reinterpret_cast<float*>(ws + p.off_splitk) : nullptr;
    const int rc = c.precision >= 1 ? forward_f16x3(e, c, p, nullptr, out, hidden_states, bufs, s, true)
                                    : forward_f32(e, p, nullptr, out, hidden_states, bufs, s, true);
    return rc ? rc : range_end(c, s);
}
}  // namespace

int loco_forward_text(loco_encoder* e, const int32_t* input_ids, const int32_t* attention_mask, int32_t B, int32_t T, float* out,
                      int32_t* out_frames, float* const* hidden_states, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e) return fail(LOCO_E_INVALID, "loco_forward_text: null argument");
    return forward_text_impl(e, e->precision, e->own, input_ids, attention_mask, B, T, out, out_frames, hidden_states, workspace, workspace_bytes, stream);
}

int loco_forward_text_async(loco_encoder* e, int precision, const int32_t* input_ids, const int32_t* attention_mask, int32_t B, int32_t T,
                            float* out, int32_t* out_frames, float* const* hidden_states, void* workspace, size_t workspace_bytes,
                            void* stream, void* status) {
    if (!e || !status) return fail(LOCO_E_INVALID, "loco_forward_text_async: null argument");
    if (precision != -1 && !loco_precision_name(precision))
        return fail(LOCO_E_INVALID, "loco_forward_text_async: precision must be -1 (the handle's mode) or one of " LOCO_PRECISION_MODES);
    if (reinterpret_cast<uintptr_t>(status) & 7) return fail(LOCO_E_INVALID, "loco_forward_text_async: the status block must be 8-byte aligned");
    StatusBlock* st = reinterpret_cast<StatusBlock*>(status);
    st->magic = 0;
    const int rc = forward_text_impl(e, precision < 0 ? e->precision : precision, st, input_ids, attention_mask, B, T, out, out_frames, hidden_states,
                                     workspace, workspace_bytes, stream);
    if (rc) st->magic = 0;
    return rc;
}

// ---- profiling -----------------------------------------------------------------------------------------
int loco_set_profiling(loco_encoder* e, int on) {
    if (!e) return fail(LOCO_E_INVALID, "null encoder");
    e->profiling = on != 0;
    return LOCO_OK;
}

int loco_set_profiling_filter(loco_encoder* e, const char* bucket) {
    if (!e) return fail(LOCO_E_INVALID, "null encoder");
    if (!bucket || !*bucket) {
        e->profile_only = -1;
        return LOCO_OK;
    }
    for (int i = 0; i < K_COUNT; ++i)
        if (!strcmp(bucket, kKernelNames[i])) {
            e->profile_only = i;
            return LOCO_OK;
        }
    return fail(LOCO_E_INVALID, "loco_set_profiling_filter: no kernel bucket named '%s'", bucket);
}

static int drain_records(loco_encoder* e) {
    for (size_t i = 0; i < e->recs_used; ++i) {
        ProfRec& r = e->recs[i];
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        loco_kernel_stat& st = e->stats[r.kid];
        st.launches += 1;
        st.ms += ms;
        st.flops += r.flops;
        st.bytes += r.bytes;
    }
    e->recs_used = 0;
    return LOCO_OK;
}

int loco_profile_reset(loco_encoder* e) {
    if (!e) return fail(LOCO_E_INVALID, "null encoder");
    int rc = drain_records(e);
    if (rc) return rc;
    for (int i = 0; i < K_COUNT; ++i) {
        e->stats[i].launches = 0;
        e->stats[i].ms = e->stats[i].flops = e->stats[i].bytes = 0.0;
    }
    return LOCO_OK;
}

int loco_profile_read(loco_encoder* e, loco_kernel_stat* stats, int max_stats) {
    if (!e || !stats) return fail(LOCO_E_INVALID, "null argument");
    int rc = drain_records(e);
    if (rc) return rc;
    int n = 0;
    for (int i = 0; i < K_COUNT && n < max_stats; ++i)
        if (e->stats[i].launches) stats[n++] = e->stats[i];
    return n;
}

// ---- single operators ------------------------------------------------------------------------------------
int loco_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t dim, float eps,
                      void* stream) {
    if (!x || !gamma || !beta || !y) return fail(LOCO_E_INVALID, "loco_op_layernorm: null argument");
    if (dim != 512 && dim != 768) return fail(LOCO_E_INVALID, "loco_op_layernorm: dim %d not in {512,768}", dim);
    HIP_TRY(launch_layernorm(x, gamma, beta, y, rows, dim, eps, (hipStream_t)stream));
    return LOCO_OK;
}

int loco_op_gemm(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, const float* R, int64_t ldr,
                 float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t nb1, int32_t nb2,
                 int64_t sA1, int64_t sA2, int64_t sC1, int64_t sC2, void* stream) {
    if (!A || !Wt || !C) return fail(LOCO_E_INVALID, "loco_op_gemm: null argument");
    if (K % 32 || (lda | ldw) & 3) return fail(LOCO_E_INVALID, "loco_op_gemm: K %% 32 and lda/ldw %% 4 must be 0");
    if (nb1 < 1 || nb2 < 1) return fail(LOCO_E_INVALID, "loco_op_gemm: batch counts must be >= 1");
    GemmArgs a{A, Wt, bias, R, C, M, N, K, lda, ldw, ldc, ldr, nb1, nb2, sA1, sA2, sC1, sC2, epilogue};
    HIP_TRY(launch_gemm(a, (hipStream_t)stream));
    return LOCO_OK;
}

size_t loco_conv0_scratch_bytes(int32_t B) { return conv0_scratch_bytes(B); }

int loco_op_conv0_gn_gelu(const float* wav, int32_t B, int64_t L, const float* w, const float* gn_w, const float* gn_b,
                          float* out, void* scratch, void* stream) {
    if (!wav || !w || !gn_w || !gn_b || !out || !scratch) return fail(LOCO_E_INVALID, "loco_op_conv0_gn_gelu: null argument");
    if (L < 10) return fail(LOCO_E_INVALID, "loco_op_conv0_gn_gelu: L < 10");
    HIP_TRY(launch_conv0_gn_gelu(wav, B, L, w, gn_w, gn_b, out, scratch, 1e-5f, (hipStream_t)stream));
    return LOCO_OK;
}

int loco_op_frame_counts(const int32_t* mask, int32_t B, int64_t L, int32_t* frames, void* stream) {
    if (!frames || B <= 0) return fail(LOCO_E_INVALID, "loco_op_frame_counts: invalid argument");
    HIP_TRY(launch_frame_counts(mask, B, L, frames, (hipStream_t)stream));
    return LOCO_OK;
}

int loco_op_pos_conv(const float* h, const float* w_folded, const float* bias, const float* sin_table, const int32_t* frames,
                     float* out, int32_t B, int32_t T, void* stream) {
    if (!h || !w_folded || !bias || !sin_table || !out) return fail(LOCO_E_INVALID, "loco_op_pos_conv: null argument");
    HIP_TRY(launch_pos_conv(h, w_folded, bias, sin_table, frames, out, B, T, (hipStream_t)stream));
    return LOCO_OK;
}


size_t loco_normalize_scratch_bytes(int32_t B) { return B > 0 ? normalize_scratch_bytes(B) : 0; }

int loco_op_normalize_waveform(const float* wav, const int32_t* attention_mask, int32_t B, int64_t L, float padding_value, float* out,
                               void* scratch, size_t scratch_bytes, void* stream) {
    if (!wav || !out || !scratch || B <= 0 || L <= 0) return fail(LOCO_E_INVALID, "loco_op_normalize_waveform: null/invalid argument");
    if (scratch_bytes < normalize_scratch_bytes(B) || (reinterpret_cast<uintptr_t>(scratch) & 7))
        return fail(LOCO_E_WORKSPACE, "loco_op_normalize_waveform: scratch %zu < %zu bytes (or not 8-byte aligned)", scratch_bytes,
                    normalize_scratch_bytes(B));
    HIP_TRY(launch_normalize_waveform(wav, attention_mask, B, L, padding_value, out, scratch, (hipStream_t)stream));
    return LOCO_OK;
}

// ---- sample-rate conversion ("next" row f-4) -----------------------------------------------------------------------------
int loco_resample_design(int32_t sr_in, int32_t sr_out, int32_t* up, int32_t* down, int32_t* taps_per_phase, float* taps_host) {
    if (!up || !down || !taps_per_phase) return fail(LOCO_E_INVALID, "loco_resample_design: null argument");
    int L = 0, M = 0, K = 0;
    const int rc = resample_design(sr_in, sr_out, &L, &M, &K, taps_host);
    if (rc) return fail(LOCO_E_INVALID, "loco_resample_design: unsupported rates %d -> %d Hz", sr_in, sr_out);
    *up = L; *down = M; *taps_per_phase = K;
    return LOCO_OK;
}

int64_t loco_resample_length(int64_t n_in, int32_t up, int32_t down) {
    if (n_in <= 0 || up <= 0 || down <= 0) return 0;
    return (n_in * up + down - 1) / down;  // ceil(n * sr_out / sr_in): librosa.resample's n_samples (+ fix_length)
}

int loco_op_resample(const float* x, int32_t B, int64_t n_in, int64_t x_stride, const float* taps_dev, int32_t up, int32_t down,
                     int32_t taps_per_phase, float* y, int64_t n_out, int64_t y_stride, void* stream) {
    if (!x || !taps_dev || !y) return fail(LOCO_E_INVALID, "loco_op_resample: null argument");
    if (B <= 0 || n_in <= 0 || n_out <= 0 || n_out > loco_resample_length(n_in, up, down) || x_stride < n_in || y_stride < n_out ||
        (reinterpret_cast<uintptr_t>(taps_dev) & 15))
        return fail(LOCO_E_INVALID, "loco_op_resample: invalid shape (n_out must be <= ceil(n_in * up / down); taps 16-byte aligned)");
    HIP_TRY(launch_resample(x, B, n_in, x_stride, taps_dev, up, down, taps_per_phase, y, n_out, y_stride, (hipStream_t)stream));
    return LOCO_OK;
}

int loco_op_split_f16(const float* x, void* hi, void* lo, int64_t n, void* stream) {
    if (!x || !hi || !lo || n <= 0 || (n & 3)) return fail(LOCO_E_INVALID, "loco_op_split_f16: invalid argument");
    HIP_TRY(launch_split_f16(x, hi, lo, n, (hipStream_t)stream));
    return LOCO_OK;
}

int loco_op_gemm_f16x3(const void* Ahi, const void* Alo, int64_t lda, const void* Whi, const void* Wlo, int64_t ldw,
                       const float* bias, const float* R, int64_t ldr, float* C, void* Chi, void* Clo, int64_t ldc, int32_t M,
                       int32_t N, int32_t K, int32_t epilogue, int32_t nb1, int32_t nb2, int64_t sA1, int64_t sA2, int64_t sC1,
                       int64_t sC2, void* stream) {
    if (!Ahi || !Alo || !Whi || !Wlo || (!C && !Chi)) return fail(LOCO_E_INVALID, "loco_op_gemm_f16x3: null argument");
    if (K % 32 || (lda | ldw) & 7) return fail(LOCO_E_INVALID, "loco_op_gemm_f16x3: K %% 32 and lda/ldw %% 8 must be 0");
    GemmSplitArgs a{(const _Float16*)Ahi, (const _Float16*)Alo, (const _Float16*)Whi, (const _Float16*)Wlo, bias, R, C,
                    (_Float16*)Chi, (_Float16*)Clo, M, N, K, lda, ldw, ldc, ldr, nb1, nb2, sA1, sA2, sC1, sC2, epilogue};
    HIP_TRY(launch_gemm_split(a, (hipStream_t)stream));
    return LOCO_OK;
}

int loco_op_permute_conv_k(const float* w, float* out, int32_t N, int32_t taps, int32_t C, void* stream) {
    if (!w || !out) return fail(LOCO_E_INVALID, "loco_op_permute_conv_k: null argument");
    HIP_TRY(launch_permute_conv_k(w, out, N, taps, C, (hipStream_t)stream));
    return LOCO_OK;
}

int loco_op_conv_gemm_f16x3(const void* Ahi, const void* Alo, int64_t lda, const void* Whi, const void* Wlo, float* C, void* Chi, void* Clo,
                            int32_t Tout, int32_t N, int32_t Cin, int32_t taps, int32_t epilogue, int32_t B, int64_t sA1, void* stream) {
    if (!Ahi || !Alo || !Whi || !Wlo || (!C && !Chi)) return fail(LOCO_E_INVALID, "loco_op_conv_gemm_f16x3: null argument");
    if (taps < 1 || taps > 3 || Cin % 64 || lda & 7) return fail(LOCO_E_INVALID, "loco_op_conv_gemm_f16x3: taps in 1..3, Cin %% 64 and lda %% 8");
    const int K = taps * Cin;
    GemmSplitArgs a{(const _Float16*)Ahi, (const _Float16*)Alo, (const _Float16*)Whi, (const _Float16*)Wlo, nullptr, nullptr, C,
                    (_Float16*)Chi, (_Float16*)Clo, Tout, N, K, lda, K, N, 0, B, 1, sA1, 0, (int64_t)Tout * N, 0, epilogue};
    a.ktaps = taps;
    HIP_TRY(launch_gemm_split(a, (hipStream_t)stream));
    return LOCO_OK;
}

size_t loco_gemm_splitk_bytes(void) { return kSplitKBytes; }

void loco_debug_reload_gemm_knobs(void) {
    reload_gemm_knobs();
    reload_attention_knobs();
}

int loco_op_gemm_f16x3_splitk(const void* Ahi, const void* Alo, int64_t lda, const void* Whi, const void* Wlo, int64_t ldw,
                              const float* bias, const float* R, int64_t ldr, float* C, void* Chi, void* Clo, int64_t ldc, int32_t M,
                              int32_t N, int32_t K, int32_t epilogue, void* splitk_ws, size_t splitk_bytes, void* stream) {
    if (!Ahi || !Alo || !Whi || !Wlo || (!C && !Chi) || !splitk_ws) return fail(LOCO_E_INVALID, "loco_op_gemm_f16x3_splitk: null argument");
    if (K % 32 || (lda | ldw) & 7) return fail(LOCO_E_INVALID, "loco_op_gemm_f16x3_splitk: K %% 32 and lda/ldw %% 8 must be 0");
    if (splitk_bytes < kSplitKBytes) return fail(LOCO_E_WORKSPACE, "loco_op_gemm_f16x3_splitk: workspace %zu < %zu bytes", splitk_bytes, kSplitKBytes);
    GemmSplitArgs a{(const _Float16*)Ahi, (const _Float16*)Alo, (const _Float16*)Whi, (const _Float16*)Wlo, bias, R, C,
                    (_Float16*)Chi, (_Float16*)Clo, M, N, K, lda, ldw, ldc, ldr, 1, 1, 0, 0, 0, 0, epilogue};
    a.splitk_ws = reinterpret_cast<float*>(splitk_ws);
    HIP_TRY(launch_gemm_split(a, (hipStream_t)stream));
    return LOCO_OK;
}

int loco_op_attention_f16x3(const void* qhi, const void* qlo, const void* khi, const void* klo, const void* vhi, const void* vlo,
                            const float* qp, const int32_t* frames, float* ctx, int32_t B, int32_t T, void* stream) {
    if (!qhi || !qlo || !khi || !klo || !vhi || !vlo || !qp || !ctx) return fail(LOCO_E_INVALID, "loco_op_attention_f16x3: null argument");
    HIP_TRY(launch_attention_f16x3((const _Float16*)qhi, (const _Float16*)qlo, (const _Float16*)khi, (const _Float16*)klo,
                                   (const _Float16*)vhi, (const _Float16*)vlo, qp, frames, nullptr, nullptr, ctx, B, T,
                                   (hipStream_t)stream));
    return LOCO_OK;
}

int loco_op_attention_f16x3_pe(const void* qhi, const void* qlo, const void* khi, const void* klo, const void* vhi, const void* vlo,
                               const void* pe_hi, const void* pe_lo, float pe_scale, float* qp_scratch, const int32_t* frames, float* ctx,
                               int32_t B, int32_t T, void* stream) {
    if (!qhi || !qlo || !khi || !klo || !vhi || !vlo || !pe_hi || !pe_lo || !qp_scratch || !ctx)
        return fail(LOCO_E_INVALID, "loco_op_attention_f16x3_pe: null argument");
    HIP_TRY(launch_attention_f16x3((const _Float16*)qhi, (const _Float16*)qlo, (const _Float16*)khi, (const _Float16*)klo,
                                   (const _Float16*)vhi, (const _Float16*)vlo, qp_scratch, frames, nullptr, nullptr, ctx, B, T,
                                   (hipStream_t)stream, (const _Float16*)pe_hi, (const _Float16*)pe_lo, pe_scale));
    return LOCO_OK;
}

int loco_op_attention(const float* qkv, const float* qp, const int32_t* frames, float* ctx, int32_t B, int32_t T, void* stream) {
    if (!qkv || !qp || !ctx) return fail(LOCO_E_INVALID, "loco_op_attention: null argument");
    HIP_TRY(launch_attention(qkv, qp, frames, ctx, B, T, (hipStream_t)stream));
    return LOCO_OK;
}

}  // extern "C"

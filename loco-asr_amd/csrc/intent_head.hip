// "Next" row f-1: the downstream intent head that consumes the encoder's embeddings,
//   IntentClassifier  (/root/reference/speech_text/intent_classifier.py:24-49): pooling over time
//       average / max / learned-query attention  alpha = softmax_t(x_t . q), pooled = sum_t alpha_t x_t   (:32-36)
//       + Linear(768, 101)
//   one optimisation step of train_classifier.py:104-115: CrossEntropyLoss on one-hot FLOAT targets (soft-label form,
//       mean over the batch), backward, Adam(lr 1e-3, weight_decay 1e-4) (:66-68).
// The encoder is frozen (the reference trains on pre-extracted embeddings), so only q, W, b receive gradients.
// Padded frames take part in the pooling exactly as in the reference (pad_sequence zeros, no mask; :47-50).
//
// HBM-bound byte work (x is read 2-3 times, 2.4 MB per 16 x 50 frames ... 590 MB at 16 x 12k): the time axis is
// split over a (B, splits) grid flash-style -- each block produces a partial (max, sum, weighted row sum) that a
// tiny combine kernel merges -- so long recordings fill the chip instead of 16 CUs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/loco_asr.h"

namespace {

constexpr int D = 768;       // embedding size
constexpr int C = 101;       // classes
constexpr int kRows = 128;   // frames per split
constexpr int kNParams = D + C * D + C;

thread_local char g_head_err[256];

struct Part {  // per (b, split)
    float m, l, s;  // running max, sum exp, sum alpha*dalpha (backward)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// z[b,t] = x[b,t,:] . q  (one wavefront per frame, 12 floats per lane)
__global__ __launch_bounds__(256) void head_scores_kernel(const float* __restrict__ x, const float* __restrict__ q,
                                                          float* __restrict__ z, long rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    const float4* qr = reinterpret_cast<const float4*>(q);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float4 a = xr[lane + 64 * i], b = qr[lane + 64 * i];
        acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) z[row] = acc;
}

// partial pooling over frames [t0, t1) of clip b.  method 0: sum, 1: max, 2: softmax-weighted sum with local max.
// BWD (attention backward): the query gradient of clip b is  dq_b = sum_t alpha_t dalpha_t (x_t - p_b),  dalpha_t = dpooled_b . x_t,
// p_b = pooled_b = sum_t alpha_t x_t.  Written as  sum_t alpha_t dalpha_t x_t  -  (sum_t alpha_t dalpha_t) p_b  it is a one-pass
// covariance: two sums of ~|dalpha| |p| that cancel to their difference -- on LayerNorm'd embeddings (a large common component p,
// T = 1499 frames) fp32 keeps two digits of it (measured: 2.6e-2 relative to fp64, torch's fp32 autograd 1e-2).  Because
// sum_t alpha_t (x_t - p_b) = 0, dalpha_t may be replaced by dalpha_t - dpooled_b . p_b = dpooled_b . (x_t - p_b):
//     dq_b = sum_t alpha_t (dpooled_b . (x_t - p_b)) (x_t - p_b)                  = Cov_alpha(x) dpooled_b
// -- every factor centred, nothing cancels, plain fp32 sums suffice.  The block accumulates that form (alpha from the global lse).
template <int METHOD, bool BWD>
__global__ __launch_bounds__(256) void head_partial_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                           const float* __restrict__ lse, const float* __restrict__ dpooled,
                                                           const float* __restrict__ pooled,
                                                           float* __restrict__ part_vec, Part* __restrict__ part, int T,
                                                           int splits) {
    __shared__ float w[kRows];
    __shared__ float red[4];
    const int b = blockIdx.y, s = blockIdx.x;
    const int t0 = s * kRows;
    const int nt = min(kRows, T - t0);
    const float* xb = x + ((long)b * T + t0) * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m_loc = 0.f, l_loc = 0.f, s_loc = 0.f;
    if (METHOD == 2) {
        if (!BWD) {
            float m = -INFINITY;
            for (int t = tid; t < nt; t += 256) m = fmaxf(m, z[(long)b * T + t0 + t]);
            m = wave_max(m);
            if (lane == 0) red[wave] = m;
            __syncthreads();
            m_loc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            __syncthreads();
            float l = 0.f;
            for (int t = tid; t < nt; t += 256) {
                const float p = expf(z[(long)b * T + t0 + t] - m_loc);
                w[t] = p;
                l += p;
            }
            l = wave_sum(l);
            if (lane == 0) red[wave] = l;
            __syncthreads();
            l_loc = (red[0] + red[1]) + (red[2] + red[3]);
        } else {
            // alpha_t * dpooled_b . (x_t - p_b)  (one wavefront per frame; the centring happens element by element, before the dot)
            const float4* dp = reinterpret_cast<const float4*>(dpooled + (long)b * D);
            const float4* pp = reinterpret_cast<const float4*>(pooled + (long)b * D);
            float4 g4[3], p4[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { g4[i] = dp[lane + 64 * i]; p4[i] = pp[lane + 64 * i]; }
            float sacc = 0.f;
            for (int t = wave; t < nt; t += 4) {
                const float4* xr = reinterpret_cast<const float4*>(xb + (long)t * D);
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float4 a = xr[lane + 64 * i], g = g4[i], c = p4[i];
                    acc += ((a.x - c.x) * g.x + (a.y - c.y) * g.y) + ((a.z - c.z) * g.z + (a.w - c.w) * g.w);
                }
                acc = wave_sum(acc);
                const float wa = expf(z[(long)b * T + t0 + t] - lse[b]) * acc;
                if (lane == 0) w[t] = wa;
                sacc += wa;
            }
            if (lane == 0) red[wave] = sacc;
            __syncthreads();
            s_loc = (red[0] + red[1]) + (red[2] + red[3]);
        }
        __syncthreads();
    }
    // weighted row sum / sum / max over the split: thread owns columns tid, tid+256, tid+512
    float a0 = METHOD == 1 ? -INFINITY : 0.f, a1 = a0, a2 = a0;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;  // BWD: the clip's pooled vector, subtracted from every frame
    if (METHOD == 2 && BWD) {
        const float* pb = pooled + (long)b * D;
        c0 = pb[tid]; c1 = pb[tid + 256]; c2 = pb[tid + 512];
    }
    for (int t = 0; t < nt; ++t) {
        const float* xr = xb + (long)t * D;
        const float v0 = xr[tid] - c0, v1 = xr[tid + 256] - c1, v2 = xr[tid + 512] - c2;
        if (METHOD == 0) { a0 += v0; a1 += v1; a2 += v2; }
        if (METHOD == 1) { a0 = fmaxf(a0, v0); a1 = fmaxf(a1, v1); a2 = fmaxf(a2, v2); }
        if (METHOD == 2) { const float ww = w[t]; a0 = fmaf(ww, v0, a0); a1 = fmaf(ww, v1, a1); a2 = fmaf(ww, v2, a2); }
    }
    float* pv = part_vec + ((long)b * splits + s) * D;
    pv[tid] = a0; pv[tid + 256] = a1; pv[tid + 512] = a2;
    if (tid == 0) part[b * splits + s] = Part{m_loc, l_loc, s_loc};
}

// merge the splits of clip b -> pooled[b,:] (+ lse[b] for attention); thread owns columns tid, tid+256, tid+512
template <int METHOD>
__global__ __launch_bounds__(256) void head_combine_kernel(const float* __restrict__ part_vec, const Part* __restrict__ part,
                                                           float* __restrict__ pooled, float* __restrict__ lse, int T,
                                                           int splits) {
    const int b = blockIdx.x, tid = threadIdx.x;
    float M = -INFINITY, L = 0.f;
    if (METHOD == 2) {
        for (int s = 0; s < splits; ++s) M = fmaxf(M, part[b * splits + s].m);
        for (int s = 0; s < splits; ++s) L += part[b * splits + s].l * expf(part[b * splits + s].m - M);
    }
    float a0 = METHOD == 1 ? -INFINITY : 0.f, a1 = a0, a2 = a0;
    for (int s = 0; s < splits; ++s) {
        const float* pv = part_vec + ((long)b * splits + s) * D;
        if (METHOD == 1) {
            a0 = fmaxf(a0, pv[tid]); a1 = fmaxf(a1, pv[tid + 256]); a2 = fmaxf(a2, pv[tid + 512]);
        } else {
            const float f = METHOD == 2 ? expf(part[b * splits + s].m - M) : 1.f;
            a0 = fmaf(f, pv[tid], a0); a1 = fmaf(f, pv[tid + 256], a1); a2 = fmaf(f, pv[tid + 512], a2);
        }
    }
    const float sc = METHOD == 0 ? 1.f / (float)T : (METHOD == 2 ? 1.f / L : 1.f);
    float* pb = pooled + (long)b * D;
    pb[tid] = a0 * sc; pb[tid + 256] = a1 * sc; pb[tid + 512] = a2 * sc;
    if (METHOD == 2 && tid == 0) lse[b] = M + logf(L);
}

// logits[b,c] = pooled[b,:] . W[c,:] + bias[c]   (one wavefront per (b,c))
__global__ __launch_bounds__(256) void head_logits_kernel(const float* __restrict__ pooled, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ logits, int B) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    const float4* p = reinterpret_cast<const float4*>(pooled + (long)b * D);
    const float4* w = reinterpret_cast<const float4*>(W + (long)c * D);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float4 a = p[lane + 64 * i], g = w[lane + 64 * i];
        acc += (a.x * g.x + a.y * g.y) + (a.z * g.z + a.w * g.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) logits[idx] = acc + bias[c];
}

// soft-label cross entropy, mean over the batch (torch CrossEntropyLoss with probability targets):
//   loss = -1/B sum_b sum_c t_bc log_softmax(logits_b)_c ;  dlogits_bc = (softmax_bc * sum_c t_bc - t_bc) / B
__global__ __launch_bounds__(128) void head_ce_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                      float* __restrict__ dlogits, float* __restrict__ loss_b, int B) {
    __shared__ float red[2];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float v = tid < C ? logits[b * C + tid] : -INFINITY;
    const float t = tid < C ? target[b * C + tid] : 0.f;
    float m = wave_max(v);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(red[0], red[1]);
    __syncthreads();
    const float e = tid < C ? expf(v - m) : 0.f;
    float se = wave_sum(e);
    if (lane == 0) red[wave] = se;
    __syncthreads();
    se = red[0] + red[1];
    __syncthreads();
    float ts = wave_sum(t);
    if (lane == 0) red[wave] = ts;
    __syncthreads();
    ts = red[0] + red[1];
    __syncthreads();
    const float logp = v - m - logf(se);
    float lt = wave_sum(tid < C ? -t * logp : 0.f);
    if (lane == 0) red[wave] = lt;
    __syncthreads();
    if (tid < C) dlogits[b * C + tid] = (e / se * ts - t) / (float)B;
    if (tid == 0) loss_b[b] = red[0] + red[1];
}

// grads layout: [q (768) | W (101*768) | b (101)].  dW = dlogits^T pooled, db = sum_b dlogits, dpooled = dlogits W,
// loss = mean_b loss_b.  grid = C + B + 1 blocks of 256 threads.
__global__ __launch_bounds__(256) void head_param_grads_kernel(const float* __restrict__ dlogits, const float* __restrict__ pooled,
                                                               const float* __restrict__ W, const float* __restrict__ loss_b,
                                                               float* __restrict__ grads, float* __restrict__ dpooled,
                                                               float* __restrict__ loss, int B) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk < C) {  // row c of dW and db[c]
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, sb = 0.f;
        for (int b = 0; b < B; ++b) {
            const float g = dlogits[b * C + blk];
            const float* p = pooled + (long)b * D;
            a0 = fmaf(g, p[tid], a0); a1 = fmaf(g, p[tid + 256], a1); a2 = fmaf(g, p[tid + 512], a2);
            sb += g;
        }
        float* gw = grads + D + (long)blk * D;
        gw[tid] = a0; gw[tid + 256] = a1; gw[tid + 512] = a2;
        if (tid == 0) grads[D + C * D + blk] = sb;
    } else if (blk < C + B) {  // dpooled[b,:]
        const int b = blk - C;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float g = dlogits[b * C + c];
            const float* w = W + (long)c * D;
            a0 = fmaf(g, w[tid], a0); a1 = fmaf(g, w[tid + 256], a1); a2 = fmaf(g, w[tid + 512], a2);
        }
        float* dp = dpooled + (long)b * D;
        dp[tid] = a0; dp[tid + 256] = a1; dp[tid + 512] = a2;
    } else if (tid == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += loss_b[b];
        *loss = s / (float)B;
    }
}

// dq = sum_b sum_splits part_vec (the centred partial sums of head_partial_kernel<2, true>: nothing left to subtract)
__global__ __launch_bounds__(256) void head_dq_kernel(const float* __restrict__ part_vec, float* __restrict__ grads, int B, int splits) {
    const int tid = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int b = 0; b < B; ++b) {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        for (int s = 0; s < splits; ++s) {
            const float* pv = part_vec + ((long)b * splits + s) * D;
            g0 += pv[tid]; g1 += pv[tid + 256]; g2 += pv[tid + 512];
        }
        a0 += g0; a1 += g1; a2 += g2;
    }
    grads[tid] = a0; grads[tid + 256] = a1; grads[tid + 512] = a2;
}

// torch.optim.Adam (L2 weight decay folded into the gradient, bias-corrected), elements [first, first+n)
__global__ void head_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 int first, int n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = first + i;
    const float gr = g[k] + wd * p[k];
    const float mk = b1 * m[k] + (1.f - b1) * gr;
    const float vk = b2 * v[k] + (1.f - b2) * gr * gr;
    m[k] = mk;
    v[k] = vk;
    p[k] -= (lr / bc1) * mk / (sqrtf(vk) / sqrtf(bc2) + eps);
}

int head_fail(int code, const char* msg) {
    snprintf(g_head_err, sizeof g_head_err, "%s", msg);
    return code;
}

}  // namespace

struct loco_head {
    int method;
    float* params = nullptr;  // [q | W | b]
    float* m = nullptr;
    float* v = nullptr;
    long step = 0;
};

namespace {
struct HeadWs {
    float *z, *lse, *pooled, *part_vec, *logits, *dlogits, *dpooled, *loss_b;
    Part* part;
    size_t total;
    int splits;
};
size_t up(size_t n) { return (n + 255) & ~size_t(255); }
HeadWs carve(char* base, int B, int T) {
    HeadWs w;
    w.splits = (T + kRows - 1) / kRows;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += up(bytes); return base ? base + at : (char*)nullptr; };
    w.z = (float*)take((size_t)B * T * 4);
    w.lse = (float*)take((size_t)B * 4);
    w.pooled = (float*)take((size_t)B * D * 4);
    w.part_vec = (float*)take((size_t)B * w.splits * D * 4);
    w.part = (Part*)take((size_t)B * w.splits * sizeof(Part));
    w.logits = (float*)take((size_t)B * C * 4);
    w.dlogits = (float*)take((size_t)B * C * 4);
    w.dpooled = (float*)take((size_t)B * D * 4);
    w.loss_b = (float*)take((size_t)B * 4);
    w.total = o;
    return w;
}

template <bool BWD>
void launch_partial(int method, const float* x, const HeadWs& w, int B, int T, hipStream_t s) {
    dim3 grid(w.splits, B);
    if (method == 0) hipLaunchKernelGGL((head_partial_kernel<0, false>), grid, dim3(256), 0, s, x, w.z, w.lse, w.dpooled, w.pooled, w.part_vec, w.part, T, w.splits);
    else if (method == 1) hipLaunchKernelGGL((head_partial_kernel<1, false>), grid, dim3(256), 0, s, x, w.z, w.lse, w.dpooled, w.pooled, w.part_vec, w.part, T, w.splits);
    else hipLaunchKernelGGL((head_partial_kernel<2, BWD>), grid, dim3(256), 0, s, x, w.z, w.lse, w.dpooled, w.pooled, w.part_vec, w.part, T, w.splits);
}

int head_forward_impl(loco_head* h, const float* x, int B, int T, const HeadWs& w, hipStream_t s) {
    const float* q = h->params;
    const float* W = h->params + D;
    const float* bias = h->params + D + C * D;
    if (h->method == 2) hipLaunchKernelGGL(head_scores_kernel, dim3((unsigned)(((long)B * T + 3) / 4)), dim3(256), 0, s, x, q, w.z, (long)B * T);
    launch_partial<false>(h->method, x, w, B, T, s);
    if (h->method == 0) hipLaunchKernelGGL(head_combine_kernel<0>, dim3(B), dim3(256), 0, s, w.part_vec, w.part, w.pooled, w.lse, T, w.splits);
    else if (h->method == 1) hipLaunchKernelGGL(head_combine_kernel<1>, dim3(B), dim3(256), 0, s, w.part_vec, w.part, w.pooled, w.lse, T, w.splits);
    else hipLaunchKernelGGL(head_combine_kernel<2>, dim3(B), dim3(256), 0, s, w.part_vec, w.part, w.pooled, w.lse, T, w.splits);
    hipLaunchKernelGGL(head_logits_kernel, dim3((B * C + 3) / 4), dim3(256), 0, s, w.pooled, W, bias, w.logits, B);
    return hipGetLastError() == hipSuccess ? LOCO_OK : head_fail(LOCO_E_HIP, "intent head forward launch failed");
}
}  // namespace

extern "C" {

const char* loco_head_last_error(void) { return g_head_err; }

loco_head* loco_head_create(int method) {
    if (method < 0 || method > 2) { head_fail(LOCO_E_INVALID, "method must be 0 (average), 1 (max) or 2 (attention)"); return nullptr; }
    loco_head* h = new loco_head();
    h->method = method;
    const size_t bytes = (size_t)kNParams * sizeof(float);
    if (hipMalloc(&h->params, bytes) != hipSuccess || hipMalloc(&h->m, bytes) != hipSuccess || hipMalloc(&h->v, bytes) != hipSuccess ||
        hipMemset(h->params, 0, bytes) != hipSuccess || hipMemset(h->m, 0, bytes) != hipSuccess || hipMemset(h->v, 0, bytes) != hipSuccess) {
        head_fail(LOCO_E_HIP, "loco_head_create: device allocation failed");
        delete h;
        return nullptr;
    }
    return h;
}

void loco_head_destroy(loco_head* h) {
    if (!h) return;
    (void)hipFree(h->params); (void)hipFree(h->m); (void)hipFree(h->v);
    delete h;
}

int32_t loco_head_num_params(void) { return kNParams; }

int loco_head_set_params(loco_head* h, const float* flat) {
    if (!h || !flat) return head_fail(LOCO_E_INVALID, "null argument");
    if (hipMemcpy(h->params, flat, (size_t)kNParams * 4, hipMemcpyDefault) != hipSuccess) return head_fail(LOCO_E_HIP, "copy failed");
    const size_t bytes = (size_t)kNParams * 4;
    if (hipMemset(h->m, 0, bytes) != hipSuccess || hipMemset(h->v, 0, bytes) != hipSuccess) return head_fail(LOCO_E_HIP, "memset failed");
    h->step = 0;
    return hipDeviceSynchronize() == hipSuccess ? LOCO_OK : head_fail(LOCO_E_HIP, "sync failed");
}

int loco_head_get_params(const loco_head* h, float* flat) {
    if (!h || !flat) return head_fail(LOCO_E_INVALID, "null argument");
    if (hipMemcpy(flat, h->params, (size_t)kNParams * 4, hipMemcpyDefault) != hipSuccess) return head_fail(LOCO_E_HIP, "copy failed");
    return hipDeviceSynchronize() == hipSuccess ? LOCO_OK : head_fail(LOCO_E_HIP, "sync failed");
}

size_t loco_head_workspace_bytes(int32_t B, int32_t T) {
    if (B <= 0 || T <= 0) return 0;
    return carve(nullptr, B, T).total;
}

int loco_head_forward(loco_head* h, const float* x, int32_t B, int32_t T, float* logits, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !x || !logits || !ws || B <= 0 || T <= 0 || B > 65535) return head_fail(LOCO_E_INVALID, "loco_head_forward: invalid argument");
    HeadWs w = carve((char*)ws, B, T);
    if (ws_bytes < w.total) return head_fail(LOCO_E_WORKSPACE, "loco_head_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int rc = head_forward_impl(h, x, B, T, w, s);
    if (rc) return rc;
    if (hipMemcpyAsync(logits, w.logits, (size_t)B * C * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return head_fail(LOCO_E_HIP, "copy failed");
    return LOCO_OK;
}

int loco_head_loss_grad(loco_head* h, const float* x, const float* target, int32_t B, int32_t T, float* loss, float* logits,
                        float* grads, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !x || !target || !loss || !grads || !ws || B <= 0 || T <= 0 || B > 65535)
        return head_fail(LOCO_E_INVALID, "loco_head_loss_grad: invalid argument");
    HeadWs w = carve((char*)ws, B, T);
    if (ws_bytes < w.total) return head_fail(LOCO_E_WORKSPACE, "loco_head_loss_grad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int rc = head_forward_impl(h, x, B, T, w, s);
    if (rc) return rc;
    hipLaunchKernelGGL(head_ce_kernel, dim3(B), dim3(128), 0, s, w.logits, target, w.dlogits, w.loss_b, B);
    hipLaunchKernelGGL(head_param_grads_kernel, dim3(C + B + 1), dim3(256), 0, s, w.dlogits, w.pooled, h->params + D, w.loss_b, grads,
                       w.dpooled, loss, B);
    if (h->method == 2) {
        launch_partial<true>(2, x, w, B, T, s);
        hipLaunchKernelGGL(head_dq_kernel, dim3(1), dim3(256), 0, s, w.part_vec, grads, B, w.splits);
    } else if (hipMemsetAsync(grads, 0, D * 4, s) != hipSuccess) {
        return head_fail(LOCO_E_HIP, "memset failed");
    }
    if (logits && hipMemcpyAsync(logits, w.logits, (size_t)B * C * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return head_fail(LOCO_E_HIP, "copy failed");
    return hipGetLastError() == hipSuccess ? LOCO_OK : head_fail(LOCO_E_HIP, "intent head backward launch failed");
}

int loco_head_adam_step(loco_head* h, const float* grads, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (!h || !grads) return head_fail(LOCO_E_INVALID, "null argument");
    h->step += 1;
    const float bc1 = 1.f - powf(beta1, (float)h->step), bc2 = 1.f - powf(beta2, (float)h->step);
    // q has no gradient under average / max pooling: torch's Adam skips parameters whose .grad is None
    const int first = h->method == 2 ? 0 : D;
    const int n = kNParams - first;
    hipLaunchKernelGGL(head_adam_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->params, grads, h->m, h->v, first, n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    return hipGetLastError() == hipSuccess ? LOCO_OK : head_fail(LOCO_E_HIP, "adam launch failed");
}

}  // extern "C"

// Sample-rate conversion to the model's 16 kHz on the device ("next" row f-4): the step the reference performs on the host with
// librosa.load(path, sr=16000) (/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:56; librosa 0.10.0.post2
// -> soxr 0.3.5 'soxr_hq', requirements.txt:63,138) before the processor sees the clip.  Fisher telephone speech is 8 kHz and
// podcasts are 44.1 kHz, so BASELINE.json configs[2] / configs[3] go through here on real audio.
//
// soxr's source is not in /root/reference and the library is not installed, so its arithmetic cannot be restated line by
// line.  What is restated is its PUBLISHED specification of quality 'HQ' (soxr.h: 20-bit precision, pass-band end 0.913 of
// the lower Nyquist frequency, stop-band begin 1.0, linear phase): one Kaiser-windowed-sinc low-pass, evaluated as a
// rational polyphase filter
//
//     y[n] = sum_m x[m] * h(n*M - m*L)          up = L, down = M, L/M = 16000/sr_in in lowest terms,
//     h(t) = g * sinc(fc * t) * kaiser(t / half),   t in units of 1 / (L * sr_in)
//
// with cut-off in the middle of the transition band, attenuation 20 * 6.02 + ~5 dB and the output length librosa produces
// (ceil(n_in * 16000 / sr_in), resample() + fix_length).  Agreement with soxr itself is therefore at the level of the two
// designs' pass-band ripple / stop-band leakage (~1e-5 relative), not bit for bit: "parity unpinned vs librosa.load" -- the
// oracle (oracle/resample_oracle.py) is an fp64 restatement of THIS specification, and the tests add design-independent
// properties (tone gain, alias rejection, DC gain, agreement with scipy's polyphase engine on the same taps).
//
// Kernel: HBM-bound byte mover with a few hundred MACs per output sample (8 kHz: 180, 44.1 kHz: ~500).  One workgroup = 256
// consecutive output samples: their input window (256 * M / L + taps samples) is staged in LDS with coalesced loads; every
// thread walks its own phase row of the tap table [L][K] (L2-resident: <= 320 KB) with fp32 FMAs in a fixed order.
#include <cmath>
#include <vector>

#include "loco_kernels.h"

namespace loco {

namespace {
double bessel_i0(double x) {  // power series, converges fast for the beta <= 15 used here
    double s = 1.0, t = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) {
        t *= q / ((double)k * k);
        s += t;
        if (t < s * 1e-17) break;
    }
    return s;
}
long gcd_l(long a, long b) { return b ? gcd_l(b, a % b) : a; }
}  // namespace

// Design (host, fp64): fills taps[L][K] (row p = phase, column j = tap) when `taps` is non-null; returns K through *K_out.
// taps[p][j] multiplies x[base - j + K/2] where base = floor(n*M/L), p = (n*M) mod L.
int resample_design(int sr_in, int sr_out, int* L_out, int* M_out, int* K_out, float* taps) {
    if (sr_in <= 0 || sr_out <= 0 || sr_in > 768000) return -1;
    const long g = gcd_l(sr_in, sr_out);
    const int L = (int)(sr_out / g), M = (int)(sr_in / g);
    const double att = 125.0;                                   // dB: 20-bit precision of soxr 'HQ' plus margin
    const double beta = 0.1102 * (att - 8.7);                   // Kaiser's formula for att > 50 dB
    const double nyq_low = 0.5 * (sr_in < sr_out ? sr_in : sr_out);
    const double f_pass = 0.913 * nyq_low, f_stop = 1.0 * nyq_low;
    const double rate = (double)L * sr_in;                       // rate of the zero-stuffed signal the prototype runs at
    const double dw = 2.0 * M_PI * (f_stop - f_pass) / rate;     // transition width, rad / sample
    long N = (long)std::ceil((att - 7.95) / (2.285 * dw)) + 1;   // Kaiser's length estimate (prototype taps)
    int K = (int)((N + L - 1) / L);                              // taps per phase
    K = (K + 3) & ~3;                                            // multiple of 4
    if (K < 8) K = 8;
    if (K > 4096) return -2;
    *L_out = L; *M_out = M; *K_out = K;
    if (!taps) return 0;
    const double half = 0.5 * (double)K * L;                     // half-width of the prototype in its own samples
    const double fc = (f_pass + f_stop) / rate;                  // cut-off (middle of the transition band) as a fraction of rate/2 ... x2 below
    const double i0b = bessel_i0(beta);
    // h(t) = L * (fc) * sinc(fc * t) * w(t): unity DC gain after the zero stuffing (factor L) -- normalised per phase below
    for (int p = 0; p < L; ++p) {
        double sum = 0.0;
        std::vector<double> row(K);
        for (int j = 0; j < K; ++j) {
            // tap j of phase p sits at prototype time t = (j - K/2) * L + p ... relative to the output instant
            const double t = ((double)j - K / 2) * L + p;
            double w = 0.0;
            const double r = t / half;
            if (std::fabs(r) < 1.0) w = bessel_i0(beta * std::sqrt(1.0 - r * r)) / i0b;
            const double a = M_PI * fc * t;
            const double s = std::fabs(a) < 1e-12 ? 1.0 : std::sin(a) / a;
            row[j] = fc * s * w * L;
            sum += row[j];
        }
        (void)sum;
        for (int j = 0; j < K; ++j) taps[(size_t)p * K + j] = (float)row[j];
    }
    return 0;
}

constexpr int kRsBlock = 256;

__global__ __launch_bounds__(kRsBlock) void resample_kernel(const float* __restrict__ x, long n_in, long x_stride, const float* __restrict__ taps,
                                                             int L, int M, int K, float* __restrict__ y, long n_out, long y_stride) {
    extern __shared__ __attribute__((aligned(16))) float win[];
    const int b = blockIdx.y;
    const long n0 = (long)blockIdx.x * kRsBlock;
    const float* xb = x + (long)b * x_stride;
    // input samples needed by outputs n0 .. n0+255: base(n) - K/2 + 1 .. base(n) + K/2, base(n) = floor(n*M/L)
    const long base0 = (n0 * M) / L;
    const long lo = base0 - K / 2 + 1;
    const long n_last = (n0 + kRsBlock - 1 < n_out ? n0 + kRsBlock - 1 : n_out - 1);
    const long hi = (n_last * M) / L + K / 2;  // inclusive
    const int nw = (int)(hi - lo + 1);
    for (int i = threadIdx.x; i < nw; i += kRsBlock) {
        const long m = lo + i;
        win[i] = (m >= 0 && m < n_in) ? xb[m] : 0.f;  // zero extension beyond both ends
    }
    __syncthreads();
    const long n = n0 + threadIdx.x;
    if (n >= n_out) return;
    const long nm = n * M;
    const long base = nm / L;
    const int p = (int)(nm - base * L);
    const float* tp = taps + (size_t)p * K;
    // y[n] = sum_j taps[p][j] * x[base + K/2 - j]: the tap at prototype time (j - K/2) L + p weights the input sample that lies
    // that far BEFORE the output instant n M (in prototype samples): m L = n M - t  ->  m = base - (j - K/2)
    const int w0 = (int)(base + K / 2 - lo);
    float acc = 0.f;
    for (int j = 0; j < K; j += 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(tp + j);
        acc = fmaf(t4[0], win[w0 - j], acc);
        acc = fmaf(t4[1], win[w0 - j - 1], acc);
        acc = fmaf(t4[2], win[w0 - j - 2], acc);
        acc = fmaf(t4[3], win[w0 - j - 3], acc);
    }
    y[(long)b * y_stride + n] = acc;
}

size_t resample_lds_bytes(int L, int M, int K) { return (size_t)(((long)(kRsBlock - 1) * M) / L + K + 8) * sizeof(float); }

hipError_t launch_resample(const float* x, int B, long n_in, long x_stride, const float* taps, int L, int M, int K, float* y, long n_out,
                           long y_stride, hipStream_t s) {
    if (B <= 0 || n_in <= 0 || n_out <= 0 || L <= 0 || M <= 0 || K < 4 || (K & 3)) return hipErrorInvalidValue;
    const size_t lds = resample_lds_bytes(L, M, K);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(resample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid((unsigned)((n_out + kRsBlock - 1) / kRsBlock), (unsigned)B);
    hipLaunchKernelGGL(resample_kernel, grid, dim3(kRsBlock), lds, s, x, n_in, x_stride, taps, L, M, K, y, n_out, y_stride);
    return hipGetLastError();
}

}  // namespace loco

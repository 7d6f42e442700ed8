// How many dependent-chain kernel launches per second does the device retire, from 1..16 streams?  (tools/launch_rate/run.sh)
// Kernels: `spin` cycles of s_sleep-free busy work in `blocks` workgroups of 256 threads -- 0 cycles = the pure launch + completion cost.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void work(float* out, int spin) {
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.678f) out[blockIdx.x] = x;  // never: keeps the loop
}

int main(int argc, char** argv) {
    const int per_stream = argc > 1 ? atoi(argv[1]) : 2000;
    float* out;
    hipMalloc(&out, 1 << 20);
    struct Cfg { int blocks, spin; const char* what; };
    const Cfg cfgs[] = {{1, 0, "1 workgroup, empty"}, {256, 0, "256 workgroups, empty"}, {256, 2000, "256 workgroups x ~28 us"}, {64, 2000, "64 workgroups x ~28 us"},
                        {256, 6000, "256 workgroups x ~82 us"}, {64, 300, "64 workgroups x ~4 us"}, {256, 300, "256 workgroups x ~4 us"}};
    for (const Cfg& c : cfgs) {
        for (int ns : {1, 2, 4, 8, 16}) {
            std::vector<hipStream_t> st(ns);
            for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            for (auto& s : st) for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(work, dim3(c.blocks), dim3(256), 0, s, out, c.spin);
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < per_stream; ++i)
                for (auto& s : st) hipLaunchKernelGGL(work, dim3(c.blocks), dim3(256), 0, s, out, c.spin);
            const auto t1 = std::chrono::steady_clock::now();
            hipDeviceSynchronize();
            const auto t2 = std::chrono::steady_clock::now();
            const double n = (double)per_stream * ns;
            const double host = std::chrono::duration<double>(t1 - t0).count(), tot = std::chrono::duration<double>(t2 - t0).count();
            printf("%-26s %2d streams: %7.2f us per launch (device side, all streams together), %7.0f k launches/s; host enqueue %5.2f us each\n", c.what, ns,
                   1e6 * tot / n, n / tot / 1e3, 1e6 * host / n);
            for (auto& s : st) hipStreamDestroy(s);
        }
    }
    return 0;
}

// Micro-benchmark: sustained v_fma_f64 rate, alone and interleaved 1:1 with 32-bit integer VALU ops, at 1 and 4
// waves per SIMD.  hipcc -O3 --offload-arch=gfx950 fp64_rate.hip -o fp64_rate && ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MIX>
__global__ void __launch_bounds__(1024) k(double* out, int iters, double seed) {
    double a[8];
    int q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-9 + i; q[i] = threadIdx.x + i; }
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a[i] = fma(a[i], m, c);
                if (MIX) q[i] = q[i] * 3 + (q[i] >> 7);  // v_mad / shift: 2 int ops per fma
            }
        }
    }
    double s = 0; int t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += a[i]; t += q[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + t;
}

template <int MIX>
void run(int threads, const char* tag) {
    const int blocks = 256 * 4, iters = 20000;
    double* d; hipMalloc(&d, sizeof(double) * blocks * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fmas = (double)blocks * threads * iters * 32.0;
    printf("%-28s threads/WG %4d : %7.2f ms  %6.2f TFLOP/s fp64 (fma=2)\n", tag, threads, ms, 2 * fmas / ms / 1e9);
    hipFree(d);
}
int main() {
    run<0>(256, "fma_f64 only");   // 1 wave / SIMD per WG, 4 WG per CU possible
    run<0>(1024, "fma_f64 only");
    run<1>(256, "fma_f64 + 2 int ops");
    run<1>(1024, "fma_f64 + 2 int ops");
    return 0;
}

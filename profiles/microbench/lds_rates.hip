// LDS read cost on gfx950 for the access patterns of the sweep kernel's tap gathers (one workgroup of 1024 threads per
// CU, 16 reads per loop iteration, one s_waitcnt per iteration).  Addresses are in 8-byte elements.
// hipcc -O3 --offload-arch=gfx950 lds_rates.hip -o lds_rates && ./lds_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(S) S S S S S S S S S S S S S S S S

enum { B64 = 0, B128 = 1, B64X2 = 2, READ2 = 3 };

template <int KIND>
__global__ void __launch_bounds__(1024) k(double* out, int iters, int pattern, int pitch) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = (double)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int el;
    switch (pattern) {
        case 0: el = lane; break;                                   // stride 1
        case 1: el = 2 * lane; break;                               // stride 2, aligned pairs
        case 2: el = 2 * lane + 1; break;                           // stride 2, odd start (b128 misaligned by 8 B)
        case 3: el = 2 * (lane & 15) + pitch * 2 * (lane >> 4); break;          // 16 x 4 lag patch, rows 2 apart
        case 4: el = 2 * (lane & 15) + 1 + pitch * 2 * (lane >> 4); break;      // same, odd columns
        case 5: el = 2 * (lane & 15) + ((lane >> 4) & 1) + pitch * 2 * (lane >> 4); break;  // rows alternate parity
        case 6: el = (lane & 15) * 2 + ((lane & 15) > 7) + pitch * (2 * (lane >> 4) + ((lane & 15) > 11)); break;  // jitter
        default: el = 3 * lane; break;
    }
    unsigned addr = (unsigned)(uintptr_t)lds + 8u * (unsigned)(el % 4000);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (KIND == B64) {
            REP16(asm volatile("ds_read_b64 %0, %1" : "=v"(a0) : "v"(addr));)
        } else if (KIND == B128) {
            REP16(asm volatile("ds_read_b128 %0, %1" : "=v"(*(double2*)&a0) : "v"(addr));)
        } else if (KIND == B64X2) {
            REP16(asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=&v"(a0), "=&v"(a2) : "v"(addr));)
        } else {
            REP16(asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(*(double2*)&a0) : "v"(addr));)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        acc += a0 + a1 + a2 + a3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc / iters;
}

template <int KIND>
void run(const char* name, int pattern, int pitch, int reads_per_rep) {
    const int blocks = 256, iters = 2000;
    double* d;
    (void)hipMalloc(&d, sizeof(double) * blocks * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 65536, 0, d, 20, pattern, pitch);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 65536, 0, d, iters, pattern, pitch);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double h[4];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double inst_per_cu = 16.0 * iters * 16.0 * reads_per_rep;  // 16 waves per CU
    printf("%-14s pattern %d pitch %3d: %8.3f ms  %6.2f ns per wave-read per CU  (lane0..2 = %.0f %.0f %.0f)\n", name,
           pattern, pitch, ms, ms * 1e6 / inst_per_cu, h[0], h[1], h[2]);
    (void)hipFree(d);
}
int main() {
    run<B64>("ds_read_b64", 0, 0, 1);
    run<B64>("ds_read_b64", 1, 0, 1);
    run<B64>("ds_read_b64", 7, 0, 1);
    for (int pitch : {64, 65, 77, 80, 96}) {
        run<B64>("ds_read_b64", 3, pitch, 1);
        run<B64>("ds_read_b64", 5, pitch, 1);
        run<B64>("ds_read_b64", 6, pitch, 1);
    }
    run<B128>("ds_read_b128", 1, 0, 1);
    run<B128>("ds_read_b128", 2, 0, 1);
    for (int pitch : {64, 65, 77, 96}) {
        run<B128>("ds_read_b128", 3, pitch, 1);
        run<B128>("ds_read_b128", 4, pitch, 1);
        run<B128>("ds_read_b128", 5, pitch, 1);
        run<B128>("ds_read_b128", 6, pitch, 1);
    }
    run<B64X2>("2x ds_read_b64", 1, 0, 2);
    run<B64X2>("2x ds_read_b64", 3, 64, 2);
    run<READ2>("ds_read2_b64", 1, 0, 1);
    run<READ2>("ds_read2_b64", 2, 0, 1);
    return 0;
}

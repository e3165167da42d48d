// How far do float64 VALU work and LDS reads overlap on one gfx950 CU with the occupancy k_sweep has (one workgroup of
// 1024 threads = 4 waves per SIMD)?  Every iteration is one "sample" of the sweep kernel in miniature: NR ds_read_b64,
// NW float64 FMAs that do not need the loaded values (the spline weights), one s_waitcnt, NV - NW FMAs that consume them.
//   serial:     reads, weights, wait, rest                      (the structure of k_sweep)
//   pipelined:  the NEXT iteration's reads are issued before this iteration's wait (two register sets, lgkmcnt(NR))
//   valu only / lds only: the two halves alone
// and with conflict-free (stride 1) or two-way conflicting (stride 2) addresses.  Cycles are per iteration per wave at
// the 2.4 GHz the counters are priced against; "SIMD round" = 4 waves, "CU round" = 16 waves.
// hipcc -O3 --offload-arch=gfx950 valu_lds_overlap.hip -o valu_lds_overlap && ./valu_lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

// how the nine taps are fetched: 9 x ds_read_b64 (k_sweep), 9 x ds_read_b32 (half the bytes come back), 5 x ds_read_b128
// (16-byte aligned addresses only: 10 values, fewer instructions), 4 x ds_read2_b64 + 1 x ds_read_b64 (what hipcc fuses
// neighbouring reads into)
enum { RD_B64 = 0, RD_B32 = 1, RD_B128 = 2, RD_READ2 = 3 };
typedef double double2v __attribute__((ext_vector_type(2)));
struct Taps9 {
    double t[10];
    template <int RD>
    __device__ __forceinline__ void issue(unsigned a, double& p0, double& p1) {
        if (RD == RD_B64) {
            asm volatile(
                "ds_read_b64 %0, %11\n\tds_read_b64 %1, %11 offset:8\n\tds_read_b64 %2, %11 offset:16\n\t"
                "ds_read_b64 %3, %11 offset:968\n\tds_read_b64 %4, %11 offset:976\n\tds_read_b64 %5, %11 offset:984\n\t"
                "ds_read_b64 %6, %11 offset:1936\n\tds_read_b64 %7, %11 offset:1944\n\tds_read_b64 %8, %11 offset:1952"
                : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
                  "=&v"(t[8]), "+v"(p0), "+v"(p1)
                : "v"(a));
        } else if (RD == RD_B32) {
            // (no conversion: only the low dword of each tap's register pair is replaced, the arithmetic stays float64)
            typedef unsigned uint2v __attribute__((ext_vector_type(2)));
            unsigned f[9];
            asm volatile(
                "ds_read_b32 %0, %11\n\tds_read_b32 %1, %11 offset:8\n\tds_read_b32 %2, %11 offset:16\n\t"
                "ds_read_b32 %3, %11 offset:968\n\tds_read_b32 %4, %11 offset:976\n\tds_read_b32 %5, %11 offset:984\n\t"
                "ds_read_b32 %6, %11 offset:1936\n\tds_read_b32 %7, %11 offset:1944\n\tds_read_b32 %8, %11 offset:1952"
                : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7]),
                  "=&v"(f[8]), "+v"(p0), "+v"(p1)
                : "v"(a));
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                uint2v u = __builtin_bit_cast(uint2v, t[k]);
                u.x = f[k];
                t[k] = __builtin_bit_cast(double, u);
            }
        } else if (RD == RD_B128) {
            double2v q0, q1, q2, q3, q4;
            asm volatile(
                "ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:16\n\tds_read_b128 %2, %7 offset:976\n\t"
                "ds_read_b128 %3, %7 offset:1936\n\tds_read_b128 %4, %7 offset:1952"
                : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "+v"(p0), "+v"(p1)
                : "v"(a));
            t[0] = q0.x; t[1] = q0.y; t[2] = q1.x; t[3] = q1.y; t[4] = q2.x; t[5] = q2.y; t[6] = q3.x; t[7] = q3.y;
            t[8] = q4.x; t[9] = q4.y;
        } else {
            double2v q0, q1, q2, q3;
            asm volatile(
                "ds_read2_b64 %0, %7 offset0:0 offset1:1\n\tds_read2_b64 %1, %7 offset0:2 offset1:121\n\t"
                "ds_read2_b64 %2, %7 offset0:122 offset1:123\n\tds_read2_b64 %3, %7 offset0:242 offset1:243\n\t"
                "ds_read_b64 %4, %7 offset:1952"
                : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(t[8]), "+v"(p0), "+v"(p1)
                : "v"(a));
            t[0] = q0.x; t[1] = q0.y; t[2] = q1.x; t[3] = q1.y; t[4] = q2.x; t[5] = q2.y; t[6] = q3.x; t[7] = q3.y;
        }
    }
    // N = instructions of the NEXT sample that may stay in flight (0: wait for everything)
    template <int N>
    __device__ __forceinline__ void wait(double& w0, double& w1, double& w2, double& w3, double& w4, double& w5) {
        if (N == 0)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                           "+v"(t[8]), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5));
        else if (N == 9)
            asm volatile("s_waitcnt lgkmcnt(9)"
                         : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                           "+v"(t[8]), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5));
        else
            asm volatile("s_waitcnt lgkmcnt(5)"
                         : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                           "+v"(t[8]), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5));
    }
};

// the 10 weight operations of gather_o2 (5 per axis) and the 27 that follow (12 gather + 5 sums + 10 more, to make 37)
__device__ __forceinline__ void weights(double fx, double fy, double* wx, double* wy) {
    const double gx = 1.0 - fx, gy = 1.0 - fy;
    wx[0] = gx * gx;
    wx[2] = fx * fx;
    wx[1] = (2.0 - wx[0]) - wx[2];
    wy[0] = gy * gy;
    wy[2] = fy * fy;
    wy[1] = (2.0 - wy[0]) - wy[2];
}
__device__ __forceinline__ void consume(const double* t, const double* wx, const double* wy, double* acc, double av) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double row = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) row = fma(t[r * 3 + c], wx[c], row);
        v = fma(row, wy[r], v);
    }
    acc[0] += av;
    acc[1] += v;
    acc[2] = fma(av, av, acc[2]);
    acc[3] = fma(v, v, acc[3]);
    acc[4] = fma(av, v, acc[4]);
    // stand-ins for the coordinate / address / mask instructions of the loop (10 more VALU operations)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        acc[5 + k] = fma(acc[5 + k], 1.0000001, av);
        acc[10 + k] = fma(acc[10 + k], 0.9999999, av);
    }
}

enum { SERIAL = 0, PIPELINED = 1, VALU_ONLY = 2, LDS_ONLY = 3 };

template <int KIND, int RD>
__global__ void __launch_bounds__(1024) k(double* out, int iters, int stride) {
    constexpr int NI = (RD == RD_B64 || RD == RD_B32) ? 9 : 5;  // LDS instructions per sample
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = 1.0 + 1e-9 * (double)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)lds + 8u * (unsigned)(stride * lane + 8 * (threadIdx.x >> 6));
    double acc[15];
#pragma unroll
    for (int k2 = 0; k2 < 15; ++k2) acc[k2] = 0.0;
    double fx = 0.25 + 1e-6 * lane, fy = 0.75 - 1e-6 * lane;
    const double av = 1.0 + 1e-3 * lane;
    Taps9 A, B;
#pragma unroll
    for (int k2 = 0; k2 < 10; ++k2) A.t[k2] = B.t[k2] = 1.0;
    double wx[3], wy[3];
    if (KIND == PIPELINED) A.template issue<RD>(addr, fx, fy);
    for (int it = 0; it < iters; it += 2) {
        if (KIND == SERIAL) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                A.template issue<RD>(addr, fx, fy);
                weights(fx, fy, wx, wy);
                A.template wait<0>(wx[0], wx[1], wx[2], wy[0], wy[1], wy[2]);
                consume(A.t, wx, wy, acc, av);
            }
        } else if (KIND == PIPELINED) {
            B.template issue<RD>(addr, fx, fy);
            weights(fx, fy, wx, wy);
            A.template wait<NI>(wx[0], wx[1], wx[2], wy[0], wy[1], wy[2]);
            consume(A.t, wx, wy, acc, av);
            A.template issue<RD>(addr, fx, fy);
            weights(fx, fy, wx, wy);
            B.template wait<NI>(wx[0], wx[1], wx[2], wy[0], wy[1], wy[2]);
            consume(B.t, wx, wy, acc, av);
        } else if (KIND == VALU_ONLY) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm volatile("" : "+v"(fx), "+v"(fy), "+v"(A.t[0]), "+v"(A.t[1]), "+v"(A.t[2]), "+v"(A.t[3]), "+v"(A.t[4]),
                             "+v"(A.t[5]), "+v"(A.t[6]), "+v"(A.t[7]), "+v"(A.t[8]));
                weights(fx, fy, wx, wy);
                asm volatile("" : "+v"(wx[0]), "+v"(wx[1]), "+v"(wx[2]), "+v"(wy[0]), "+v"(wy[1]), "+v"(wy[2]));
                consume(A.t, wx, wy, acc, av);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                A.template issue<RD>(addr, fx, fy);
                wx[0] = wx[1] = wx[2] = wy[0] = wy[1] = wy[2] = 0.0;
                A.template wait<0>(wx[0], wx[1], wx[2], wy[0], wy[1], wy[2]);
            }
        }
    }
    if (KIND == PIPELINED) A.template wait<0>(wx[0], wx[1], wx[2], wy[0], wy[1], wy[2]);
    double s = 0.0;
#pragma unroll
    for (int k2 = 0; k2 < 15; ++k2) s += acc[k2];
#pragma unroll
    for (int k2 = 0; k2 < 10; ++k2) s += A.t[k2] + B.t[k2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int RD>
double run(const char* name, int stride) {
    const int blocks = 256, iters = 4000;
    double* d;
    (void)hipMalloc(&d, sizeof(double) * blocks * 1024);
    (void)hipFuncSetAttribute((const void*)k<KIND, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {  // (the first launches ramp the clocks)
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, RD>), dim3(blocks), dim3(1024), 159 * 1024, 0, d, iters, stride);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double ns_it = best * 1e6 / iters;          // per iteration of every wave (all waves run concurrently)
    const double cyc = ns_it * 2.4;                   // cycles per iteration per wave at 2.4 GHz
    printf("%-10s stride %d: %7.3f ms   %6.1f cycles per sample per wave  = %5.1f per SIMD round of 4 waves / 4\n", name,
           stride, best, cyc, cyc / 4.0);
    (void)hipFree(d);
    return cyc;
}

template <int RD>
void flavour(const char* what, int stride) {
    printf("--- %s, lane stride %d elements\n", what, stride);
    const double v = run<VALU_ONLY, RD>("valu only", stride);
    const double l = run<LDS_ONLY, RD>("lds only", stride);
    const double s = run<SERIAL, RD>("serial", stride);
    const double p = run<PIPELINED, RD>("pipelined", stride);
    constexpr int ni = (RD == RD_B64 || RD == RD_B32) ? 9 : 5;
    printf("    max(valu, lds) = %.1f, sum = %.1f; serial = %.2f x max, pipelined = %.2f x max;  beyond the VALU-only loop: "
           "%.1f / %.1f cycles per LDS instruction per wave (serial / pipelined)\n",
           v > l ? v : l, v + l, s / (v > l ? v : l), p / (v > l ? v : l), (s - v) / (4.0 * ni), (p - v) / (4.0 * ni));
}

int main() {
    // (clock ramp: the first kernels of a process run slower)
    for (int i = 0; i < 3; ++i) (void)run<VALU_ONLY, RD_B64>("warm-up", 1);
    flavour<RD_B64>("9 x ds_read_b64", 1);
    flavour<RD_B64>("9 x ds_read_b64", 2);
    flavour<RD_B32>("9 x ds_read_b32", 1);
    flavour<RD_B32>("9 x ds_read_b32", 2);
    flavour<RD_B128>("5 x ds_read_b128 (aligned)", 2);
    flavour<RD_READ2>("4 x ds_read2_b64 + ds_read_b64", 1);
    flavour<RD_READ2>("4 x ds_read2_b64 + ds_read_b64", 2);
    return 0;
}

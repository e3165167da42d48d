// Per-instruction issue cost on gfx950 relative to v_fma_f64: 16 independent ops per loop iteration, 4 waves/SIMD.
// hipcc -O3 --offload-arch=gfx950 op_rates.hip -o op_rates && ./op_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(S) S S S S S S S S S S S S S S S S

template <int OP>
__global__ void __launch_bounds__(1024) k(double* out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-6, b = 0.5;
    float f = 1.5f + threadIdx.x;
    int i = threadIdx.x + 7, j = 3;
    unsigned long long m = 0;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_floor_f64 %0, %1" : "=v"(b) : "v"(a));) }
        if (OP == 2) { REP16(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i) : "v"(a));) }
        if (OP == 3) { REP16(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(b) : "v"(i));) }
        if (OP == 4) { REP16(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a));) }
        if (OP == 5) { REP16(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(b) : "v"(f));) }
        if (OP == 6) { REP16(asm volatile("v_cmp_class_f64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(j));) }
        if (OP == 7) { REP16(asm volatile("v_rcp_f64 %0, %1" : "=v"(b) : "v"(a));) }
        if (OP == 8) { REP16(asm volatile("v_fract_f64 %0, %1" : "=v"(b) : "v"(a));) }
        if (OP == 9) { REP16(asm volatile("v_mul_lo_u32 %0, %1, %1" : "=v"(j) : "v"(i));) }
        if (OP == 10) { REP16(asm volatile("v_mad_i32_i24 %0, %1, %1, %1" : "=v"(j) : "v"(i));) }
        if (OP == 11) { REP16(asm volatile("v_lshl_add_u32 %0, %1, 3, %1" : "=v"(j) : "v"(i));) }
        if (OP == 12) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 13) { REP16(asm volatile("v_cmp_le_f64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));) }
        if (OP == 14) { REP16(asm volatile("v_add_u32 %0, %1, %1" : "=v"(j) : "v"(i));) }
        if (OP == 15) { REP16(asm volatile("v_mul_f64 %0, %1, %1" : "=v"(b) : "v"(a));) }
        if (OP == 16) { REP16(asm volatile("v_cndmask_b32 %0, %1, %1, vcc" : "=v"(j) : "v"(i));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + f + i + j + (double)m;
}

template <int OP>
double run(const char* name, double base) {
    const int blocks = 256, iters = 4000;
    double* d;
    (void)hipMalloc(&d, sizeof(double) * blocks * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // cycles per wave-instruction per SIMD at 2.1 GHz nominal: 4 waves/SIMD issue back to back
    const double inst_per_simd = 4.0 * iters * 16.0;
    const double ns = ms * 1e6 / inst_per_simd;
    printf("%-18s %8.3f ms   %6.2f ns/wave-instr/SIMD   x%.2f of v_fma_f64\n", name, ms, ns, base > 0 ? ns / base : 1.0);
    (void)hipFree(d);
    return ns;
}
int main() {
    const double b = run<0>("v_fma_f64", 0);
    run<12>("v_add_f64", b); run<15>("v_mul_f64", b); run<1>("v_floor_f64", b); run<8>("v_fract_f64", b);
    run<2>("v_cvt_i32_f64", b); run<3>("v_cvt_f64_i32", b); run<4>("v_cvt_f32_f64", b); run<5>("v_cvt_f64_f32", b);
    run<6>("v_cmp_class_f64", b); run<13>("v_cmp_le_f64", b); run<7>("v_rcp_f64", b);
    run<9>("v_mul_lo_u32", b); run<10>("v_mad_i32_i24", b); run<11>("v_lshl_add_u32", b); run<14>("v_add_u32", b);
    run<16>("v_cndmask_b32", b);
    return 0;
}

// Window staging into LDS: register path (global_load float -> cvt -> ds_write_b64, what k_sweep does) against LDS-DMA
// (global_load_lds_dword from a pre-converted float64 image; wave-uniform LDS base + lane * 4, so one instruction fills
// 32 consecutive doubles of a window row of odd pitch).  hipcc -O3 --offload-arch=gfx950 glds_stage.hip -o glds_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kWaves = 16;

template <int METHOD>
__global__ void __launch_bounds__(1024) k(const float* img32, const double* img64, int W, int H, int ww, int wh, int pitch,
                                          int iters, double* out) {
    extern __shared__ __align__(16) unsigned char raw[];
    double* lds = (double*)raw;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        const int ox = (blockIdx.x * 37 + it * 11) % (W - ww), oy = (blockIdx.x * 53 + it * 7) % (H - wh);
        if (METHOD == 0) {
            for (int r = wave; r < wh; r += kWaves) {
                const float* src = img32 + (size_t)(oy + r) * W + ox;
                for (int c = lane; c < ww; c += 64) lds[r * pitch + c] = ((double)src[c] - 100.0) * 0.25;
            }
        } else if (METHOD == 1) {
            for (int r = wave; r < wh; r += kWaves) {
                const double* src = img64 + (size_t)(oy + r) * W + ox;
                for (int c0 = 0; c0 < ww; c0 += 32) {
                    const int c = c0 + (lane >> 1);
                    if (c < ww) {
                        const char* g = (const char*)(src + c) + (lane & 1) * 4;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                         (__attribute__((address_space(3))) void*)(lds + r * pitch + c0), 4, 0,
                                                         0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (METHOD == 2) {
            // one global_load_lds_dwordx4 per row: 64 lanes x 16 B = 128 doubles, rows of pitch >= 128 (odd pitch: the
            // LDS base of odd rows is only 8-byte aligned)
            for (int r = wave; r < wh; r += kWaves) {
                const double* src = img64 + (size_t)(oy + r) * W + ox;
                const int c = 2 * lane;
                if (c < ww) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                     (__attribute__((address_space(3))) void*)(lds + r * pitch), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (METHOD == 1) {
        }
        __syncthreads();
        // consume something so that the staging is not dead, and check it
        const int r = (threadIdx.x * 7) % wh, c = (threadIdx.x * 13) % ww;
        acc += lds[r * pitch + c];
        __syncthreads();
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main() {
    const int W = 2048, H = 2048, ww = 116, wh = 143, pitch = 117, blocks = 256, iters = 200;
    std::vector<float> h32((size_t)W * H);
    std::vector<double> h64((size_t)W * H);
    for (size_t i = 0; i < h32.size(); ++i) {
        h32[i] = (float)((i * 2654435761u) % 4096) * 0.5f;
        h64[i] = ((double)h32[i] - 100.0) * 0.25;
    }
    float* d32;
    double *d64, *dout;
    (void)hipMalloc(&d32, h32.size() * 4);
    (void)hipMalloc(&d64, h64.size() * 8);
    (void)hipMalloc(&dout, sizeof(double) * blocks * 1024);
    (void)hipMemcpy(d32, h32.data(), h32.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d64, h64.data(), h64.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> r0(blocks * 1024), r1(blocks * 1024), r2(blocks * 1024);
    for (int m = 0; m < 3; ++m) {
        const int p = m == 2 ? 129 : pitch;
        const size_t lds_bytes = (size_t)p * wh * 8;
        (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), lds_bytes, 0, d32, d64, W, H, ww, wh, p, iters, dout);
            else if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), lds_bytes, 0, d32, d64, W, H, ww, wh, p, iters, dout);
            else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), lds_bytes, 0, d32, d64, W, H, ww, wh, p, iters, dout);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
        }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(m == 0 ? r0.data() : (m == 1 ? r1.data() : r2.data()), dout, sizeof(double) * blocks * 1024,
                        hipMemcpyDeviceToHost);
        printf("%s: %.3f ms for %d windows per CU -> %.2f us per window (err %s)\n",
               m == 0 ? "register staging  " : (m == 1 ? "LDS-DMA dword     " : "LDS-DMA x4, p=129 "), ms, iters, ms * 1e3 / iters,
               hipGetErrorString(hipGetLastError()));
    }
    size_t bad2 = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad2 += r0[i] != r2[i];
    printf("mismatches register vs x4: %zu\n", bad2);
    size_t bad = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
    printf("mismatches between the two stagings: %zu of %zu\n", bad, r0.size());
    return 0;
}

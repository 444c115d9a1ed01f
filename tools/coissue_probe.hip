// Probe: does fp64 VALU work on a second wave of the same SIMD slow down fp64 MFMA issue on gfx950?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run the MFMA chain, waves 4-7 (their
// SIMD partners) run mode 0: nothing, 1: v_fma_f64 chain, 2: v_fma_f32 chain, 3: LDS traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_mix(int iters, int mode, int valu_iters, double* sink, unsigned long long* clk) {
    __shared__ double buf[512];
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        v4d acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = v4d{0, 0, 0, 0};
        double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        double s = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        if (s == 12345.678) sink[0] = s;
        if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
    } else if (mode == 1) {
        double acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = threadIdx.x * 1e-3 + t;
        double a = 1.0000001, b = 1e-9;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = __builtin_fma(acc[t], a, b);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        double s = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += acc[t];
        if (s == 12345.678) sink[1] = s;
        if (blockIdx.x == 0 && threadIdx.x == 256) clk[1] = t1 - t0;
    } else if (mode == 2) {
        float acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = threadIdx.x * 1e-3f + t;
        float a = 1.0000001f, b = 1e-9f;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = __builtin_fmaf(acc[t], a, b);
        }
        float s = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += acc[t];
        if (s == 12345.678f) sink[1] = s;
    } else if (mode == 3) {
        double s = 0;
        for (int it = 0; it < valu_iters * 4; ++it) {
            buf[threadIdx.x] = s + it;
            s += buf[(threadIdx.x * 7 + it) & 511];
        }
        if (s == 12345.678) sink[1] = s;
    }
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    double* sink; unsigned long long* clk; hipMalloc(&sink, 16); hipMalloc(&clk, 16);
    const int iters = 20000;
    const char* names[] = {"partner idle", "partner v_fma_f64", "partner v_fma_f32", "partner LDS"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int vi : {iters * 4, iters * 16}) {
            if (mode == 0 && vi != iters * 4) continue;
            hipMemset(clk, 0, 16);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_mix, dim3(cus), dim3(512), 0, 0, iters / 10, mode, vi / 10, sink, clk);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mix, dim3(cus), dim3(512), 0, 0, iters, mode, vi, sink, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            double cyc = (double)h[0] / ((double)iters * 8);
            printf("%-20s valu_iters=%7d  kernel %8.3f ms  mfma wave: %6.1f cyc/mfma (%5.1f TF while running)", names[mode], vi, ms, cyc,
                   cus * 4 * 2048.0 / cyc * 100e6 * 24 / 1e12 / 1.0 * 1.0);
            if (mode == 1) printf("  fma wave: %.2f cyc/fma", (double)h[1] / ((double)vi * 16));
            printf("\n");
        }
    }
    return 0;
}

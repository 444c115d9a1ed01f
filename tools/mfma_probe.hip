// Probe: fp64 MFMA vs VALU FMA issue rates on gfx950, at 1/2/4 waves per SIMD, with in-kernel clocks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_mfma(int iters, double* sink, unsigned long long* clk) {
    v4d acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = v4d{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

__global__ void k_fma(int iters, double* sink, unsigned long long* clk) {
    double acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = threadIdx.x * 1e-3 + t;
    double a = 1.0000001, b = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = __builtin_fma(acc[t], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t];
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename F>
void run(const char* name, F launch, double flops) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.3f ms  %8.2f TFLOP/s", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("CUs=%d clock=%d kHz\n", cus, p.clockRate);
    double* sink; unsigned long long* clk; hipMalloc(&sink, 8); hipMalloc(&clk, 16);
    unsigned long long h[2];
    int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD
        for (int nacc : {4, 16}) {
            char nm[96]; snprintf(nm, sizeof nm, "mfma_f64 16x16x4 waves/SIMD=%d nacc=%d", wps, nacc);
            int blocks = cus * wps;
            double fl = (double)blocks * 4 * iters * nacc * 2048.0;
            if (nacc == 4) run(nm, [&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, iters, sink, clk); }, fl);
            else run(nm, [&] { hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(256), 0, 0, iters, sink, clk); }, fl);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("  cyc/mfma/wave=%.1f  clk=%.0f MHz\n", (double)h[0] / ((double)iters * nacc), (double)h[0] / (double)h[1] * 100.0);
        }
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        char nm[96]; snprintf(nm, sizeof nm, "v_fma_f64 waves/SIMD=%d", wps);
        int blocks = cus * wps;
        double fl = (double)blocks * 256 * (double)iters * 16 * 2.0;
        run(nm, [&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, iters, sink, clk); }, fl);
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        printf("  cyc/fma/wave=%.2f  clk=%.0f MHz\n", (double)h[0] / ((double)iters * 16), (double)h[0] / (double)h[1] * 100.0);
    }
    return 0;
}

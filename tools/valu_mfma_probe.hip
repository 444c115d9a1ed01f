// fp64 MFMA issue on MI355X, measured per wave with the 100 MHz wall clock:
//   (1) the v_mfma_f64_16x16x4 stream of ONE wave per SIMD, 4 or 8 independent accumulators, and of TWO waves per SIMD;
//   (2) the same stream with a second wave on the SIMD that runs dependence-free fp64 (or fp32) FMAs: does vector fp64 work of
//       another wave overlap with the matrix pipe (MI355X quotes the same 78.6 TFLOP/s for vector and matrix fp64)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_mfma_probe tools/valu_mfma_probe.hip && /tmp/valu_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
// mode: 0 = companions idle, 1 = fp64 FMA companions, 2 = fp32 FMA companions, 3 = companions run MFMAs too; acc8: 8 accumulators
__global__ __launch_bounds__(512) void probe(double* out, unsigned long long* ticks, int iters, int mode, int acc8) {
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = wall_clock64();
    double res = 0.0;
    if (wave < 4 || mode == 3) {
        v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
        double x = threadIdx.x * 1e-3, y = 1.0 + x;
        if (acc8) {
            for (int i = 0; i < iters / 2; ++i) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a4, 0, 0, 0);
                a5 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a5, 0, 0, 0);
                a6 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a6, 0, 0, 0);
                a7 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a7, 0, 0, 0);
            }
        } else {
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            }
        }
        res = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[1] + a6[2] + a7[3];
    } else if (mode == 1) {
        double f0 = threadIdx.x, f1 = 1.5, f2 = 2.5, f3 = 3.5, f4 = 4.5, f5 = 5.5, f6 = 6.5, f7 = 7.5;
        const double m = 1.0000001, c = 1e-9;
        for (int i = 0; i < iters; ++i) {
            f0 = __builtin_fma(f0, m, c); f1 = __builtin_fma(f1, m, c); f2 = __builtin_fma(f2, m, c); f3 = __builtin_fma(f3, m, c);
            f4 = __builtin_fma(f4, m, c); f5 = __builtin_fma(f5, m, c); f6 = __builtin_fma(f6, m, c); f7 = __builtin_fma(f7, m, c);
        }
        res = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    } else if (mode == 2) {
        float f0 = threadIdx.x, f1 = 1.5f, f2 = 2.5f, f3 = 3.5f, f4 = 4.5f, f5 = 5.5f, f6 = 6.5f, f7 = 7.5f;
        const float m = 1.0001f, c = 1e-5f;
        for (int i = 0; i < iters; ++i) {
            f0 = __builtin_fmaf(f0, m, c); f1 = __builtin_fmaf(f1, m, c); f2 = __builtin_fmaf(f2, m, c); f3 = __builtin_fmaf(f3, m, c);
            f4 = __builtin_fmaf(f4, m, c); f5 = __builtin_fmaf(f5, m, c); f6 = __builtin_fmaf(f6, m, c); f7 = __builtin_fmaf(f7, m, c);
        }
        res = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    }
    const unsigned long long t1 = wall_clock64();
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    double* out;
    unsigned long long* ticks;
    (void)hipMalloc(&out, 256 * 512 * 8);
    (void)hipMalloc(&ticks, 256 * 8 * 8);
    const int iters = 20000;
    std::vector<unsigned long long> h(256 * 8);
    struct { int mode, acc8; const char* name; } cases[] = {
        {0, 0, "1 MFMA wave/SIMD, 4 accumulators"}, {0, 1, "1 MFMA wave/SIMD, 8 accumulators"},   // (more MFMA waves: mfma_rate_probe.hip)
        {1, 0, "1 MFMA wave + 1 fp64-FMA wave per SIMD"}, {2, 0, "1 MFMA wave + 1 fp32-FMA wave per SIMD"}};
    for (auto& cs : cases) {
        probe<<<256, 512>>>(out, ticks, 100, cs.mode, cs.acc8);
        probe<<<256, 512>>>(out, ticks, iters, cs.mode, cs.acc8);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
        double tm = 0, tc = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? tm : tc) += h[b * 8 + w] * 1e-8 / (256 * 4);   // 100 MHz ticks -> s, mean
        printf("%-44s MFMA waves %7.3f ms -> %5.1f TFLOP/s", cs.name, tm * 1e3, 256.0 * 4 * iters * 4 * 2048 / tm / 1e12);
        if (cs.mode == 1 || cs.mode == 2)
            printf("   companion waves %7.3f ms for %d FMAs each = %.0f ns per FMA (alone: 4 cycles = 1.7 ns)", tc * 1e3, iters * 8, tc * 1e9 / (iters * 8));
        printf("\n");
    }
    return 0;
}

// How many v_mfma_f64_16x16x4 per second does a SIMD of MI355X retire as a function of the waves that feed it?
// W waves per SIMD (W = 1, 2, 3, 4; one workgroup of 256·W threads per CU), each issuing a dependence-free stream over 4 accumulators.
// Timed with HIP events around the launch (all 256 CUs busy) and per wave with the 100 MHz wall clock.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate_probe tools/mfma_rate_probe.hip && /tmp/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void probe(double* out, unsigned long long* ticks, int iters) {
    const unsigned long long t0 = wall_clock64();
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + x;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = wall_clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    double* out;
    unsigned long long* ticks;
    (void)hipMalloc(&out, 256 * 1024 * 8);
    (void)hipMalloc(&ticks, 256 * 16 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<unsigned long long> h(256 * 16);
    for (int nwg : {256, 64}) {
        for (int W = 1; W <= 4; ++W) {
            probe<<<nwg, 256 * W>>>(out, ticks, 100);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            probe<<<nwg, 256 * W>>>(out, ticks, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
            double tw = 0;
            for (int b = 0; b < nwg; ++b)
                for (int w = 0; w < 4 * W; ++w) tw += h[b * 16 + w] * 1e-8 / (nwg * 4 * W);
            const double mfmas = (double)nwg * 4 * W * iters * 4;
            printf("%3d workgroups, %d wave(s)/SIMD: launch %7.3f ms (mean wave %7.3f ms) -> %6.1f TFLOP/s, %5.1f ns per MFMA and SIMD\n", nwg, W, ms,
                   tw * 1e3, mfmas * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * 4.0 * W));
        }
    }
    return 0;
}

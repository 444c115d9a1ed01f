// Probe: time per 128x128xK fp64 MFMA tile of GemmNT::run (1 tile, 256 tiles, 512 tiles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../boss.jl_amd/csrc/gemm_f64.hpp"
using namespace boss;

template <class G, bool STORE>
__global__ __launch_bounds__(256) void tile_kernel(const double* A, const double* B, double* C, int ld, int K,
                                                   unsigned long long* clk) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / G::WC, wc = wave % G::WC;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    v4d acc[G::TM][G::TN];
#pragma unroll
    for (int m = 0; m < G::TM; ++m)
#pragma unroll
        for (int n = 0; n < G::TN; ++n) acc[m][n] = v4d{0, 0, 0, 0};
    const double* a = A + (size_t)(blockIdx.x % 16) * G::BM;
    const double* b = B + (size_t)(blockIdx.x / 16 % 16) * G::BN;
    G::run(a, ld, b, ld, K, acc, lds);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double* c = C + (size_t)blockIdx.x * G::BM * G::BN;
#pragma unroll
    for (int m = 0; m < G::TM; ++m)
#pragma unroll
        for (int n = 0; n < G::TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                size_t off = (size_t)G::row_of(wr, m, lane) + (size_t)G::col_of(wc, n, i, lane) * G::BM;
                if (STORE) c[off] -= acc[m][n][i];
                else if (acc[m][n][i] == 1234.5) c[off] = 1;
            }
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; }
}

template <class G, bool STORE>
void bench(const char* name, int K) {
    const int ld = 4224;
    double *A, *B, *C; unsigned long long* clk;
    hipMalloc(&A, sizeof(double) * ld * 2048); hipMalloc(&B, sizeof(double) * ld * 2048);
    hipMalloc(&C, sizeof(double) * 1024 * G::BM * G::BN); hipMalloc(&clk, 16);
    hipMemset(A, 0, sizeof(double) * ld * 2048); hipMemset(B, 0, sizeof(double) * ld * 2048);
    hipMemset(C, 0, sizeof(double) * 1024 * G::BM * G::BN);
    hipFuncSetAttribute((const void*)tile_kernel<G, STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_DOUBLES * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int tiles : {1, 256, 512, 1024}) {
        hipLaunchKernelGGL((tile_kernel<G, STORE>), dim3(tiles), dim3(256), G::LDS_DOUBLES * 8, 0, A, B, C, ld, K, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r)
            hipLaunchKernelGGL((tile_kernel<G, STORE>), dim3(tiles), dim3(256), G::LDS_DOUBLES * 8, 0, A, B, C, ld, K, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double fl = 2.0 * G::BM * G::BN * K * tiles;
        printf("%-28s K=%4d tiles=%4d  %8.2f us  %6.2f TF  run=%llu cyc epi=%llu cyc\n", name, K, tiles, ms * 1e3, fl / (ms * 1e-3) / 1e12, h[0], h[1]);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(clk);
}

int main() {
    bench<GemmNT<2, 2, 4, 4>, true>("128x128 store", 128);
    bench<GemmNT<2, 2, 4, 4>, false>("128x128 nostore", 128);
    bench<GemmNT<2, 2, 4, 4>, false>("128x128 nostore", 1024);
    bench<GemmNT<2, 2, 2, 2>, true>("64x64 store", 128);
    bench<GemmNT<2, 2, 2, 2>, false>("64x64 nostore", 1024);
    bench<GemmNT<4, 1, 2, 2>, false>("128x32 nostore", 1024);
    return 0;
}

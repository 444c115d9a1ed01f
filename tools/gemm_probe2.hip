// Probe 2: LDS-staged GemmNT vs direct-from-global fragment streaming (register prefetch ring, no LDS/barriers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../boss.jl_amd/csrc/gemm_f64.hpp"
using namespace boss;

template <int WR, int WC, int TM, int TN, int D>
struct GemmDirect {
    static constexpr int BM = WR * TM * 16, BN = WC * TN * 16;
    __device__ static __forceinline__ void run(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                               int K, v4d (&acc)[TM][TN]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wr = wave / WC, wc = wave % WC;
        const double* Ap = A + wr * (TM * 16) + (lane & 15) + (size_t)(lane >> 4) * lda;
        const double* Bp = B + wc * (TN * 16) + (lane & 15) + (size_t)(lane >> 4) * ldb;
        const int n = K / 4;
        double af[D][TM], bf[D][TN];
#pragma unroll
        for (int s = 0; s < D; ++s) {
            int ks = s < n ? s : n - 1;
#pragma unroll
            for (int m = 0; m < TM; ++m) af[s][m] = Ap[(size_t)(4 * ks) * lda + m * 16];
#pragma unroll
            for (int t = 0; t < TN; ++t) bf[s][t] = Bp[(size_t)(4 * ks) * ldb + t * 16];
        }
        for (int k0 = 0; k0 < n; k0 += D) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                if (k0 + s < n) {
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int t = 0; t < TN; ++t) acc[m][t] = mfma_f64(bf[s][t], af[s][m], acc[m][t]);
                }
                int ks = k0 + s + D;
                ks = ks < n ? ks : n - 1;
#pragma unroll
                for (int m = 0; m < TM; ++m) af[s][m] = Ap[(size_t)(4 * ks) * lda + m * 16];
#pragma unroll
                for (int t = 0; t < TN; ++t) bf[s][t] = Bp[(size_t)(4 * ks) * ldb + t * 16];
            }
        }
    }
};

template <class G, bool STORE>
__global__ __launch_bounds__(256) void tile_kernel(const double* A, const double* B, double* C, int ld, int K, unsigned long long* clk) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / G::WC_, wc = wave % G::WC_;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    v4d acc[G::TM_][G::TN_];
#pragma unroll
    for (int m = 0; m < G::TM_; ++m)
#pragma unroll
        for (int n = 0; n < G::TN_; ++n) acc[m][n] = v4d{0, 0, 0, 0};
    const double* a = A + (size_t)(blockIdx.x % 16) * G::BM;
    const double* b = B + (size_t)(blockIdx.x / 16 % 16) * G::BN;
    G::run(a, ld, b, ld, K, acc);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double* c = C + (size_t)blockIdx.x * G::BM * G::BN;
#pragma unroll
    for (int m = 0; m < G::TM_; ++m)
#pragma unroll
        for (int n = 0; n < G::TN_; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                size_t off = (size_t)(wr * G::TM_ * 16 + m * 16 + (lane & 15)) + (size_t)(wc * G::TN_ * 16 + n * 16 + (lane >> 4) + 4 * i) * G::BM;
                if (STORE) c[off] -= acc[m][n][i];
                else if (acc[m][n][i] == 1234.5) c[off] = 1;
            }
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; }
}

template <int WR, int WC, int TM, int TN, int D>
struct GD : GemmDirect<WR, WC, TM, TN, D> { static constexpr int WC_ = WC, TM_ = TM, TN_ = TN; };

template <class G, bool STORE>
void bench(const char* name, int K) {
    const int ld = 4224;
    double *A, *B, *C; unsigned long long* clk;
    hipMalloc(&A, sizeof(double) * ld * 4096); hipMalloc(&B, sizeof(double) * ld * 4096);
    hipMalloc(&C, sizeof(double) * 1024 * G::BM * G::BN); hipMalloc(&clk, 16);
    hipMemset(A, 0, sizeof(double) * ld * 4096); hipMemset(B, 0, sizeof(double) * ld * 4096);
    hipMemset(C, 0, sizeof(double) * 1024 * G::BM * G::BN);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int tiles : {1, 256, 512, 1024}) {
        hipLaunchKernelGGL((tile_kernel<G, STORE>), dim3(tiles), dim3(256), 0, 0, A, B, C, ld, K, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((tile_kernel<G, STORE>), dim3(tiles), dim3(256), 0, 0, A, B, C, ld, K, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double fl = 2.0 * G::BM * G::BN * K * tiles;
        printf("%-30s K=%4d tiles=%4d  %8.2f us  %6.2f TF  run=%llu cyc epi=%llu cyc\n", name, K, tiles, ms * 1e3, fl / (ms * 1e-3) / 1e12, h[0], h[1]);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(clk);
}

int main() {
    bench<GD<2, 2, 4, 4, 2>, true>("direct 128x128 D=2 store", 128);
    bench<GD<2, 2, 4, 4, 4>, true>("direct 128x128 D=4 store", 128);
    bench<GD<2, 2, 4, 4, 4>, false>("direct 128x128 D=4", 1024);
    bench<GD<4, 1, 2, 2, 4>, false>("direct 128x32 D=4", 1024);
    bench<GD<4, 1, 2, 2, 8>, false>("direct 128x32 D=8", 1024);
    bench<GD<4, 1, 2, 2, 8>, false>("direct 128x32 D=8", 2048);
    bench<GD<2, 2, 2, 2, 8>, true>("direct 64x64 D=8 store", 128);
    bench<GD<2, 2, 2, 2, 8>, false>("direct 64x64 D=8", 1024);
    return 0;
}

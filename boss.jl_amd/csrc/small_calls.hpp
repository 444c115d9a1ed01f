// small_calls.hpp — one to four candidates on the resident inverse factor in ONE launch.
//
// The reference evaluates its acquisition one candidate per call (`acq.(eachcol(xs))`, expected_improvement.jl:75,79).  With
// U = L⁻ᵀ resident such a call is one pass over U's upper triangle (winv_gemv_kernel, predict_kernels.hpp); here the two
// small kernels around that pass are folded into it: every workgroup evaluates the rows of K* it needs itself (the scaled
// candidates arrive in the kernel arguments), and the workgroup that finishes last reduces the per-workgroup partials in the
// same fixed order as winv_finish_host_kernel and writes μ, σ² and the sequence number into mapped host memory.  Two kernel
// boundaries and two launches less per call (one, for a single candidate: see kst).
#pragma once
#include "potrf.hpp"

namespace boss {

// The same pass as the first half of a rank-one append (boss_gp_append with one observation on resident inverse factors,
// host_append.inc): v = L⁻¹k is ALSO the new factor row (vout), and the last workgroup's epilogue is append_scalars_kernel's —
// d = sqrt(k(x,x) + σ² − vᵀv), z_new, logdet and zᵀz advanced, {logdet, zᵀz, info} and the sequence number to mapped host memory:
// the host returns while the second pass (L⁻ᵀ row, patches) is still running behind it on the stream.
struct AppendTail {
    double y, mean;                                          // the new observation and its prior mean
    int N0;
    const double* hyp;                                       // {α², σ²}
    double *scal, *dz, *vout;
    int* info;
};
// done: a device counter that grows by gridDim.x per call (the host passes the value it has after this call)
template <int MC, bool APPEND = false>
__global__ __launch_bounds__(256) void winv_args_kernel(FewCand par, const double* __restrict__ Xsc, int N, int d, int kern, double amp2,
                                                        const double* __restrict__ U, int ldu, int Np,
                                                        const double* __restrict__ Afac, int ld, const double* __restrict__ kst,
                                                        double* __restrict__ part,
                                                        unsigned long long* __restrict__ done, unsigned long long done_after,
                                                        double* __restrict__ host_out, unsigned long long seq, AppendTail tail = AppendTail()) {
    extern __shared__ double ks[];                           // K* [c][MC]; afterwards the reduction scratch (>= 8*256 + 64 doubles)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kb = (gridDim.x - 1 - blockIdx.x) * WINV_ROWS;  // longest rows first
    // K* for the rows this workgroup's dot products reach (c <= kb + WINV_ROWS - 1): eight rows per thread and trip, all their
    // coordinate loads in flight together (one row at a time left the longest workgroup 16 dependent L2 round trips behind)
    // kst (or null): K* [c][MC] already evaluated by kstar_args_kernel (the host passes it for ONE candidate only) — for ONE candidate that launch (one row per thread across
    // the chip) costs less than the 16 evaluations per thread the longest workgroup would do here; for two and four it does not
    constexpr int KU = 8;
    const int cend = kb + WINV_ROWS;
    if (kst) {
        // nothing to stage: the walk below reads K* from global memory (32 KB, L2-resident) beside the two rows of U — no staging loop,
        // no barrier in front of the first load of U (the staged form of the same pass: 22 against 14 µs)
    } else
    for (int c0 = tid; c0 < cend; c0 += 256 * KU) {
        double r2[KU][MC];
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int j = 0; j < MC; ++j) r2[u][j] = 0.0;
        for (int kd = 0; kd < d; ++kd) {
            double xr[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) xr[u] = Xsc[(size_t)kd * Np + min(c0 + 256 * u, Np - 1)];
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int j = 0; j < MC; ++j) {
                    const double df = xr[u] - par.x[j][kd];
                    r2[u][j] = __builtin_fma(df, df, r2[u][j]);
                }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int c = c0 + 256 * u;
            if (c < cend)
#pragma unroll
                for (int j = 0; j < MC; ++j) ks[c * MC + j] = (c < N && j < par.ncols) ? amp2 * kappa_r2(kern, r2[u][j]) : 0.0;
        }
    }
    __syncthreads();
    const int k0 = kb + 2 * wave;                            // this wave's two rows, walked together (as winv_gemv_kernel)
    const double* col0 = U + (size_t)k0 * ldu;
    const double* col1 = col0 + ldu;
    double a0[MC], a1[MC];
#pragma unroll
    for (int j = 0; j < MC; ++j) a0[j] = a1[j] = 0.0;
    if (kst) winv_walk<MC>(col0, col1, kst, k0, lane, a0, a1);
    else winv_walk<MC>(col0, col1, ks, k0, lane, a0, a1);
    const double z0 = Afac[(size_t)k0 * ld + Np], z1 = Afac[(size_t)(k0 + 1) * ld + Np];
    double ss[MC], mz[MC];
#pragma unroll
    for (int j = 0; j < MC; ++j) {
        double v0 = a0[j], v1 = a1[j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            v0 += __shfl_xor(v0, off);
            v1 += __shfl_xor(v1, off);
        }
        ss[j] = __builtin_fma(v0, v0, v1 * v1);
        mz[j] = __builtin_fma(v0, z0, v1 * z1);
        if (APPEND && j == 0 && lane == 0) {
            tail.vout[k0] = v0;
            tail.vout[k0 + 1] = v1;
        }
    }
    __syncthreads();                                         // ks no longer needed: its head takes the wave partials
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < WINV_MAX_M; ++j) {
            ks[wave * 8 + 2 * j] = (j < MC) ? ss[j < MC ? j : 0] : 0.0;
            ks[wave * 8 + 2 * j + 1] = (j < MC) ? mz[j < MC ? j : 0] : 0.0;
        }
    }
    __syncthreads();
    // the workgroup's eight partials leave write-through; the counter says how many workgroups' partials are in memory
    if (tid < 8) st_sc1(part + (size_t)blockIdx.x * 8 + tid, ks[tid] + ks[8 + tid] + ks[16 + tid] + ks[24 + tid]);
    drain_stores();
    __syncthreads();
    int* lastflag = reinterpret_cast<int*>(ks + 40);
    if (tid == 0) {
        const unsigned long long before = __hip_atomic_fetch_add(as_global(done), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lastflag = (before + 1 == done_after) ? 1 : 0;
    }
    __syncthreads();
    if (*lastflag == 0) return;
    __syncthreads();
    // ---- last workgroup: the reduction of winv_finish_host_kernel (same order: deterministic, equal to the three-launch path)
    const int nwg = gridDim.x, M = par.ncols;
    double* red = ks + 64;                                   // [8][256]
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    for (int w = tid; w < nwg; w += 256)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += ld_sc1(part + (size_t)w * 8 + q);
#pragma unroll
    for (int q = 0; q < 8; ++q) red[q * 256 + tid] = acc[q];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off)
#pragma unroll
            for (int q = 0; q < 8; ++q) red[q * 256 + tid] += red[q * 256 + tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        if constexpr (APPEND) {
            // append_scalars_kernel's arithmetic, on the same sums
            const double d2 = tail.hyp[0] + tail.hyp[1] - red[0];
            const int inf = (d2 > 0.0) ? 0 : tail.N0 + 1;    // a non-positive d² is reported like a failed pivot
            *tail.info = inf;
            const double dd = sqrt(d2), zn = (tail.y - tail.mean - red[256]) / dd;
            tail.dz[0] = dd;
            tail.dz[1] = zn;
            tail.scal[0] += 2.0 * log(dd);
            tail.scal[1] = __builtin_fma(zn, zn, tail.scal[1]);
            host_out[0] = tail.scal[0];
            host_out[1] = tail.scal[1];
            reinterpret_cast<int*>(host_out + 2)[0] = inf;
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + 3), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        long long bad = -1;
        for (int j = 0; j < M; ++j) {
            const double s = red[(2 * j) * 256], z = red[(2 * j + 1) * 256];
            double v = amp2 - s + PREDICT_JITTER;
            if (!(v >= 0.0)) {                                // (NaN counts as offending, as in clip_var_kernel)
                if (v >= -MAX_NEG_VAR) v = 0.0;
                else if (bad < 0) bad = j;
            }
            host_out[j] = par.mean[j] + z;
            host_out[4 + j] = v;
        }
        reinterpret_cast<long long*>(host_out)[8] = bad;
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out) + 9, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace boss

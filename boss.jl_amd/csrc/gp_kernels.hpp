// gp_kernels.hpp — Gram build, fused prediction (K* tile → blocked trsm → Σv², v·z), EI epilogue, arg-max.
#pragma once
#include "gemm_f64.hpp"

namespace boss {

// ------------------------------------------------------------------------------------------
// Layout helpers.  Points are stored dimension-major, point-contiguous: P[k*ldp + j] is
// coordinate k of point j (so that 16 consecutive lanes read 128 contiguous bytes).
// ------------------------------------------------------------------------------------------

// Xsc[b][k][j] = Xraw[k][j] * invlam[b][k]      (ARDTransform(1 ./ λ), gaussian_process.jl:243)
__global__ void scale_points_kernel(const double* __restrict__ Xraw, double* __restrict__ Xsc, size_t xs_bstride,
                                    const double* __restrict__ invlam, int d, int ldp) {
    const int b = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ldp) return;
    for (int k = 0; k < d; ++k) Xsc[(size_t)b * xs_bstride + (size_t)k * ldp + j] = Xraw[(size_t)k * ldp + j] * invlam[b * d + k];
}

// RHS row block: row Np = (y - m)^T for j < N, everything else in rows Np..Np+31 zero.
// col0: first column to (re)write — 0 for a full fit, the first column of the re-factorised block
// row for boss_gp_append (the z entries of the columns before it are final and must survive).
__global__ void rhs_rows_kernel(double* __restrict__ Abase, int ld, size_t bstride, int N, int Np,
                                const double* __restrict__ y, const double* __restrict__ mean, size_t mean_bstride,
                                int col0) {
    const int b = blockIdx.z;
    const int j = col0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Np) return;
    double* col = Abase + (size_t)b * bstride + (size_t)j * ld + Np;
    double m = mean ? mean[(size_t)b * mean_bstride + j] : 0.0;
    col[0] = (j < N) ? (y[j] - m) : 0.0;
#pragma unroll 1
    for (int r = 1; r < 32; ++r) col[r] = 0.0;
}

// K1 gram_build: lower 64×64 tiles of K = α² κ(r) + σ² I  (padding rows/cols = identity).
// hyp[b] = {α², σ²}.
__global__ __launch_bounds__(256) void gram_kernel(const double* __restrict__ Xsc, size_t xs_bstride, int d, int N,
                                                   int Np, int kern, const double* __restrict__ hyp,
                                                   double* __restrict__ Abase, int ld, size_t bstride, int tile0) {
    // tile0: first tile of the row-major enumeration of the lower 64×64 tile triangle (0 = whole
    // matrix; T(2kb) = kb(2kb+1) starts block row kb, which boss_gp_append rebuilds alone)
    __shared__ double xj[16][64];
    const int b = blockIdx.z, tid = threadIdx.x;
    const int t = tile0 + blockIdx.x;
    int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    const double* X = Xsc + (size_t)b * xs_bstride;
    const double amp2 = hyp[2 * b], noise2 = hyp[2 * b + 1];
    const int r = tid & 63, cg = tid >> 6;
    const int i = bi * 64 + r;
    double r2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) r2[c] = 0.0;
    for (int k0 = 0; k0 < d; k0 += 16) {
        const int kc = (d - k0 < 16) ? (d - k0) : 16;
        __syncthreads();
        for (int idx = tid; idx < kc * 64; idx += 256) xj[idx >> 6][idx & 63] = X[(size_t)(k0 + (idx >> 6)) * Np + bj * 64 + (idx & 63)];
        __syncthreads();
        for (int kk = 0; kk < kc; ++kk) {
            const double xi = X[(size_t)(k0 + kk) * Np + i];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double diff = xi - xj[kk][cg * 16 + c];
                r2[c] = __builtin_fma(diff, diff, r2[c]);
            }
        }
    }
    double* A = Abase + (size_t)b * bstride;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = bj * 64 + cg * 16 + c;
        if (i < j) continue;
        double v;
        if (i < N && j < N) v = amp2 * kappa_r2(kern, r2[c]) + ((i == j) ? noise2 : 0.0);
        else v = (i == j) ? 1.0 : 0.0;
        A[(size_t)j * ld + i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// GradientGaussianProcess (src/models/gradient_gp.jl): n points observed as values AND
// gradients give an n(1+d) square system over the ordering [f(x_1..n), ∂_1 f(x_1..n), …, ∂_d f(x_1..n)].
// Observation a = l·n + i is (l, point i): l = 0 the value, l ≥ 1 the derivative along coordinate l−1.
// Only the Gram build and the cross-covariances differ from the plain model; the factorisation, the
// substitutions and the acquisition kernels run unchanged on the larger matrix.
// Points are RAW here (P[k*ldx + j]); il = 1/(λ+1e-8).
// ------------------------------------------------------------------------------------------
constexpr int AUG_MAX_D = 16;
constexpr double ISAPPROX_RTOL2 = 2.220446049250313e-16;     // Julia `≈` on Float64 vectors: rtol = √eps, squared

// One entry of `_build_augmented_kernel` (gradient_gp.jl:175-199) / `_build_cross_cov` (:221-243):
// row observation (lr, xi), column observation (lc, xj).  The value block uses the points as given; the
// derivative blocks are evaluated at (xi, xj + 1e-8) when xi ≈ xj (:148-152, :233).
__device__ __forceinline__ double aug_entry(int kern, double amp2, int d, const double* il, const double* xi, int si,
                                            const double* xj, int sj, int lr, int lc) {
    double du2 = 0.0, ni = 0.0, nj = 0.0, r2 = 0.0;
    for (int k = 0; k < d; ++k) {
        const double a = xi[k * si], b = xj[k * sj], u = a - b, t = u * il[k];
        du2 = __builtin_fma(u, u, du2);
        ni = __builtin_fma(a, a, ni);
        nj = __builtin_fma(b, b, nj);
        r2 = __builtin_fma(t, t, r2);
    }
    if (lr == 0 && lc == 0) return amp2 * kappa_r2(kern, r2);
    double eps = 0.0;
    if (du2 <= ISAPPROX_RTOL2 * fmax(ni, nj)) {
        eps = MIN_PARAM_VALUE;
        r2 = 0.0;
        for (int k = 0; k < d; ++k) {
            const double t = (xi[k * si] - (xj[k * sj] + eps)) * il[k];
            r2 = __builtin_fma(t, t, r2);
        }
    }
    const double h = kappa_prime_over_r_r2(kern, r2);
    if (lr == 0 || lc == 0) {
        const int m = (lr == 0 ? lc : lr) - 1;
        const double s = (xi[m * si] - (xj[m * sj] + eps)) * il[m] * il[m];
        return (lr == 0) ? -amp2 * h * s : amp2 * h * s;     // ∂k/∂(xj)_m  |  ∂k/∂(xi)_l
    }
    const int l = lr - 1, m = lc - 1;
    const double sl = (xi[l * si] - (xj[l * sj] + eps)) * il[l] * il[l];
    const double sm = (xi[m * si] - (xj[m * sj] + eps)) * il[m] * il[m];
    double v = kappa_second_r2(kern, r2) * sl * sm;
    if (l == m) v = __builtin_fma(h, il[l] * il[l], v);
    return -amp2 * v;
}

// Lower 64×64 tiles of the augmented matrix.  `cholesky(Symmetric(K))` (:209,:325) reads the UPPER
// triangle, so the stored entry (a, b), a ≥ b, is the reference's K[b, a].  hyp = {α², σ², σ_∂²};
// padding rows/columns = identity.
__global__ __launch_bounds__(256) void aug_gram_kernel(const double* __restrict__ Xraw, int ldx, int d, int n, int N, int Np,
                                                       int kern, const double* __restrict__ hyp,
                                                       const double* __restrict__ invlam, double* __restrict__ A, int ld) {
    __shared__ double xa[AUG_MAX_D][64], xb[AUG_MAX_D][64], il[AUG_MAX_D];
    __shared__ int la[64], lb[64];
    const int tid = threadIdx.x, t = blockIdx.x;
    int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    if (tid < 128) {
        const int c = tid & 63, obs = (tid < 64 ? bi : bj) * 64 + c;
        const int l = obs < N ? obs / n : -1, pt = obs < N ? obs - l * n : 0;
        if (tid < 64) la[c] = l; else lb[c] = l;
        for (int k = 0; k < d; ++k) (tid < 64 ? xa : xb)[k][c] = Xraw[(size_t)k * ldx + pt];
    }
    if (tid >= 128 && tid < 128 + d) il[tid - 128] = invlam[tid - 128];
    __syncthreads();
    const double amp2 = hyp[0];
    const int r = tid & 63, cg = tid >> 6, a = bi * 64 + r;
#pragma unroll 1
    for (int c = 0; c < 16; ++c) {
        const int cc = cg * 16 + c, b = bj * 64 + cc;
        if (a < b) continue;
        double v;
        if (a < N && b < N) {
            v = aug_entry(kern, amp2, d, il, &xb[0][cc], 64, &xa[0][r], 64, lb[cc], la[r]);
            if (a == b) v += (la[r] == 0) ? hyp[1] : hyp[2];
        } else {
            v = (a == b) ? 1.0 : 0.0;
        }
        A[(size_t)b * ld + a] = v;
    }
}

// Cross-covariances of `_build_cross_cov` for every candidate, written where the substitution kernels
// expect their right-hand side: out[tile][row][BN] (the V slabs of predict_kernel<G, true>, or the residual
// array of the few-candidates path).  One training observation per thread; padding rows = 0.
__global__ __launch_bounds__(256) void aug_kstar_kernel(const double* __restrict__ Xraw, int ldx, int d, int n, int N, int Np,
                                                        const double* __restrict__ Craw, int Mp, int kern, double amp2,
                                                        const double* __restrict__ invlam, double* __restrict__ out, int BN) {
    extern __shared__ double sm[];                           // cs[d][BN] | xt[d][256] | il[d]
    double* cs = sm;
    double* xt = cs + d * BN;
    double* il = xt + d * 256;
    const int tid = threadIdx.x, c0 = blockIdx.y * BN;
    out += (size_t)blockIdx.y * Np * BN;
    const int row = blockIdx.x * 256 + tid;
    const int l = row < N ? row / n : -1, pt = row < N ? row - l * n : 0;
    for (int idx = tid; idx < d * BN; idx += 256) cs[idx] = Craw[(size_t)(idx / BN) * Mp + c0 + (idx % BN)];
    for (int k = 0; k < d; ++k) xt[k * 256 + tid] = Xraw[(size_t)k * ldx + pt];
    if (tid < d) il[tid] = invlam[tid];
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < BN; ++c)
        out[(size_t)row * BN + c] = (l >= 0) ? aug_entry(kern, amp2, d, il, cs + c, BN, xt + tid, 256, 0, l) : 0.0;
}

// ------------------------------------------------------------------------------------------
// NonstationaryGP (src/models/nonstationary_gp/nonstationary_gp.jl:61-107): the Gibbs kernel
//   k(x, y) = ((α(x) + α(y))/2)² Π_i sqrt(2 λ_i(x) λ_i(y) / (λ_i(x)² + λ_i(y)²)) exp(−(x_i − y_i)² / (λ_i(x)² + λ_i(y)²))
// with per-point noise σ(x)² on the diagonal (finite_nongp, :183-196).  λ(·), α(·), σ(·) are the caller's
// latent models evaluated at the training points / candidates; they cross the ABI as arrays.
// Points raw (rounded where discrete), P[k*ldp + j]; Lam likewise.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void gibbs_dim(double x, double lx, double y, double ly, double& prod, double& esum) {
    const double q = rcp_refined(__builtin_fma(lx, lx, ly * ly));
    const double df = x - y;
    prod *= 2.0 * lx * ly * q;
    esum = __builtin_fma(df * df, q, esum);
}

__global__ __launch_bounds__(256) void gibbs_gram_kernel(const double* __restrict__ X, const double* __restrict__ Lam,
                                                         const double* __restrict__ amp, const double* __restrict__ noise,
                                                         int d, int N, int Np, double* __restrict__ A, int ld) {
    __shared__ double xj[16][64], lj[16][64];
    const int tid = threadIdx.x, t = blockIdx.x;
    int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    const int r = tid & 63, cg = tid >> 6;
    const int i = bi * 64 + r;
    double pr[16], es[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        pr[c] = 1.0;
        es[c] = 0.0;
    }
    for (int k0 = 0; k0 < d; k0 += 16) {
        const int kc = (d - k0 < 16) ? (d - k0) : 16;
        __syncthreads();
        for (int idx = tid; idx < kc * 64; idx += 256) {
            xj[idx >> 6][idx & 63] = X[(size_t)(k0 + (idx >> 6)) * Np + bj * 64 + (idx & 63)];
            lj[idx >> 6][idx & 63] = Lam[(size_t)(k0 + (idx >> 6)) * Np + bj * 64 + (idx & 63)];
        }
        __syncthreads();
        for (int kk = 0; kk < kc; ++kk) {
            const double xi = X[(size_t)(k0 + kk) * Np + i], li = Lam[(size_t)(k0 + kk) * Np + i];
#pragma unroll
            for (int c = 0; c < 16; ++c) gibbs_dim(xi, li, xj[kk][cg * 16 + c], lj[kk][cg * 16 + c], pr[c], es[c]);
        }
    }
    const double ai = amp[i];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = bj * 64 + cg * 16 + c;
        if (i < j) continue;
        double v;
        if (i < N && j < N) {
            const double am = 0.5 * (ai + amp[j]);
            v = am * am * sqrt(pr[c]) * exp(-es[c]);
            if (i == j) v = __builtin_fma(noise[i], noise[i], v);
        } else {
            v = (i == j) ? 1.0 : 0.0;
        }
        A[(size_t)j * ld + i] = v;
    }
}

// K* of the Gibbs kernel, written where the substitution kernels take their right-hand side
// (out[tile][row][BN], see aug_kstar_kernel).  One training point per thread.
template <int BN>
__global__ __launch_bounds__(256) void gibbs_kstar_kernel(const double* __restrict__ X, const double* __restrict__ Lam,
                                                          const double* __restrict__ amp, int d, int N, int Np,
                                                          const double* __restrict__ C, const double* __restrict__ Clam,
                                                          const double* __restrict__ Camp, int Mp, double* __restrict__ out) {
    extern __shared__ double sm[];                           // cx[d][BN] | cl[d][BN] | ca[BN]
    double* cx = sm;
    double* cl = cx + d * BN;
    double* ca = cl + d * BN;
    const int tid = threadIdx.x, c0 = blockIdx.y * BN;
    out += (size_t)blockIdx.y * Np * BN;
    for (int idx = tid; idx < d * BN; idx += 256) {
        cx[idx] = C[(size_t)(idx / BN) * Mp + c0 + (idx % BN)];
        cl[idx] = Clam[(size_t)(idx / BN) * Mp + c0 + (idx % BN)];
    }
    if (tid < BN) ca[tid] = Camp[c0 + tid];
    __syncthreads();
    const int row = blockIdx.x * 256 + tid;
    double pr[BN], es[BN];
#pragma unroll
    for (int c = 0; c < BN; ++c) {
        pr[c] = 1.0;
        es[c] = 0.0;
    }
    for (int k = 0; k < d; ++k) {
        const double xr = X[(size_t)k * Np + row], lr = Lam[(size_t)k * Np + row];
#pragma unroll
        for (int c = 0; c < BN; ++c) gibbs_dim(cx[k * BN + c], cl[k * BN + c], xr, lr, pr[c], es[c]);
    }
    const double ar = amp[row];
    const bool live = row < N;
#pragma unroll
    for (int c = 0; c < BN; ++c) {
        const double am = 0.5 * (ar + ca[c]);
        out[(size_t)row * BN + c] = live ? am * am * sqrt(pr[c]) * exp(-es[c]) : 0.0;
    }
}

// σ²(x*) = k(x*,x*) − Σv² + 1e-18 with k(x*,x*) = α(x*)²; the substitution kernels left −Σv² in var.
__global__ void gibbs_var_kernel(double* __restrict__ var, const double* __restrict__ Camp, int M) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < M) var[j] = __builtin_fma(Camp[j], Camp[j], var[j]) + PREDICT_JITTER;
}

// ------------------------------------------------------------------------------------------
// K4-K7 fused prediction.  One workgroup owns BN candidates and walks the row blocks of L:
//     R_i = K*_i − Σ_{j<i} L_ij V_j          (MFMA GEMM, V_j re-read from its own scratch slab)
//     V_i = Dinv_i · R_i                      (MFMA GEMM, R_i resident in LDS)
//     ss += colsum(V_i²) ,  mz += V_i^T z_i
// which is the blocked form of  V = C.U' \ K*  (AbstractGPs var(post(X*))), with
// μ − m(X*) = K*^T a = V^T z  accumulated in the same pass.  K* never touches HBM.
// ------------------------------------------------------------------------------------------
template <class G>
struct PredictLds {
    static constexpr int LDR = G::BN + 16;
    static constexpr int PART = 2 * G::TN * 4;       // per-thread Σv², v·z partials, parked in LDS between row blocks
    static constexpr int BYTES = (G::BM * LDR + 2 * G::WR * G::BN + G::NTHREADS * PART) * 8;
};

// G = GemmDirect<WR,1,TM,TN,D> with RB = WR·TM·16 ∈ {128, 256}: WR waves stacked along the RB rows of a
// substitution step, BN = 16·TN candidates; Dinv holds the dense inverses of the RB×RB diagonal blocks.
// Both GEMMs stream their A operand (L row block / Dinv_i) straight from L2 through a register
// ring; GEMM1's B operand is the workgroup's own V slab (global, candidate-contiguous), GEMM2's
// B operand is the R tile in LDS.  Three barriers per row block, none inside the GEMMs.
// PRE: the right-hand side K* is not evaluated here but was written to the workgroup's V slab beforehand
// (gradient-observation posteriors, aug_kstar_kernel); block ib's rows are consumed before V_ib overwrites them.
template <class G, bool PRE = false>
__global__ __launch_bounds__(G::NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void predict_kernel(const double* __restrict__ A, int ld, int Np, int N,
                                                      const double* __restrict__ Dinv,
                                                      const double* __restrict__ Xsc,
                                                      const double* __restrict__ Csc, int d, int Mp, int kern,
                                                      double amp2, double* __restrict__ Vscratch,
                                                      const double* __restrict__ mean_s, int M,
                                                      double* __restrict__ mu_out, double* __restrict__ var_out, int dbg) {
    static_assert(G::WC == 1 && (G::BM == BLK || G::BM == 2 * BLK), "waves stacked along a 128- or 256-row block");
    constexpr int RB = G::BM;                          // rows per substitution step; Dinv holds RB×RB inverses
    extern __shared__ double lds[];
    constexpr int BN = G::BN, TM = G::TM, TN = G::TN, LDR = PredictLds<G>::LDR;
    double* Rs = lds;
    double* red = Rs + RB * LDR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar wave index (see gemm_f64.hpp)
    const int wr = wave, wc = 0;
    const int c0 = blockIdx.x * BN;
    double* V = Vscratch + (size_t)blockIdx.x * Np * BN;
    const int nblk = Np / RB;

    // per-thread partial sums live in LDS between row blocks (slot-major [slot][tid]: conflict-free).
    // In registers they push the 64-candidate instantiation over 256 VGPRs; hipcc then spills to AGPRs
    // and copies the inline-asm prefetch ring's registers BEFORE their loads have landed.
    double* part = red + 2 * G::WR * G::BN;
#pragma unroll
    for (int u = 0; u < 2 * TN * 4; ++u) part[u * G::NTHREADS + tid] = 0.0;

    for (int ib = 0; ib < nblk; ++ib) {
        v4d acc[TM][TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
        if (!(dbg & 4)) G::template run<1>(A + (size_t)ib * RB, ld, V, BN, ib * RB, acc);

        // K*_ib tile in the accumulator layout
        double r2[TM][TN][4];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) r2[m][n][i] = 0.0;
        for (int kd = 0; kd < ((dbg & 1) || PRE ? 0 : d); ++kd) {
            double xr[TM], xc[TN][4];
#pragma unroll
            for (int m = 0; m < TM; ++m) xr[m] = Xsc[(size_t)kd * Np + ib * RB + G::row_of(wr, m, lane)];
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) xc[n][i] = Csc[(size_t)kd * Mp + c0 + G::col_of(wc, n, i, lane)];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double diff = xr[m] - xc[n][i];
                        r2[m][n][i] = __builtin_fma(diff, diff, r2[m][n][i]);
                    }
        }
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int row = G::row_of(wr, m, lane);
            const bool live = (ib * RB + row) < N;
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double ks;
                    if constexpr (PRE) ks = V[(size_t)(ib * RB + row) * BN + G::col_of(wc, n, i, lane)];
                    else ks = (live && !(dbg & 1)) ? amp2 * kappa_r2(kern, r2[m][n][i]) : 0.0;
                    Rs[row * LDR + G::col_of(wc, n, i, lane)] = ks - acc[m][n][i];
                }
        }
        __syncthreads();                                   // R tile complete
        v4d acc2[TM][TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc2[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
        // Dinv_i is lower triangular: a row only needs the k up to its own index.  With 256-row steps the
        // waves take the 32-row groups {w, 7-w} (equal work); otherwise contiguous slices, k < 32 (w + 1).
        constexpr bool TRI = (RB == 256 && G::PM == 2 && G::WR == 4);
        if (!(dbg & 2)) {
            if constexpr (TRI) G::run_Blds_tri(Dinv + (size_t)ib * RB * RB, RB, Rs, LDR, acc2);
            else G::run_Blds(Dinv + (size_t)ib * RB * RB, RB, Rs, LDR, (TM * 16) * (wr + 1), acc2);   // K multiple of 16
        }

#pragma unroll
        for (int m = 0; m < TM; ++m) {
            int rloc;
            if constexpr (TRI) rloc = G::tri_row_of(wr, m, lane);
            else rloc = G::row_of(wr, m, lane);
            const int row = ib * RB + rloc;
            const double zr = A[(size_t)row * ld + Np];
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double v = acc2[m][n][i];
                    if (!(dbg & 8)) V[(size_t)row * BN + G::col_of(wc, n, i, lane)] = v;
                    double* ps = part + (size_t)(2 * (n * 4 + i)) * G::NTHREADS + tid;
                    ps[0] = __builtin_fma(v, v, ps[0]);
                    ps[G::NTHREADS] = __builtin_fma(v, zr, ps[G::NTHREADS]);
                }
        }
        __syncthreads();   // V_ib visible to the whole workgroup (it is the next block's B operand); Rs reusable
    }
    // reduce over the 16 row-lanes, then over the 4 waves stacked along rows
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double s = part[(size_t)(2 * (n * 4 + i)) * G::NTHREADS + tid], z = part[(size_t)(2 * (n * 4 + i) + 1) * G::NTHREADS + tid];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                s += __shfl_xor(s, off);
                z += __shfl_xor(z, off);
            }
            if ((lane & 15) == 0) {
                const int col = G::col_of(wc, n, i, lane);
                red[wr * BN + col] = s;
                red[G::WR * BN + wr * BN + col] = z;
            }
        }
    __syncthreads();
    if (tid < BN) {
        double s = 0.0, z = 0.0;
#pragma unroll
        for (int w = 0; w < G::WR; ++w) {
            s += red[w * BN + tid];
            z += red[G::WR * BN + w * BN + tid];
        }
        // μ = m(x*) + V^T z ;  σ² = k(x*,x*) − Σ V² + 1e-18   (unclipped; clipping is the consumer's job)
        const int j = c0 + tid;
        if (j < M) {
            mu_out[j] = (mean_s ? mean_s[j] : 0.0) + z;
            if constexpr (PRE) var_out[j] = (kern == KERN_GIBBS) ? -s       // per-candidate prior variance: gibbs_var_kernel
                                                                 : fmax(0.0, amp2 - s);   // gradient_gp.jl:346: no jitter, clamped at 0
            else var_out[j] = amp2 - s + PREDICT_JITTER;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Few candidates (M <= 32: the reference's own call pattern is ONE candidate per call,
// expected_improvement.jl:75,79).  The fused kernel above gives one workgroup per 32 candidates, i.e.
// ONE busy CU and ≈2.1 ms of latency at N=4096.  Here the substitution runs right-looking in 256-row
// steps spread over the chip, on a residual array R (Np × 32) that starts as K*:
//   kstar_rows_kernel   R = K* (all rows × 32 candidates), one row per thread
//   few_finish_kernel   V_i = Dinv2_i R_i, Σv², v·z (and μ, σ² at the last step)           — one workgroup
//   few_update_kernel   R_j −= L[j, i] V_i for every later row block j                       — one workgroup per 128 rows
// Two short launches per step instead of one long-running workgroup.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kstar_rows_kernel(const double* __restrict__ Xsc, int Np, int N,
                                                         const double* __restrict__ Csc, int d, int Mp, int kern,
                                                         double amp2, double* __restrict__ kst, int ncols) {
    // ncols: columns of the 32-wide tile that are needed (the one-to-four-candidates path reads only the first ones)
    extern __shared__ double cs[];                           // [d][32]
    const int c0 = blockIdx.y * 32;                          // candidate tile
    kst += (size_t)blockIdx.y * Np * 32;
    for (int idx = threadIdx.x; idx < d * 32; idx += 256) cs[idx] = Csc[(size_t)(idx >> 5) * Mp + c0 + (idx & 31)];
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    double r2[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) r2[c] = 0.0;
    for (int kd = 0; kd < d; ++kd) {
        const double xr = Xsc[(size_t)kd * Np + row];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            if (c < ncols) {
                const double df = xr - cs[kd * 32 + c];
                r2[c] = __builtin_fma(df, df, r2[c]);
            }
        }
    }
    const bool live = row < N;
#pragma unroll
    for (int c = 0; c < 32; ++c)
        if (c < ncols) kst[(size_t)row * 32 + c] = live ? amp2 * kappa_r2(kern, r2[c]) : 0.0;
}

// GU = GemmDirect<4,1,2,2,D>: 128 rows × 32 candidates per workgroup, K = 256
template <class GU>
__global__ __launch_bounds__(GU::NTHREADS) void few_update_kernel(const double* __restrict__ A, int ld, int Np, int ib,
                                                                  const double* __restrict__ V, double* __restrict__ R) {
    static_assert(GU::WC == 1 && GU::BM == BLK && GU::BN == 32, "128×32 tiles");
    constexpr int TM = GU::TM, TN = GU::TN;
    V += (size_t)blockIdx.x * Np * 32;                       // candidate tile (fastest in dispatch order: the tiles of one
                                                             // row block share its panel of L in L2)
    R += (size_t)blockIdx.x * Np * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = (ib + 1) * PRED_RB + blockIdx.y * BLK;    // first row of this workgroup's block
    double* Rb = R + (size_t)r0 * 32;
    v4d acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = GU::row_of(wave, m, lane);
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][n][i] = Rb[row * 32 + GU::col_of(0, n, i, lane)];
    }
    GU::template run<-1>(A + (size_t)r0 + (size_t)ib * PRED_RB * ld, ld, V + (size_t)ib * PRED_RB * 32, 32, PRED_RB, acc);
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = GU::row_of(wave, m, lane);
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) Rb[row * 32 + GU::col_of(0, n, i, lane)] = acc[m][n][i];
    }
}

template <class G>
__global__ __launch_bounds__(G::NTHREADS) void few_finish_kernel(const double* __restrict__ A, int ld, int Np, int ib,
                                                                 const double* __restrict__ R,
                                                                 const double* __restrict__ Dinv2, double* __restrict__ V,
                                                                 double* __restrict__ ssmz, int last,
                                                                 const double* __restrict__ mean_s, int M, double amp2,
                                                                 double* __restrict__ mu_out, double* __restrict__ var_out,
                                                                 int aug) {
    constexpr int RB = G::BM, TM = G::TM, TN = G::TN, LDR = PredictLds<G>::LDR, BN = 32;
    extern __shared__ double lds[];
    double* Rs = lds;
    double* red = Rs + RB * LDR;                             // [2][WR][BN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.x * BN;                          // candidate tile
    R += (size_t)blockIdx.x * Np * 32;
    V += (size_t)blockIdx.x * Np * 32;
    ssmz += (size_t)blockIdx.x * 64;
    const double* Rb = R + (size_t)ib * RB * 32;
#pragma unroll 8
    for (int q = 0; q < RB * 32 / 256; ++q) {                // coalesced copy of the step's residual rows into LDS
        const int e = tid + 256 * q;
        Rs[(e >> 5) * LDR + (e & 31)] = Rb[e];
    }
    __syncthreads();
    v4d acc2[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc2[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
    G::run_Blds_tri(Dinv2 + (size_t)ib * RB * RB, RB, Rs, LDR, acc2);
    double ps[TN][4], pz[TN][4];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) ps[n][i] = pz[n][i] = 0.0;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = ib * RB + G::tri_row_of(wave, m, lane);
        const double zr = A[(size_t)row * ld + Np];
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double v = acc2[m][n][i];
                V[(size_t)row * BN + G::col_of(0, n, i, lane)] = v;
                ps[n][i] = __builtin_fma(v, v, ps[n][i]);
                pz[n][i] = __builtin_fma(v, zr, pz[n][i]);
            }
    }
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double s = ps[n][i], z = pz[n][i];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                s += __shfl_xor(s, off);
                z += __shfl_xor(z, off);
            }
            if ((lane & 15) == 0) {
                const int col = G::col_of(0, n, i, lane);
                red[wave * BN + col] = s;
                red[G::WR * BN + wave * BN + col] = z;
            }
        }
    __syncthreads();
    if (tid < BN) {
        double s = ssmz[tid], z = ssmz[BN + tid];
#pragma unroll
        for (int w = 0; w < G::WR; ++w) {
            s += red[w * BN + tid];
            z += red[G::WR * BN + w * BN + tid];
        }
        ssmz[tid] = s;
        ssmz[BN + tid] = z;
        if (last && c0 + tid < M) {
            mu_out[c0 + tid] = (mean_s ? mean_s[c0 + tid] : 0.0) + z;
            var_out[c0 + tid] = aug == 2 ? -s : aug ? fmax(0.0, amp2 - s) : amp2 - s + PREDICT_JITTER;
        }
    }
}

// ------------------------------------------------------------------------------------------
// One to four candidates per call, many calls per posterior — the reference's own pattern
// (`acq.(eachcol(xs))`, expected_improvement.jl:75,79).  From the second such call on a factorisation the
// handle keeps U = L⁻ᵀ (recursive doubling, linv_level_kernel: ≈0.9 ms once) and a call is one pass over
// its upper triangle:  v_k = Σ_{c≤k} U[c,k] k*_c  — column k of U is contiguous, one wave per row k with the
// lanes along c, K* (≤ 4 columns) staged in LDS once per workgroup — 67 MB of coalesced reads at N=4096
// instead of 16 dependent substitution steps.  Per-workgroup partials of Σv², v·z are summed in a fixed order.
// ------------------------------------------------------------------------------------------
constexpr int WINV_ROWS = 8;                                  // rows k per workgroup (two per wave)
constexpr int WINV_MAX_M = 4;
template <int MC>                                            // candidates staged per call: 1, 2 or 4
__global__ __launch_bounds__(256) void winv_gemv_kernel(const double* __restrict__ U, int ldu, int Np,
                                                        const double* __restrict__ Afac, int ld,
                                                        const double* __restrict__ R, int M, double* __restrict__ part,
                                                        double* __restrict__ vout) {
    // vout (or null): v itself for candidate 0 — the new factor row of a rank-one append
    extern __shared__ double ks[];                           // K* [c][MC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < Np * MC; idx += 256) {
        const int c = idx / MC, j = idx % MC;
        ks[idx] = (j < M) ? R[(size_t)c * 32 + j] : 0.0;
    }
    __syncthreads();
    const int kb = (gridDim.x - 1 - blockIdx.x) * WINV_ROWS;  // longest rows first
    const int k0 = kb + 2 * wave;                            // this wave's two rows, walked together (K* read once for both)
    const double* col0 = U + (size_t)k0 * ldu;
    const double* col1 = col0 + ldu;
    double a0[MC], a1[MC];
#pragma unroll
    for (int j = 0; j < MC; ++j) a0[j] = a1[j] = 0.0;
    int c = lane;
    for (; c + 64 <= k0; c += 128) {                         // two 64-wide chunks per trip: four loads in flight per lane
        const double u00 = col0[c], u10 = col1[c], u01 = col0[c + 64], u11 = col1[c + 64];
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            const double q0 = ks[c * MC + j], q1 = ks[(c + 64) * MC + j];
            a0[j] = __builtin_fma(u00, q0, a0[j]);
            a1[j] = __builtin_fma(u10, q0, a1[j]);
            a0[j] = __builtin_fma(u01, q1, a0[j]);
            a1[j] = __builtin_fma(u11, q1, a1[j]);
        }
    }
    for (; c <= k0 + 1; c += 64) {                           // the ragged end (row k0 stops one entry before row k0+1)
        const double u0 = (c <= k0) ? col0[c] : 0.0, u1 = col1[c];
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            const double q = ks[c * MC + j];
            a0[j] = __builtin_fma(u0, q, a0[j]);
            a1[j] = __builtin_fma(u1, q, a1[j]);
        }
    }
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
        if (j == 0 && vout && lane == 0) {
            vout[k0] = v0;
            vout[k0 + 1] = v1;
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
    if (tid < 8) part[(size_t)blockIdx.x * 8 + tid] = ks[tid] + ks[8 + tid] + ks[16 + tid] + ks[24 + tid];
}

// More than four candidates but fewer than fill the fused kernel (≤ 4096), again repeatedly on one
// factorisation (multistart refinement: every iteration of HipGradientAM is such a call): with both inverse
// factors resident the substitutions are plain GEMMs without any sequential step,
//   inv_fwd_kernel   V = L⁻¹ K*   (A operand = the lower inverse, k range up to the row block),  Σv², v·z partials per row block
//   inv_bwd_kernel   W = L⁻ᵀ V    (A operand = the upper inverse, k range from the row block)    — the adjoint pass of the gradients
// on 128×32 tiles, one workgroup per (row block, candidate tile); the candidate tiles of one row block are neighbours in
// dispatch order, so they share that row block's panel of the inverse in L2 (row-block-fastest order streamed every panel
// from HBM once per tile: 37 TF instead of 50; 256×32 tiles were no faster).
template <class GU>
__global__ __launch_bounds__(GU::NTHREADS) void inv_fwd_kernel(const double* __restrict__ Linv, int ldl, int Np,
                                                               const double* __restrict__ Afac, int ld,
                                                               const double* __restrict__ Kst, double* __restrict__ Vslabs,
                                                               double* __restrict__ ssp) {
    static_assert(GU::WC == 1 && GU::BN == 32 && GU::WR == 4, "(128 or 256)×32 tiles, four waves stacked along the rows");
    constexpr int TM = GU::TM, TN = GU::TN, BM = GU::BM;
    __shared__ double red[2][4][32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rb = gridDim.y - 1 - blockIdx.y;               // deepest row blocks first; candidate tiles of one row block are
                                                             // neighbours in dispatch order and share its panel of the inverse in L2
    const int r0 = rb * BM;
    const double* B = Kst + (size_t)blockIdx.x * Np * 32;
    double* V = Vslabs + (size_t)blockIdx.x * Np * 32;
    v4d acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
    GU::template run<1>(Linv + r0, ldl, B, 32, r0 + BM, acc);
    double ps[TN][4], pz[TN][4];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) ps[n][i] = pz[n][i] = 0.0;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = r0 + GU::row_of(wave, m, lane);
        const double zr = Afac[(size_t)row * ld + Np];
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double v = acc[m][n][i];
                V[(size_t)row * 32 + GU::col_of(0, n, i, lane)] = v;
                ps[n][i] = __builtin_fma(v, v, ps[n][i]);
                pz[n][i] = __builtin_fma(v, zr, pz[n][i]);
            }
    }
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double sv = ps[n][i], zv = pz[n][i];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                sv += __shfl_xor(sv, off);
                zv += __shfl_xor(zv, off);
            }
            if ((lane & 15) == 0) {
                const int col = GU::col_of(0, n, i, lane);
                red[0][wave][col] = sv;
                red[1][wave][col] = zv;
            }
        }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int q = threadIdx.x >> 5, col = threadIdx.x & 31;
        ssp[((size_t)blockIdx.x * gridDim.y + rb) * 64 + threadIdx.x] = red[q][0][col] + red[q][1][col] + red[q][2][col] + red[q][3][col];
    }
}

// Σ over the row blocks (fixed order), then μ and σ² as in few_finish_kernel
__global__ __launch_bounds__(256) void inv_fwd_finish_kernel(const double* __restrict__ ssp, int nrb, const double* __restrict__ mean_s,
                                                             int M, double amp2, int mode, double* __restrict__ mu,
                                                             double* __restrict__ var) {
    __shared__ double red[4][64];
    const int tid = threadIdx.x, q = tid & 63, grp = tid >> 6;
    const double* p = ssp + (size_t)blockIdx.x * nrb * 64;
    double a = 0.0;
    for (int rb = grp; rb < nrb; rb += 4) a += p[(size_t)rb * 64 + q];
    red[grp][q] = a;
    __syncthreads();
    if (tid < 32) {
        const int j = blockIdx.x * 32 + tid;
        if (j < M) {
            const double sv = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            const double zv = (red[0][32 + tid] + red[1][32 + tid]) + (red[2][32 + tid] + red[3][32 + tid]);
            mu[j] = (mean_s ? mean_s[j] : 0.0) + zv;
            var[j] = mode == 2 ? -sv : mode == 1 ? fmax(0.0, amp2 - sv) : amp2 - sv + PREDICT_JITTER;
        }
    }
}

template <class GU>
__global__ __launch_bounds__(GU::NTHREADS) void inv_bwd_kernel(const double* __restrict__ Uinv, int ldu, int Np,
                                                               const double* __restrict__ Vslabs, double* __restrict__ Wslabs) {
    static_assert(GU::WC == 1 && GU::BN == 32, "(128 or 256)×32 tiles");
    constexpr int TM = GU::TM, TN = GU::TN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.y * GU::BM;                      // block 0 is the deepest here; tiles of one row block are neighbours
    const double* V = Vslabs + (size_t)blockIdx.x * Np * 32;
    double* W = Wslabs + (size_t)blockIdx.x * Np * 32;
    v4d acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
    GU::template run<1>(Uinv + r0 + (size_t)r0 * ldu, ldu, V + (size_t)r0 * 32, 32, Np - r0, acc);
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = r0 + GU::row_of(wave, m, lane);
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) W[(size_t)row * 32 + GU::col_of(0, n, i, lane)] = acc[m][n][i];
    }
}

// ------------------------------------------------------------------------------------------
// Rank-one append on resident inverse factors (boss_gp_append with one observation, from the second append on a set
// of hyper-parameters): with l = L⁻¹k (winv_gemv_kernel, vout), d = sqrt(k(x,x) + σ² − lᵀl), z_new = (y − m − lᵀz)/d
//     L ← [L 0; lᵀ d] ,   L⁻¹ ← [L⁻¹ 0; −wᵀ/d  1/d] ,  w = L⁻ᵀ l
// i.e. one pass over each inverse factor (2 × 67 MB at N = 4096) instead of sweeping the new block row through all
// earlier panels.  The diagonal-block inverses the other kernels use (16×16, 128×128, 256×256) are the diagonal blocks
// of L⁻¹, so the same row is patched into them.
// ------------------------------------------------------------------------------------------
// w_c = Σ_{r≥c} Linv[r, c] l_r  for c < N0: column c of the lower inverse is contiguous; one wave per two columns.
__global__ __launch_bounds__(256) void linv_col_gemv_kernel(const double* __restrict__ Linv, int ldl, int N0,
                                                            const double* __restrict__ l, double* __restrict__ w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 4 + wave) * 2;
    if (c0 >= N0) return;
    const double* col0 = Linv + (size_t)c0 * ldl;
    const double* col1 = col0 + ldl;
    const bool two = c0 + 1 < N0;
    double a0 = 0.0, a1 = 0.0;
    for (int r = c0 + lane; r < N0; r += 64) {
        const double lr = l[r];
        a0 = __builtin_fma(col0[r], lr, a0);
        if (two && r > c0) a1 = __builtin_fma(col1[r], lr, a1);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a0 += __shfl_xor(a0, off);
        a1 += __shfl_xor(a1, off);
    }
    if (lane == 0) {
        w[c0] = a0;
        if (two) w[c0 + 1] = a1;
    }
}

// Σ of the gemv partials (fixed-order tree), then d, z_new; scal = {logdet, zᵀz} is advanced, dz = {d, z_new};
// a non-positive d² is reported like a failed pivot (info = N0 + 1).
__global__ __launch_bounds__(256) void append_scalars_kernel(const double* __restrict__ part, int nwg, const double* __restrict__ hyp,
                                                             const double* __restrict__ y, const double* __restrict__ mean, int N0,
                                                             double* __restrict__ scal, double* __restrict__ dz, int* __restrict__ info) {
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    double s = 0.0, z = 0.0;
    for (int w = tid; w < nwg; w += 256) {
        s += part[(size_t)w * 8];
        z += part[(size_t)w * 8 + 1];
    }
    red[0][tid] = s;
    red[1][tid] = z;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) {
            red[0][tid] += red[0][tid + off];
            red[1][tid] += red[1][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double d2 = hyp[0] + hyp[1] - red[0][0];
        if (!(d2 > 0.0)) *info = N0 + 1;
        const double dd = sqrt(d2), zn = (y[N0] - mean[N0] - red[1][0]) / dd;
        dz[0] = dd;
        dz[1] = zn;
        scal[0] += 2.0 * log(dd);
        scal[1] = __builtin_fma(zn, zn, scal[1]);
    }
}

__global__ __launch_bounds__(256) void append_write_kernel(double* __restrict__ A, int ld, int Np, int N0,
                                                           const double* __restrict__ l, const double* __restrict__ w,
                                                           const double* __restrict__ dz, double* __restrict__ Linv,
                                                           double* __restrict__ U, double* __restrict__ Dinv,
                                                           double* __restrict__ Dinv2, double* __restrict__ inv16) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c > N0) return;
    const double dd = dz[0];
    const bool diag = c == N0;
    const double t = diag ? 1.0 / dd : -w[c] / dd;           // row N0 of L⁻¹
    A[(size_t)c * ld + N0] = diag ? dd : l[c];               // row N0 of L
    if (diag) A[(size_t)N0 * ld + Np] = dz[1];               // z_new (the δᵀ/z row block)
    Linv[(size_t)c * ld + N0] = t;
    U[(size_t)N0 * ld + c] = t;
    const int b1 = N0 / BLK, b2 = N0 / PRED_RB, b16 = N0 / 16;
    if (c >= b1 * BLK) Dinv[(size_t)b1 * BLK * BLK + (size_t)(c - b1 * BLK) * BLK + (N0 - b1 * BLK)] = t;
    if (c >= b2 * PRED_RB) Dinv2[(size_t)b2 * PRED_RB * PRED_RB + (size_t)(c - b2 * PRED_RB) * PRED_RB + (N0 - b2 * PRED_RB)] = t;
    if (c >= b16 * 16) inv16[(size_t)b1 * 8 * 256 + (size_t)(b16 - b1 * 8) * 256 + (c - b16 * 16) * 16 + (N0 - b16 * 16)] = t;
}

// mode: 0 plain (σ² = α² − Σv² + 1e-18), 1 gradient observations (max(0, α² − Σv²)), 2 nonstationary (−Σv²; gibbs_var_kernel follows)
__global__ __launch_bounds__(256) void winv_finish_kernel(const double* __restrict__ part, int nwg, int M,
                                                          const double* __restrict__ mean_s, double amp2, int mode,
                                                          double* __restrict__ mu, double* __restrict__ var) {
    __shared__ double red[8][256];
    const int tid = threadIdx.x;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    for (int w = tid; w < nwg; w += 256)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += part[(size_t)w * 8 + q];
#pragma unroll
    for (int q = 0; q < 8; ++q) red[q][tid] = acc[q];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {               // fixed-order tree: deterministic
        if (tid < off)
#pragma unroll
            for (int q = 0; q < 8; ++q) red[q][tid] += red[q][tid + off];
        __syncthreads();
    }
    if (tid < M) {
        const double s = red[2 * tid][0], z = red[2 * tid + 1][0];
        mu[tid] = (mean_s ? mean_s[tid] : 0.0) + z;
        var[tid] = mode == 2 ? -s : mode == 1 ? fmax(0.0, amp2 - s) : amp2 - s + PREDICT_JITTER;
    }
}

// ------------------------------------------------------------------------------------------
// 256×256 diagonal-block inverses from the 128×128 ones (prediction with 256-row steps halves the
// V-slab re-reads and the number of dependent steps per candidate tile):
//     inv [ A 0 ; B C ] = [ A⁻¹ 0 ; −C⁻¹ B A⁻¹  C⁻¹ ]
// small_gemm128_kernel: C_s = alpha · A_s · B_s for 128×128 column-major operands (32×32 output
// tile per workgroup, grid (16, pairs)); dinv_pair_assemble_kernel copies the diagonal quadrants.
// Runs once per factorisation, off the prediction kernel's path (≈10 µs).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void small_gemm128_kernel(const double* __restrict__ Abase, int lda, size_t sA,
                                                            const double* __restrict__ Bbase, int ldb, size_t sB,
                                                            double* __restrict__ Cbase, int ldc, size_t sC, double alpha) {
    __shared__ double As[32][33], Bs[32][33];
    const double* A = Abase + (size_t)blockIdx.y * sA;
    const double* B = Bbase + (size_t)blockIdx.y * sB;
    double* C = Cbase + (size_t)blockIdx.y * sC;
    const int r0 = (blockIdx.x & 3) * 32, c0 = (blockIdx.x >> 2) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // ty 0..7 → 4 columns each
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < BLK; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[ty * 4 + j][tx] = A[(size_t)(k0 + ty * 4 + j) * lda + r0 + tx];      // As[k][r]
            Bs[ty * 4 + j][tx] = B[(size_t)(c0 + ty * 4 + j) * ldb + k0 + tx];      // Bs[c][k]
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const double a = As[kk][tx];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fma(a, Bs[ty * 4 + j][kk], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) C[(size_t)(c0 + ty * 4 + j) * ldc + r0 + tx] = alpha * acc[j];
}

// Dinv2_p (256×256, column-major): diagonal quadrants = Dinv128 of blocks 2p, 2p+1; upper right = 0.
__global__ __launch_bounds__(256) void dinv_pair_assemble_kernel(const double* __restrict__ Dinv128,
                                                                 double* __restrict__ Dinv2) {
    const int p = blockIdx.y, c = blockIdx.x;                     // column c of the 256×256 block
    const double* src = Dinv128 + (size_t)(2 * p + (c >> 7)) * BLK * BLK + (size_t)(c & 127) * BLK;
    double* dst = Dinv2 + (size_t)p * 4 * BLK * BLK + (size_t)c * 2 * BLK;
    const int r = threadIdx.x;                                    // 0..255
    if (c < BLK) {
        if (r < BLK) dst[r] = src[r];                             // lower-left quadrant is written by the GEMM
    } else {
        dst[r] = (r < BLK) ? 0.0 : src[r - BLK];
    }
}

// ------------------------------------------------------------------------------------------
// analytic gradients of the posterior moments w.r.t. the candidates,
//     ∇μ(x*)  = ∇m(x*) + Σ_i a_i ∇k(x_i, x*),   a = (K+σ²I)⁻¹(y−m) = L⁻ᵀ z
//     ∇σ²(x*) = −2 Σ_i w_i ∇k(x_i, x*),           w = (K+σ²I)⁻¹ k* = L⁻ᵀ v
// (what the reference obtains by pushing ForwardDiff duals through AbstractGPs,
//  src/acquisition_maximizers/optimization.jl:36,89-118).  The adjoint (backward) substitution
// W = L⁻ᵀ V runs on the V slabs the prediction kernel left behind, in place, with the same
// 256-row-step / register-ring machinery on a transposed copy of the factor; `a` is solved once per
// factorisation by a chain of small GEMV launches.
// ------------------------------------------------------------------------------------------
// out[c + r*ldo] = in[r + c*ldi] for an n×n matrix (batched over blockIdx.z with the given strides)
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in, int ldi, size_t si,
                                                        double* __restrict__ out, int ldo, size_t so, int n) {
    __shared__ double t[64][65];
    const double* I = in + (size_t)blockIdx.z * si;
    double* O = out + (size_t)blockIdx.z * so;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4)
        if (r0 + tx < n && c0 + c < n) t[c][tx] = I[(size_t)(c0 + c) * ldi + r0 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4)
        if (c0 + tx < n && r0 + r < n) O[(size_t)(r0 + r) * ldo + c0 + tx] = t[tx][r];
}

// a = L⁻ᵀ z (= (K+σ²I)⁻¹(y−m)), once per factorisation, in 256-row steps from the last to the first:
//     bt_gemv_partial_kernel   partial[c][r] = Σ_{k in 256-column chunk c} Lᵀ[i0+r, k] a[k]      (one workgroup per chunk)
//     bt_finish_kernel         a[i0..i0+255] = Dinv2ᵀ_i (z_i − Σ_c partial[c])                   (fixed summation order)
__global__ __launch_bounds__(256) void bt_gemv_partial_kernel(const double* __restrict__ LT, int ldt, int ib,
                                                              const double* __restrict__ a, double* __restrict__ partial) {
    const int r = threadIdx.x, kc = (ib + 1 + blockIdx.x) * PRED_RB;
    const double* col = LT + (size_t)ib * PRED_RB + r + (size_t)kc * ldt;
    double s = 0.0;
#pragma unroll 8
    for (int k = 0; k < PRED_RB; ++k) s = __builtin_fma(col[(size_t)k * ldt], a[kc + k], s);
    partial[(size_t)blockIdx.x * PRED_RB + r] = s;
}

__global__ __launch_bounds__(256) void bt_finish_kernel(const double* __restrict__ A, int ld, int Np, int N, int ib, int nchunks,
                                                        const double* __restrict__ partial, const double* __restrict__ DT2,
                                                        double* __restrict__ a) {
    __shared__ double rv[PRED_RB];
    const int r = threadIdx.x, i = ib * PRED_RB + r;
    double v = (i < N) ? A[(size_t)i * ld + Np] : 0.0;      // z_i sits in row Np of the factor array
    for (int c = 0; c < nchunks; ++c) v -= partial[(size_t)c * PRED_RB + r];
    rv[r] = v;
    __syncthreads();
    const double* D = DT2 + (size_t)ib * PRED_RB * PRED_RB;
    double s = 0.0;
#pragma unroll 16
    for (int k = 0; k < PRED_RB; ++k) s = __builtin_fma(D[r + (size_t)k * PRED_RB], rv[k], s);   // upper triangular: zeros below the diagonal
    a[i] = s;
}

// W = L⁻ᵀ V in place on every slab: for the row steps from the last to the first,
//     R_i = V_i − Σ_{j>i} Lᵀ_ij W_j        (GemmDirect: A = rows of LT, B = this slab's finished rows)
//     W_i = Dinv2_iᵀ R_i                    (R in LDS; DT2 holds the transposed 256×256 inverses: upper
//                                            triangular, a row only needs the k ≥ its own 64-row slice)
template <class G>
__global__ __launch_bounds__(G::NTHREADS) void backsolve_kernel(const double* __restrict__ LT, int ldt, int Np,
                                                                const double* __restrict__ DT2,
                                                                double* __restrict__ Vscratch) {
    static_assert(G::WC == 1 && G::BM == 2 * BLK, "written for 256-row steps");
    constexpr int RB = G::BM, BN = G::BN, TM = G::TM, TN = G::TN, LDR = PredictLds<G>::LDR;
    extern __shared__ double lds[];
    double* Rs = lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave, wc = 0;
    double* V = Vscratch + (size_t)blockIdx.x * Np * BN;
    const int nb = Np / RB;
    for (int ib = nb - 1; ib >= 0; --ib) {
        v4d acc[TM][TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
        const int k0 = (ib + 1) * RB;
        G::template run<1>(LT + (size_t)ib * RB + (size_t)k0 * ldt, ldt, V + (size_t)k0 * BN, BN, Np - k0, acc);
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int row = G::row_of(wr, m, lane);
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = G::col_of(wc, n, i, lane);
                    Rs[row * LDR + col] = V[(size_t)(ib * RB + row) * BN + col] - acc[m][n][i];
                }
        }
        __syncthreads();                                   // R tile complete
        v4d acc2[TM][TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc2[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
        const int kk0 = (TM * 16) * wr;                    // upper triangular: rows of wave w need k >= 64 w
        G::run_Blds(DT2 + (size_t)ib * RB * RB + (size_t)kk0 * RB, RB, Rs + (size_t)kk0 * LDR, LDR, RB - kk0, acc2);
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int row = ib * RB + G::row_of(wr, m, lane);
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) V[(size_t)row * BN + G::col_of(wc, n, i, lane)] = acc2[m][n][i];
        }
        __syncthreads();                                   // W_ib visible to the workgroup; Rs reusable
    }
}

// Few candidates: the adjoint substitution W = L⁻ᵀV right-looking, from the LAST 256-row step to the first, in
// place on the slabs (the unfinished rows hold the running residual):
//   few_back_finish_kernel   W_i = Dinv2ᵀ_i R_i                                   — one workgroup per tile
//   few_back_update_kernel   R_j −= Lᵀ[j, i] W_i for every EARLIER row block j    — one workgroup per 128 rows per tile
template <class G>
__global__ __launch_bounds__(G::NTHREADS) void few_back_finish_kernel(const double* __restrict__ DT2, int Np, int ib,
                                                                      double* __restrict__ Vslabs) {
    constexpr int RB = G::BM, TM = G::TM, TN = G::TN, LDR = PredictLds<G>::LDR, BN = 32;
    extern __shared__ double lds[];
    double* Rs = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* V = Vslabs + (size_t)blockIdx.x * Np * BN + (size_t)ib * RB * BN;
#pragma unroll 8
    for (int q = 0; q < RB * 32 / 256; ++q) {
        const int e = tid + 256 * q;
        Rs[(e >> 5) * LDR + (e & 31)] = V[e];
    }
    __syncthreads();
    v4d acc2[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc2[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
    const int kk0 = (TM * 16) * wave;                      // upper triangular: rows of wave w need k >= 64 w
    G::run_Blds(DT2 + (size_t)ib * RB * RB + (size_t)kk0 * RB, RB, Rs + (size_t)kk0 * LDR, LDR, RB - kk0, acc2);
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = G::row_of(wave, m, lane);
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) V[(size_t)row * BN + G::col_of(0, n, i, lane)] = acc2[m][n][i];
    }
}

template <class GU>
__global__ __launch_bounds__(GU::NTHREADS) void few_back_update_kernel(const double* __restrict__ LT, int ldt, int Np, int ib,
                                                                       double* __restrict__ Vslabs) {
    static_assert(GU::WC == 1 && GU::BM == BLK && GU::BN == 32, "128×32 tiles");
    constexpr int TM = GU::TM, TN = GU::TN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* V = Vslabs + (size_t)blockIdx.x * Np * 32;
    const int r0 = blockIdx.y * BLK;                         // rows before step ib
    double* Rb = V + (size_t)r0 * 32;
    v4d acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = GU::row_of(wave, m, lane);
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][n][i] = Rb[row * 32 + GU::col_of(0, n, i, lane)];
    }
    GU::template run<-1>(LT + (size_t)r0 + (size_t)ib * PRED_RB * ldt, ldt, V + (size_t)ib * PRED_RB * 32, 32, PRED_RB, acc);
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = GU::row_of(wave, m, lane);
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) Rb[row * 32 + GU::col_of(0, n, i, lane)] = acc[m][n][i];
    }
}

// ∇μ, ∇σ² from the W slabs:  with q_i = α² h(r_i),
//     ∇μ_m  = ∇m_m + (u*_m Σ_i a_i q_i − Σ_i a_i q_i u_i,m) / λ_m ,   ∇σ²_m = −2 (u*_m Σ_i w_i q_i − Σ_i w_i q_i u_i,m) / λ_m
// (u = x ⊘ λ).  One workgroup per 32-candidate slab: lanes run along the candidates, 8 row subsets;
// the rows' coordinates and a_i are staged through LDS 64 rows at a time (broadcast reads), the W
// loads of a chunk are issued together.  Dimensions are handled 16 at a time (registers).
constexpr int GRAD_MAX_D = 16;
constexpr int GRAD_CHUNK = 64;
__global__ __launch_bounds__(256) void grad_accum_kernel(const double* __restrict__ Wslabs, const double* __restrict__ avec,
                                                         int Np, int N, const double* __restrict__ Xsc,
                                                         const double* __restrict__ Csc, int d, int Mp, int M, int kern,
                                                         double amp2, const double* __restrict__ invlam,
                                                         const unsigned char* __restrict__ discrete,
                                                         const double* __restrict__ mean_grad,
                                                         double* __restrict__ dmu, double* __restrict__ dvar,
                                                         double* __restrict__ part) {
    // gridDim.y > 1 (few tiles, d <= 16): the rows are split over gridDim.y workgroups per tile, each writes its
    // sums to part[(tile·gridDim.y + y)][2(GRAD_MAX_D+1)][32]; grad_finalize_kernel adds them in a fixed order
    constexpr int BN = 32;
    extern __shared__ double glds[];
    double* xs = glds;                                       // [d][GRAD_CHUNK] scaled coordinates of the chunk's rows
    double* as = xs + (size_t)d * GRAD_CHUNK;                // [GRAD_CHUNK]     a_i
    double* red = as + GRAD_CHUNK;                           // [8][2(GRAD_MAX_D+1)][BN]
    const int tid = threadIdx.x, c = tid & 31, rs = tid >> 5;
    const int j = blockIdx.x * BN + c;
    const double* W = Wslabs + (size_t)blockIdx.x * Np * BN;
    for (int m0 = 0; m0 < d; m0 += GRAD_MAX_D) {            // d > 16: passes of 16 dimensions
        const int dm = (d - m0 < GRAD_MAX_D) ? (d - m0) : GRAD_MAX_D;
        double S1 = 0.0, S2 = 0.0, T1[GRAD_MAX_D], T2[GRAD_MAX_D];
#pragma unroll
        for (int m = 0; m < GRAD_MAX_D; ++m) T1[m] = T2[m] = 0.0;
        const int nchunk = (N + GRAD_CHUNK - 1) / GRAD_CHUNK;
        const int cpb = (nchunk + gridDim.y - 1) / gridDim.y;
        const int rbeg = blockIdx.y * cpb * GRAD_CHUNK;
        const int rend = (rbeg + cpb * GRAD_CHUNK < N) ? rbeg + cpb * GRAD_CHUNK : N;
        for (int r0 = rbeg; r0 < rend; r0 += GRAD_CHUNK) {
            __syncthreads();
            for (int idx = tid; idx < d * GRAD_CHUNK; idx += 256) {
                const int m = idx / GRAD_CHUNK, rr = idx - m * GRAD_CHUNK;
                xs[idx] = Xsc[(size_t)m * Np + r0 + rr];     // rows beyond N are padding inside Np: harmless, masked below
            }
            if (tid < GRAD_CHUNK) as[tid] = avec[r0 + tid];
            double w[GRAD_CHUNK / 8];
#pragma unroll
            for (int k = 0; k < GRAD_CHUNK / 8; ++k) w[k] = W[(size_t)(r0 + rs + 8 * k) * BN + c];
            __syncthreads();
#pragma unroll 2
            for (int k = 0; k < GRAD_CHUNK / 8; ++k) {
                const int rr = rs + 8 * k;
                if (r0 + rr >= N) break;
                double r2 = 0.0;
                for (int m = 0; m < d; ++m) {
                    const double diff = Csc[(size_t)m * Mp + j] - xs[m * GRAD_CHUNK + rr];
                    r2 = __builtin_fma(diff, diff, r2);
                }
                const double q = amp2 * kappa_prime_over_r_r2(kern, r2);
                const double qa = q * as[rr], qw = q * w[k];
                S1 += qa;
                S2 += qw;
#pragma unroll
                for (int m = 0; m < GRAD_MAX_D; ++m)
                    if (m < dm) {
                        const double x = xs[(m0 + m) * GRAD_CHUNK + rr];
                        T1[m] = __builtin_fma(qa, x, T1[m]);
                        T2[m] = __builtin_fma(qw, x, T2[m]);
                    }
            }
        }
        __syncthreads();
        double* rd = red + (size_t)rs * (2 * (GRAD_MAX_D + 1)) * BN;
        rd[0 * BN + c] = S1;
        rd[1 * BN + c] = S2;
#pragma unroll
        for (int m = 0; m < GRAD_MAX_D; ++m) {
            rd[(2 + 2 * m) * BN + c] = T1[m];
            rd[(3 + 2 * m) * BN + c] = T2[m];
        }
        __syncthreads();
        if (gridDim.y > 1) {
            // this workgroup's sums (over its 8 row subsets) go to global; the finalize kernel finishes
            double* pw = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (2 * (GRAD_MAX_D + 1)) * BN;
            for (int slot = rs; slot < 2 * (GRAD_MAX_D + 1); slot += 8) {
                double v = 0.0;
                for (int k = 0; k < 8; ++k) v += red[((size_t)k * (2 * (GRAD_MAX_D + 1)) + slot) * BN + c];
                pw[slot * BN + c] = v;
            }
            return;
        }
        if (rs == 0 && j < M) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < 8; ++k) {
                const double* rk = red + (size_t)k * (2 * (GRAD_MAX_D + 1)) * BN;
                s1 += rk[c];
                s2 += rk[BN + c];
            }
            for (int m = 0; m < dm; ++m) {
                double t1 = 0.0, t2 = 0.0;
                for (int k = 0; k < 8; ++k) {
                    const double* rk = red + (size_t)k * (2 * (GRAD_MAX_D + 1)) * BN;
                    t1 += rk[(2 + 2 * m) * BN + c];
                    t2 += rk[(3 + 2 * m) * BN + c];
                }
                const int mm = m0 + m;
                const double u = Csc[(size_t)mm * Mp + j], il = invlam[mm];
                const bool disc = discrete && discrete[mm];
                const double g1 = disc ? 0.0 : (u * s1 - t1) * il;
                const double g2 = disc ? 0.0 : -2.0 * (u * s2 - t2) * il;
                dmu[(size_t)j * d + mm] = g1 + (mean_grad ? mean_grad[(size_t)j * d + mm] : 0.0);
                dvar[(size_t)j * d + mm] = g2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tracked candidates (boss_track_t): the V = L⁻¹K* slabs of a fixed candidate set stay resident, so
// after boss_gp_append the predictive moments are UPDATED instead of re-solved — per new observation
// r one more row of V,   v_r = (k(x_r, x*) − Σ_{i<r} L[r,i] V[i,·]) / L[r,r] ,   σ² −= v_r² ,  μ += v_r z_r ,
// an O(N·M) pass over the slabs (≈0.25 GB at N=4096, M=8192) instead of the O(N²M) substitution.
// Up to 8 new rows per launch (one read of V for all of them).  One workgroup per 32-candidate slab:
// lanes along the candidates, 8 subsets of the old rows; the new rows of L are staged through LDS.
// ------------------------------------------------------------------------------------------
constexpr int TRACK_ROWS = 8;
__global__ __launch_bounds__(256) void track_append_kernel(const double* __restrict__ A, int ld, int Np, int N0, int n,
                                                           double* __restrict__ Vslabs, int Ncap,
                                                           const double* __restrict__ Xsc, int Npx,
                                                           const double* __restrict__ Csc, int d, int Mp, int M, int kern,
                                                           double amp2, double* __restrict__ mu, double* __restrict__ var) {
    constexpr int BN = 32, CH = 64;
    __shared__ double Lr[TRACK_ROWS][CH];
    __shared__ double red[8][TRACK_ROWS][BN];
    const int tid = threadIdx.x, c = tid & 31, rs = tid >> 5;
    double* V = Vslabs + (size_t)blockIdx.x * Ncap * BN;
    double acc[TRACK_ROWS];
#pragma unroll
    for (int q = 0; q < TRACK_ROWS; ++q) acc[q] = 0.0;
    for (int i0 = 0; i0 < N0; i0 += CH) {
        __syncthreads();
        for (int idx = tid; idx < TRACK_ROWS * CH; idx += 256) {
            const int q = idx / CH, ii = idx - q * CH;
            Lr[q][ii] = (q < n && i0 + ii < N0) ? A[(size_t)(i0 + ii) * ld + N0 + q] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CH / 8; ++k) {
            const int ii = rs + 8 * k;
            const double v = (i0 + ii < N0) ? V[(size_t)(i0 + ii) * BN + c] : 0.0;
#pragma unroll
            for (int q = 0; q < TRACK_ROWS; ++q) acc[q] = __builtin_fma(Lr[q][ii], v, acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < TRACK_ROWS; ++q) red[rs][q][c] = acc[q];
    __syncthreads();
    if (rs == 0) {
        const int j = blockIdx.x * BN + c;
        double vnew[TRACK_ROWS];
        double dvar = 0.0, dmu = 0.0;
        for (int q = 0; q < n; ++q) {
            double dot = 0.0;
            for (int k = 0; k < 8; ++k) dot += red[k][q][c];
            double r2 = 0.0;
            for (int m = 0; m < d; ++m) {
                const double diff = Csc[(size_t)m * Mp + j] - Xsc[(size_t)m * Npx + N0 + q];
                r2 = __builtin_fma(diff, diff, r2);
            }
            double t = amp2 * kappa_r2(kern, r2) - dot;
            for (int qq = 0; qq < q; ++qq) t = __builtin_fma(-A[(size_t)(N0 + qq) * ld + N0 + q], vnew[qq], t);
            const double v = t / A[(size_t)(N0 + q) * ld + N0 + q];
            vnew[q] = v;
            V[(size_t)(N0 + q) * BN + c] = v;
            dvar = __builtin_fma(v, v, dvar);
            dmu = __builtin_fma(v, A[(size_t)(N0 + q) * ld + Np], dmu);      // z_r sits in row Np of the factor array
        }
        if (j < M) {
            var[j] -= dvar;
            mu[j] += dmu;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Gradient of the log marginal likelihood w.r.t. the hyper-parameters (second half):
//     ∂ℓ/∂θ = ½ Σ_ij G_ij ∂K_ij/∂θ ,   G = a aᵀ − K⁻¹ ,  K⁻¹ = L⁻ᵀL⁻¹
// (what ForwardDiff / Zygote deliver to OptimizationMAP, src/model_fitters/optimization.jl:146-164).
//   linvt_kernel        LinvT[c + k·ldt] = (L⁻¹)[k, c]: forward substitution of the identity, 32 columns per
//                       workgroup, the prediction kernel's 256-row-step machinery; the result matrix is also
//                       the GEMM's B operand (ldb = ldt), and each tile starts at its own diagonal step
//   kinv_syrk_kernel    K⁻¹ = LinvT·LinvTᵀ, 128×128 tiles, k runs from the tile's row block to the end
//   avec_partial_kernel a = LinvT z (= L⁻ᵀ z), 8 k-chunks per 256 rows, summed in a fixed order
//   llgrad_tile_kernel  per 64×64 lower tile: Σ G_ij α² h(r_ij) Δu²_ij,m (m < d), tr K⁻¹, ‖a‖²
//   llgrad_reduce_kernel deterministic sum of the tile partials
// ------------------------------------------------------------------------------------------
template <class G>
__global__ __launch_bounds__(G::NTHREADS) void linvt_kernel(const double* __restrict__ A, int ld, int Np,
                                                            const double* __restrict__ Dinv2,
                                                            double* __restrict__ LinvT, int ldt) {
    static_assert(G::WC == 1 && G::BM == 2 * BLK && G::PM == 2 && G::WR == 4, "written for 256-row steps, 4 waves");
    constexpr int RB = G::BM, BN = G::BN, TM = G::TM, TN = G::TN, LDR = PredictLds<G>::LDR;
    extern __shared__ double lds[];
    double* Rs = lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave, wc = 0;
    const int c0 = blockIdx.x * BN;
    double* V = LinvT + c0;                                  // V(row k, column c) at V[c + k·ldt]
    const int nb = Np / RB, ib0 = c0 / RB;
    for (int ib = ib0; ib < nb; ++ib) {
        v4d acc[TM][TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
        if (ib > ib0)
            G::template run<1>(A + (size_t)ib * RB + (size_t)ib0 * RB * ld, ld, V + (size_t)ib0 * RB * ldt, ldt, (ib - ib0) * RB, acc);
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int row = G::row_of(wr, m, lane);
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = G::col_of(wc, n, i, lane);
                    Rs[row * LDR + col] = ((ib * RB + row == c0 + col) ? 1.0 : 0.0) - acc[m][n][i];
                }
        }
        __syncthreads();
        v4d acc2[TM][TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc2[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
        G::run_Blds_tri(Dinv2 + (size_t)ib * RB * RB, RB, Rs, LDR, acc2);
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int row = ib * RB + G::tri_row_of(wr, m, lane);
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) V[(size_t)row * ldt + G::col_of(wc, n, i, lane)] = acc2[m][n][i];
        }
        __syncthreads();
    }
}

// needs SyrkG (potrf.hpp is included after this header by bosship.hip, so the kernel is templated on it)
// L⁻¹ by recursive doubling instead of a substitution (the substitution's 32-column workgroups are
// latency-bound and fill half the chip).  With L = [A 0; B C]:  L⁻ᵀ = [A⁻ᵀ  X; 0  C⁻ᵀ],  X = −(A⁻ᵀ Bᵀ) C⁻ᵀ.
// Two work matrices: U = L⁻ᵀ (upper, what the consumers below read as LinvT) and Lw = L⁻¹ (lower); both GEMMs
// are of the C += A·Bᵀ form of GemmDirect when the second factor is taken from the other matrix:
//   phase 1   T1 = U_A · Bᵀ            (T1 parked in Lw's structurally-zero upper-right block; k ≥ row block: U_A is upper)
//   phase 2   X  = −T1 · (Lw_C)ᵀ       (k ≤ column block: Lw_C is lower)   → U[A-range, C-range] = X,  Lw[C-range, A-range] = Xᵀ
// Seeded with the 256×256 diagonal inverses (Dinv2), then chunk sizes s = 256, 512, … ; one launch per phase and level,
// grid.y = pair of chunks (the last pair may be ragged or absent: Np/256 need not be a power of two).
__global__ __launch_bounds__(256) void linv_seed_kernel(const double* __restrict__ Dinv2, double* __restrict__ Lw, int ldw,
                                                        double* __restrict__ U, int ldu) {
    const int b = blockIdx.y, c = blockIdx.x, r = threadIdx.x;
    const double v = (r >= c) ? Dinv2[(size_t)b * PRED_RB * PRED_RB + r + (size_t)c * PRED_RB] : 0.0;   // upper half holds scratch
    const size_t o = (size_t)b * PRED_RB;
    Lw[(o + r) + (o + c) * ldw] = v;
    U[(o + c) + (o + r) * ldu] = v;
}

template <class SG, int PHASE>
__global__ __launch_bounds__(256, 2) void linv_level_kernel(const double* __restrict__ Afac, int ld, double* __restrict__ Lw,
                                                            int ldw, double* __restrict__ U, int ldu, int Np, int s) {
    const int a0 = 2 * blockIdx.y * s, c0 = a0 + s;
    if (c0 >= Np) return;                                    // unpaired last chunk
    const int sC = (Np - c0 < s) ? Np - c0 : s;
    const int tm = s / BLK, tn = sC / BLK;
    const int t = blockIdx.x;
    if (t >= tm * tn) return;
    const int mi = t % tm, ni = t / tm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / SG::WC, wc = wave % SG::WC;
    v4d acc[SG::TM][SG::TN];
#pragma unroll
    for (int m = 0; m < SG::TM; ++m)
#pragma unroll
        for (int n = 0; n < SG::TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
    const size_t r0 = (size_t)a0 + mi * BLK, q0 = (size_t)c0 + ni * BLK;
    if (PHASE == 1) {
        const int k0 = mi * BLK;
        SG::template run<1>(U + r0 + ((size_t)a0 + k0) * ldu, ldu, Afac + q0 + ((size_t)a0 + k0) * ld, ld, s - k0, acc);
        double* T1 = Lw + r0 + q0 * ldw;
#pragma unroll
        for (int m = 0; m < SG::TM; m += 2)
#pragma unroll
            for (int n = 0; n < SG::TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v2d c2 = {acc[m][n][i], acc[m + 1][n][i]};
                    *reinterpret_cast<v2d*>(T1 + SG::row_of(wr, m, lane) + (size_t)SG::col_of(wc, n, i, lane) * ldw) = c2;
                }
    } else {
        SG::template run<1>(Lw + r0 + (size_t)c0 * ldw, ldw, Lw + q0 + (size_t)c0 * ldw, ldw, (ni + 1) * BLK, acc);
        double* X = U + r0 + q0 * ldu;
        double* Xt = Lw + q0 + r0 * ldw;
#pragma unroll
        for (int m = 0; m < SG::TM; m += 2)
#pragma unroll
            for (int n = 0; n < SG::TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = SG::row_of(wr, m, lane), col = SG::col_of(wc, n, i, lane);
                    v2d c2 = {-acc[m][n][i], -acc[m + 1][n][i]};
                    *reinterpret_cast<v2d*>(X + row + (size_t)col * ldu) = c2;
                    Xt[col + (size_t)row * ldw] = c2[0];
                    Xt[col + (size_t)(row + 1) * ldw] = c2[1];
                }
    }
}

template <class SG>
__global__ __launch_bounds__(256, 2) void kinv_syrk_kernel(const double* __restrict__ LinvT, int ldt, int Np,
                                                           double* __restrict__ Kinv, int ldk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / SG::WC, wc = wave % SG::WC;
    const int t = blockIdx.x;
    int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    while (I * (I + 1) / 2 > t) --I;
    const int J = t - I * (I + 1) / 2;
    const int k0 = I * BLK;                                  // LinvT[i, k] = 0 for k < i
    v4d acc[SG::TM][SG::TN];
#pragma unroll
    for (int m = 0; m < SG::TM; ++m)
#pragma unroll
        for (int n = 0; n < SG::TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
    SG::template run<1>(LinvT + (size_t)I * BLK + (size_t)k0 * ldt, ldt, LinvT + (size_t)J * BLK + (size_t)k0 * ldt, ldt, Np - k0, acc);
    double* C = Kinv + (size_t)I * BLK + (size_t)J * BLK * ldk;
#pragma unroll
    for (int m = 0; m < SG::TM; m += 2)
#pragma unroll
        for (int n = 0; n < SG::TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v2d c2 = {acc[m][n][i], acc[m + 1][n][i]};
                *reinterpret_cast<v2d*>(C + SG::row_of(wr, m, lane) + (size_t)SG::col_of(wc, n, i, lane) * ldk) = c2;
            }
}

// partial[chunk][i] = Σ_{k in chunk} LinvT[i, k] z_k   (z_k in row Np of the factor array; k < N)
__global__ __launch_bounds__(256) void avec_partial_kernel(const double* __restrict__ LinvT, int ldt, int Np, int N,
                                                           const double* __restrict__ A, int ld,
                                                           double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nch = gridDim.y, ch = blockIdx.y;
    const int kbeg0 = (blockIdx.x * 256 / PRED_RB) * PRED_RB;     // first written column of these rows
    const int span = (Np - kbeg0 + nch - 1) / nch;
    const int kb = kbeg0 + ch * span, ke = (kb + span < N) ? kb + span : N;
    double s = 0.0;
    for (int k = kb; k < ke; ++k) s = __builtin_fma(LinvT[(size_t)k * ldt + i], A[(size_t)k * ld + Np], s);
    partial[(size_t)ch * Np + i] = s;
}

constexpr int LLG_MAX_D = 32;
// out[tile][0..d-1] = Σ_{i>j in tile} G_ij α² h(r_ij) Δu²_ij,m ;  out[tile][d] = Σ_i K⁻¹_ii , out[tile][d+1] = Σ_i a_i²  (diagonal tiles)
__global__ __launch_bounds__(256) void llgrad_tile_kernel(const double* __restrict__ Xsc, int d, int N, int Np, int kern,
                                                          double amp2, const double* __restrict__ Kinv, int ldk,
                                                          const double* __restrict__ apart, int nch,
                                                          double* __restrict__ out) {
    __shared__ double xj[LLG_MAX_D][64];
    __shared__ double aj[64];
    __shared__ double red[256];
    const int tid = threadIdx.x, t = blockIdx.x;
    int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    const int r = tid & 63, cg = tid >> 6;
    const int i = bi * 64 + r;
    for (int idx = tid; idx < d * 64; idx += 256) xj[idx >> 6][idx & 63] = Xsc[(size_t)(idx >> 6) * Np + bj * 64 + (idx & 63)];
    if (tid < 64) {
        double s = 0.0;
        for (int c = 0; c < nch; ++c) s += apart[(size_t)c * Np + bj * 64 + tid];
        aj[tid] = s;
    }
    double ai = 0.0;
    for (int c = 0; c < nch; ++c) ai += apart[(size_t)c * Np + i];
    __syncthreads();
    double S[LLG_MAX_D];
#pragma unroll
    for (int m = 0; m < LLG_MAX_D; ++m) S[m] = 0.0;
    double tr = 0.0, aa = 0.0;
    double xi[LLG_MAX_D];
#pragma unroll
    for (int m = 0; m < LLG_MAX_D; ++m) xi[m] = (m < d) ? Xsc[(size_t)m * Np + i] : 0.0;
    for (int c = 0; c < 16; ++c) {
        const int jl = cg * 16 + c, j = bj * 64 + jl;
        if (i >= N || j >= N) continue;
        if (i == j) {
            tr += Kinv[(size_t)j * ldk + i];
            aa += ai * ai;
            continue;
        }
        if (i < j) continue;
        double r2 = 0.0, du2[LLG_MAX_D];
#pragma unroll
        for (int m = 0; m < LLG_MAX_D; ++m) {
            const double df = (m < d) ? xi[m] - xj[m][jl] : 0.0;
            du2[m] = df * df;
            r2 += du2[m];
        }
        const double g = ai * aj[jl] - Kinv[(size_t)j * ldk + i];
        const double q = g * amp2 * kappa_prime_over_r_r2(kern, r2);
#pragma unroll
        for (int m = 0; m < LLG_MAX_D; ++m) S[m] = __builtin_fma(q, du2[m], S[m]);
    }
    // workgroup reduction of the d + 2 sums (one at a time; d is small)
    for (int m = 0; m < d + 2; ++m) {
        double v = (m < d) ? 0.0 : (m == d ? tr : aa);
#pragma unroll
        for (int mm = 0; mm < LLG_MAX_D; ++mm)
            if (mm == m) v = S[mm];
        if (m >= d) v = (m == d) ? tr : aa;
        __syncthreads();
        red[tid] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) out[(size_t)t * (d + 2) + m] = red[0];
    }
}

__global__ __launch_bounds__(256) void llgrad_reduce_kernel(const double* __restrict__ parts, int ntiles, int nv,
                                                            double* __restrict__ out) {
    __shared__ double red[256];
    const int m = blockIdx.x;
    double s = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += 256) s += parts[(size_t)t * nv + m];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[m] = red[0];
}

__global__ __launch_bounds__(32) void grad_finalize_kernel(const double* __restrict__ part, int rsplit,
                                                           const double* __restrict__ Csc, int d, int Mp, int M,
                                                           const double* __restrict__ invlam,
                                                           const unsigned char* __restrict__ discrete,
                                                           const double* __restrict__ mean_grad, double* __restrict__ dmu,
                                                           double* __restrict__ dvar) {
    constexpr int BN = 32, NS = 2 * (GRAD_MAX_D + 1);
    const int c = threadIdx.x, j = blockIdx.x * BN + c;
    if (j >= M) return;
    const double* pt = part + (size_t)blockIdx.x * rsplit * NS * BN;
    double s1 = 0.0, s2 = 0.0;
    for (int y = 0; y < rsplit; ++y) {
        s1 += pt[((size_t)y * NS + 0) * BN + c];
        s2 += pt[((size_t)y * NS + 1) * BN + c];
    }
    for (int m = 0; m < d; ++m) {
        double t1 = 0.0, t2 = 0.0;
        for (int y = 0; y < rsplit; ++y) {
            t1 += pt[((size_t)y * NS + 2 + 2 * m) * BN + c];
            t2 += pt[((size_t)y * NS + 3 + 2 * m) * BN + c];
        }
        const double u = Csc[(size_t)m * Mp + j], il = invlam[m];
        const bool disc = discrete && discrete[m];
        const double g1 = disc ? 0.0 : (u * s1 - t1) * il;
        const double g2 = disc ? 0.0 : -2.0 * (u * s2 - t2) * il;
        dmu[(size_t)j * d + m] = g1 + (mean_grad ? mean_grad[(size_t)j * d + m] : 0.0);
        dvar[(size_t)j * d + m] = g2;
    }
}

// a5: full posterior covariance  Σ = K** − VᵀV + 1e-18·I  (mean_and_cov, gaussian_process.jl:180-184;
// AbstractGPs cov(post(X*))) from the V slabs the prediction kernel left in its scratch
// (V(n, j) = Vs[(j/BN * Np + n) * BN + j % BN]).  16×16 outputs per workgroup, n staged through LDS.
// Not a hot path (EI never needs it); the diagonal is NOT clipped here (see clip_cov_diag_kernel).
__global__ __launch_bounds__(256) void predict_cov_kernel(const double* __restrict__ Vs, int Np, int BN,
                                                          const double* __restrict__ Csc, int d, int Mp, int M,
                                                          int kern, double amp2, double* __restrict__ cov) {
    __shared__ double Va[64][17], Vb[64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j1 = blockIdx.x * 16 + tx, j2 = blockIdx.y * 16 + ty;
    double acc = 0.0;
    for (int n0 = 0; n0 < Np; n0 += 64) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
            const int nn = idx >> 4, jj = idx & 15;
            const int ja = blockIdx.x * 16 + jj, jb = blockIdx.y * 16 + jj;
            Va[nn][jj] = (ja < M) ? Vs[((size_t)(ja / BN) * Np + n0 + nn) * BN + ja % BN] : 0.0;
            Vb[nn][jj] = (jb < M) ? Vs[((size_t)(jb / BN) * Np + n0 + nn) * BN + jb % BN] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int nn = 0; nn < 64; ++nn) acc = __builtin_fma(Va[nn][tx], Vb[nn][ty], acc);
    }
    if (j1 < M && j2 < M) {
        double r2 = 0.0;
        for (int kd = 0; kd < d; ++kd) {
            const double diff = Csc[(size_t)kd * Mp + j1] - Csc[(size_t)kd * Mp + j2];
            r2 = __builtin_fma(diff, diff, r2);
        }
        cov[(size_t)j2 * M + j1] = amp2 * kappa_r2(kern, r2) - acc + ((j1 == j2) ? PREDICT_JITTER : 0.0);
    }
}

// _clip_var on the diagonal of Σ (gaussian_process.jl:165,182)
__global__ void clip_cov_diag_kernel(double* __restrict__ cov, int M, unsigned long long* __restrict__ bad) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    double v = cov[(size_t)j * M + j];
    if (v >= 0.0) return;
    if (v >= -MAX_NEG_VAR) cov[(size_t)j * M + j] = 0.0;
    else atomicMin(bad, (unsigned long long)j);
}

// first index with var < -MAX_NEG_VAR  (DomainError of _clip_var); bad[0] initialised to LONG_MAX
__global__ void clip_var_kernel(double* __restrict__ var, int M, unsigned long long* __restrict__ bad) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    double v = var[j];
    if (v >= 0.0) return;
    if (v >= -MAX_NEG_VAR) var[j] = 0.0;
    else atomicMin(bad, (unsigned long long)j);
}

// ------------------------------------------------------------------------------------------
// K8 EI·feas epilogue (expected_improvement.jl:68-101,113-114).  mu/var are P×M (row p at p*ldm).
// mode bit0: has best_yet, bit1: constrained.  A candidate whose variance is < -1e-8 in any
// output is poisoned with -Inf (SafeFunction semantics, src/acquisition.jl:21-25).
// The P fitness coefficients / constraints travel in the kernel arguments (no H2D copies) when
// P <= EI_MAXP, otherwise in device arrays.
// ------------------------------------------------------------------------------------------
constexpr int EI_MAXP = 16;
struct EiPar {
    int P, mode;
    double best;
    double coefs[EI_MAXP], ymax[EI_MAXP];
};

__device__ __forceinline__ double ei_value(const double* __restrict__ mu, const double* __restrict__ var, int ldm, int j,
                                           const EiPar& par, const double* __restrict__ coefs_dev,
                                           const double* __restrict__ ymax_dev) {
    const int P = par.P, mode = par.mode;
    if (mode == 0) return 0.0;                              // construct_ei(…, nothing, …, nothing): acq ≡ 0
    double muf = 0.0, vf = 0.0, fp = 1.0;
    bool poison = false;
    for (int p = 0; p < P; ++p) {
        const double cf = (P <= EI_MAXP) ? par.coefs[p] : coefs_dev[p];
        const double ym = (mode & 2) ? ((P <= EI_MAXP) ? par.ymax[p] : ymax_dev[p]) : INFINITY;
        double m = mu[(size_t)p * ldm + j], v = var[(size_t)p * ldm + j];
        if (v < 0.0) {
            if (v >= -MAX_NEG_VAR) v = 0.0;
            else poison = true;
        }
        muf = __builtin_fma(cf, m, muf);
        vf = __builtin_fma(cf * cf, v, vf);
        if ((mode & 2) && !(isinf(ym) && ym > 0.0)) {
            double s = sqrt(v);
            double z = (s == 0.0 && ym == m) ? INFINITY : (ym - m) / s;
            fp *= normcdf_dev(z);
        }
    }
    double acq;
    if (mode & 1) {
        double sf = sqrt(vf);
        double diff = muf - par.best;
        double ei;
        if (diff == 0.0 && sf == 0.0) ei = 0.0;
        else {
            double z = diff / sf;
            ei = diff * normcdf_dev(z) + sf * normpdf_dev(z);
        }
        acq = (mode & 2) ? ei * fp : ei;
    } else {
        acq = fp;
    }
    return poison ? -INFINITY : acq;
}

// BI: acq_sum[j] += acq_s(x_j) for every hyper-parameter sample but the last (the last one is folded
// into acq_epilogue_kernel).
__global__ void ei_accumulate_kernel(const double* __restrict__ mu, const double* __restrict__ var, int ldm, int M,
                                     EiPar par, const double* __restrict__ coefs_dev,
                                     const double* __restrict__ ymax_dev, double* __restrict__ acq_sum) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    acq_sum[j] += ei_value(mu, var, ldm, j, par, coefs_dev, ymax_dev);
}

// K9 arg-max (Julia argmax: first index of the maximum, NaN counts as the largest value).
__device__ __forceinline__ bool better(double a, long ia, double b, long ib) {
    const bool an = a != a, bn = b != b;
    if (an != bn) return an;
    if (!an && a != b) return a > b;
    return ia < ib;
}

// Fused epilogue of one acquisition batch (one 1024-thread workgroup):
//   acq[j] = (acq_sum[j] (previous samples, if any) + acq_S(x_j)) / S, masked to 0 outside the domain
//   (make_safe, expected_improvement.jl:58-65), written back to acq_sum, and its first-index
//   arg-max, written straight into host-pinned memory (res[0] = value, res[1] = index as int64).
constexpr int ACQ_EPI_THREADS = 1024;
__global__ __launch_bounds__(ACQ_EPI_THREADS) void acq_epilogue_kernel(const double* __restrict__ mu,
                                                                       const double* __restrict__ var, int ldm, int M,
                                                                       EiPar par, const double* __restrict__ coefs_dev,
                                                                       const double* __restrict__ ymax_dev,
                                                                       double* __restrict__ acq_sum, int have_prev,
                                                                       double inv_s, const unsigned char* __restrict__ mask,
                                                                       double* __restrict__ res) {
    __shared__ double sv[ACQ_EPI_THREADS];
    __shared__ long si[ACQ_EPI_THREADS];
    constexpr long NONE = 0x7fffffffffffffffL;
    double bv = -INFINITY;
    long bi = NONE;
    for (int j = threadIdx.x; j < M; j += ACQ_EPI_THREADS) {
        double a = ei_value(mu, var, ldm, j, par, coefs_dev, ymax_dev);
        if (have_prev) a += acq_sum[j];
        a *= inv_s;
        if (mask && !mask[j]) a = 0.0;
        acq_sum[j] = a;
        if (bi == NONE || better(a, j, bv, bi)) { bv = a; bi = j; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int st = ACQ_EPI_THREADS / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            const double ov = sv[threadIdx.x + st];
            const long oi = si[threadIdx.x + st];
            if (oi != NONE && (si[threadIdx.x] == NONE || better(ov, oi, sv[threadIdx.x], si[threadIdx.x]))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        res[0] = sv[0];
        reinterpret_cast<long*>(res)[1] = si[0];
    }
}

// Acquisition value and gradient w.r.t. the candidate for one hyper-parameter sample, from the P
// outputs' moments and moment gradients (mu/var: [p][M]; dmu/dvar: [p][j*d + m]) — the chain rule
// through construct_ei (expected_improvement.jl:68-101,113-114):
//   μf = cᵀμ, σf = sqrt(c²ᵀσ²), z = (μf − b)/σf :  ∇EI = Φ(z) ∇μf + φ(z) ∇σf ,  ∇σf = c²ᵀ∇σ² / (2σf)
//   FP = Π_p Φ(t_p), t_p = (ymax_p − μ_p)/s_p, s_p = sqrt(σ²_p):
//        ∇FP = Σ_p (Π_{q≠p} Φ(t_q)) φ(t_p) ∇t_p ,  ∇t_p = −∇μ_p/s_p − (ymax_p − μ_p) ∇σ²_p / (2 s_p³)
//   acq = EI·FP (or EI, or FP, or 0 by mode); outside the domain mask: acq = 0, ∇acq = 0 (make_safe).
// Variances clipped to 0 (or exactly 0) contribute no σ-gradient.
__global__ void ei_grad_kernel(const double* __restrict__ mu, const double* __restrict__ var, const double* __restrict__ dmu,
                               const double* __restrict__ dvar, int M, int d, EiPar par,
                               const double* __restrict__ coefs_dev, const double* __restrict__ ymax_dev,
                               const unsigned char* __restrict__ mask, double* __restrict__ acq,
                               double* __restrict__ dacq) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int P = par.P, mode = par.mode;
    const size_t dm = (size_t)d * M;
    double* gout = dacq + (size_t)j * d;
    if (mode == 0 || (mask && !mask[j])) {
        acq[j] = 0.0;
        for (int m = 0; m < d; ++m) gout[m] = 0.0;
        return;
    }
    // pass 1: scalars
    double muf = 0.0, vf = 0.0, fp = 1.0;
    bool poison = false;
    for (int p = 0; p < P; ++p) {
        const double cf = (P <= EI_MAXP) ? par.coefs[p] : coefs_dev[p];
        const double ym = (mode & 2) ? ((P <= EI_MAXP) ? par.ymax[p] : ymax_dev[p]) : INFINITY;
        double m_ = mu[(size_t)p * M + j], v = var[(size_t)p * M + j];
        if (v < 0.0) {
            if (v >= -MAX_NEG_VAR) v = 0.0;
            else poison = true;
        }
        muf = __builtin_fma(cf, m_, muf);
        vf = __builtin_fma(cf * cf, v, vf);
        if ((mode & 2) && !(isinf(ym) && ym > 0.0)) {
            const double sd = sqrt(v);
            const double t = (sd == 0.0 && ym == m_) ? INFINITY : (ym - m_) / sd;
            fp *= normcdf_dev(t);
        }
    }
    double ei = 0.0, Phi = 0.0, phi_over_2sf = 0.0;
    if (mode & 1) {
        const double sf = sqrt(vf), diff = muf - par.best;
        if (diff == 0.0 && sf == 0.0) ei = 0.0;
        else {
            const double z = diff / sf;
            Phi = normcdf_dev(z);
            const double ph = normpdf_dev(z);
            ei = diff * Phi + sf * ph;
            phi_over_2sf = (sf > 0.0) ? ph / (2.0 * sf) : 0.0;
        }
    }
    double a;
    if (mode & 1) a = (mode & 2) ? ei * fp : ei;
    else a = fp;
    acq[j] = poison ? -INFINITY : a;
    // pass 2: gradient, one coordinate at a time
    for (int m = 0; m < d; ++m) {
        double dmuf = 0.0, dvf = 0.0, dfp = 0.0;
        for (int p = 0; p < P; ++p) {
            const double cf = (P <= EI_MAXP) ? par.coefs[p] : coefs_dev[p];
            const double gm = dmu[(size_t)p * dm + (size_t)j * d + m];
            double v = var[(size_t)p * M + j];
            const bool clipped = !(v > 0.0);
            const double gv = clipped ? 0.0 : dvar[(size_t)p * dm + (size_t)j * d + m];
            dmuf = __builtin_fma(cf, gm, dmuf);
            dvf = __builtin_fma(cf * cf, gv, dvf);
            if (mode & 2) {
                const double ym = (P <= EI_MAXP) ? par.ymax[p] : ymax_dev[p];
                if (!(isinf(ym) && ym > 0.0) && !clipped) {
                    const double m_ = mu[(size_t)p * M + j], sd = sqrt(v);
                    const double t = (ym - m_) / sd;
                    const double dt = -gm / sd - (ym - m_) * gv / (2.0 * sd * v);
                    double others = 1.0;
                    for (int q = 0; q < P; ++q) {
                        if (q == p) continue;
                        const double yq = (P <= EI_MAXP) ? par.ymax[q] : ymax_dev[q];
                        if (isinf(yq) && yq > 0.0) continue;
                        double vq = var[(size_t)q * M + j];
                        if (vq < 0.0) vq = 0.0;
                        const double mq = mu[(size_t)q * M + j], sq = sqrt(vq);
                        const double tq = (sq == 0.0 && yq == mq) ? INFINITY : (yq - mq) / sq;
                        others *= normcdf_dev(tq);
                    }
                    dfp = __builtin_fma(others * normpdf_dev(t), dt, dfp);
                }
            }
        }
        const double dei = Phi * dmuf + phi_over_2sf * dvf;
        double gA;
        if (mode & 1) gA = (mode & 2) ? dei * fp + ei * dfp : dei;
        else gA = dfp;
        gout[m] = poison ? 0.0 : gA;
    }
}

// ------------------------------------------------------------------------------------------
// issue-rate microbenchmark of v_mfma_f64_16x16x4_f64 (16 independent accumulators per wave)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_f64_rate_kernel(int iters, double* __restrict__ sink) {
    v4d acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = mfma_f64(a, b, acc[t]);
    }
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 12345.678) sink[0] = s;
}

}  // namespace boss

// gram_kernels.hpp — point layout helpers, right-hand-side rows, Gram matrices of the three models
// (stationary ARD kernels, gradient observations, Gibbs kernel) and the cross-covariance writers of the latter two.
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

// The same for one posterior whose hyper-parameters travel in the kernel arguments (no staging copy, no event): the kernel
// also deposits 1/λ, α², σ² in the handle's resident parameter block (invlam[0..d), then hyp[0..2)) for the kernels that follow.
constexpr int HYP_ARGS_MAX_D = 32;
struct HypArgs {
    int d;
    double invlam[HYP_ARGS_MAX_D];
    double hyp[2];
};
// ... and, in the same launch, the right-hand-side row block of rhs_rows_kernel (row Np = (y - m)^T, the 31 rows below zero) and the
// cleared failed-pivot flag: the update's preparation is one kernel.
__global__ void scale_points_args_kernel(HypArgs par, const double* __restrict__ Xraw, double* __restrict__ Xsc,
                                         double* __restrict__ par_dev, int ldp, double* __restrict__ A, int ld, int N,
                                         const double* __restrict__ y, const double* __restrict__ mean, int* __restrict__ info,
                                         double* __restrict__ inv16_fill) {
    // inv16_fill (or null): the 16·Np doubles of the diagonal tiles' inverses, set to the all-ones pattern the resident strips
    // recognise as "not written yet" (chain.hpp) — instead of a memset launch in front of this kernel
    if (blockIdx.x == 0 && threadIdx.x < par.d + 2)
        par_dev[threadIdx.x] = threadIdx.x < par.d ? par.invlam[threadIdx.x] : par.hyp[threadIdx.x - par.d];
    if (blockIdx.x == 0 && threadIdx.x == 0) info[0] = 0;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ldp) return;
    if (inv16_fill) {
        const v2d ones = v2d{__longlong_as_double(-1LL), __longlong_as_double(-1LL)};
#pragma unroll
        for (int r = 0; r < 8; ++r) *reinterpret_cast<v2d*>(inv16_fill + (size_t)(2 * r) * ldp + 2 * (size_t)j) = ones;   // 8 passes × 2·Np doubles = all 16·Np, 16 B per thread per pass
    }
    for (int k = 0; k < par.d; ++k) Xsc[(size_t)k * ldp + j] = Xraw[(size_t)k * ldp + j] * par.invlam[k];
    double* col = A + (size_t)j * ld + ldp;                   // ldp = Np: the δ^T row block starts at row Np
    col[0] = (j < N) ? (y[j] - mean[j]) : 0.0;
#pragma unroll 1
    for (int r = 1; r < 32; ++r) col[r] = 0.0;
}

// RHS row block: row Np = (y - m)^T for j < N, everything else in rows Np..Np+31 zero.
// col0: first column to (re)write — 0 for a full fit, the first column of the re-factorised block
// row for boss_gp_append (the z entries of the columns before it are final and must survive).
__global__ void rhs_rows_kernel(double* __restrict__ Abase, int ld, size_t bstride, int N, int Np,
                                const double* __restrict__ y, const double* __restrict__ mean, size_t mean_bstride,
                                int col0, int* __restrict__ info) {
    // info (or null): the failed-pivot flag of matrix b, cleared here so that an update needs no separate memset
    const int b = blockIdx.z;
    if (info && blockIdx.x == 0 && threadIdx.x == 0) info[b] = 0;
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

// One entry of the augmented matrix AND its derivatives w.r.t. the lengthscales (dl[m] = ∂entry/∂λ_m), same arguments and same
// perturbation rule as aug_entry (the value block at the points as given, the derivative blocks at xj + 1e-8 when xi ≈ xj).  With
// u = xi − xj, s = u ⊘ λ², w_m = −u_m²/λ_m³ and the profiles h = κ'/r, g = h'/r, q = g'/r:
//   value            α² κ                 ∂/∂λ_m = α² h w_m
//   (0, m') / (l, 0) ∓α² h s              ∂/∂λ_m = ∓α² (g w_m s − 2 h s δ/λ_m)
//   (l, l')          −α² (g s_l s_l' + h δ_ll'/λ_l²)
//                                         ∂/∂λ_m = −α² (q w_m s_l s_l' − 2 g s_l s_l' (δ_lm + δ_l'm)/λ_m + g w_m δ_ll'/λ_l² − 2 h δ_ll' δ_lm/λ_l³)
// (checked against finite differences of the likelihood in tests/)
__device__ __forceinline__ double aug_entry_dlam(int kern, double amp2, int d, const double* il, const double* xi, int si,
                                                 const double* xj, int sj, int lr, int lc, double* dl) {
    double du2 = 0.0, ni = 0.0, nj = 0.0, r2 = 0.0;
    for (int k = 0; k < d; ++k) {
        const double a = xi[k * si], b = xj[k * sj], u = a - b, t = u * il[k];
        du2 = __builtin_fma(u, u, du2);
        ni = __builtin_fma(a, a, ni);
        nj = __builtin_fma(b, b, nj);
        r2 = __builtin_fma(t, t, r2);
    }
    if (lr == 0 && lc == 0) {
        const double h0 = amp2 * kappa_prime_over_r_r2(kern, r2);
        for (int k = 0; k < d; ++k) {
            const double u = xi[k * si] - xj[k * sj];
            dl[k] = -h0 * u * u * il[k] * il[k] * il[k];
        }
        return amp2 * kappa_r2(kern, r2);
    }
    double eps = 0.0;
    if (du2 <= ISAPPROX_RTOL2 * fmax(ni, nj)) {
        eps = MIN_PARAM_VALUE;
        r2 = 0.0;
        for (int k = 0; k < d; ++k) {
            const double t = (xi[k * si] - (xj[k * sj] + eps)) * il[k];
            r2 = __builtin_fma(t, t, r2);
        }
    }
    const double h = kappa_prime_over_r_r2(kern, r2), g = kappa_second_r2(kern, r2);
    if (lr == 0 || lc == 0) {
        const int m = (lr == 0 ? lc : lr) - 1;
        const double sgn = (lr == 0) ? -amp2 : amp2;
        const double s = (xi[m * si] - (xj[m * sj] + eps)) * il[m] * il[m];
        for (int k = 0; k < d; ++k) {
            const double u = xi[k * si] - (xj[k * sj] + eps), w = -u * u * il[k] * il[k] * il[k];
            dl[k] = sgn * g * w * s;
        }
        dl[m] += sgn * (-2.0 * h * s * il[m]);
        return sgn * h * s;
    }
    const int l = lr - 1, m = lc - 1;
    const double q = kappa_third_r2(kern, r2);
    const double sl = (xi[l * si] - (xj[l * sj] + eps)) * il[l] * il[l];
    const double sm = (xi[m * si] - (xj[m * sj] + eps)) * il[m] * il[m];
    const double diag = (l == m) ? il[l] * il[l] : 0.0;
    for (int k = 0; k < d; ++k) {
        const double u = xi[k * si] - (xj[k * sj] + eps), w = -u * u * il[k] * il[k] * il[k];
        dl[k] = -amp2 * w * __builtin_fma(q * sl, sm, g * diag);
    }
    dl[l] += 2.0 * amp2 * g * sl * sm * il[l];
    dl[m] += 2.0 * amp2 * g * sl * sm * il[m];
    if (l == m) dl[l] += 2.0 * amp2 * h * il[l] * il[l] * il[l];
    return -amp2 * __builtin_fma(g * sl, sm, h * diag);
}

// Likelihood gradient of the gradient-observation model, contraction part: lower 64×64 tiles like aug_gram_kernel,
//   out[tile][0..d-1] = Σ w G_ab ∂K_ab/∂λ_m ,  out[tile][d] = Σ w G_ab K⁰_ab (noise-free entries) ,
//   out[tile][d+1] = Σ_{a value row} G_aa ,  out[tile][d+2] = Σ_{a derivative row} G_aa ,
// w = 1 below the diagonal (the pair counts twice in ½ Σ_ab), ½ on it;  G = a aᵀ − K⁻¹ with a summed from its nch partials.
__global__ __launch_bounds__(256) void aug_llgrad_tile_kernel(const double* __restrict__ Xraw, int ldx, int d, int n, int N, int Np, int kern,
                                                              const double* __restrict__ hyp, const double* __restrict__ invlam,
                                                              const double* __restrict__ Kinv, int ldk, const double* __restrict__ apart,
                                                              int nch, double* __restrict__ out) {
    __shared__ double xa[AUG_MAX_D][64], xb[AUG_MAX_D][64], il[AUG_MAX_D], ab[64], red[256];
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
        if (tid >= 64) {
            double s = 0.0;
            for (int ch = 0; ch < nch; ++ch) s += apart[(size_t)ch * Np + obs];
            ab[c] = s;
        }
    }
    if (tid >= 128 && tid < 128 + d) il[tid - 128] = invlam[tid - 128];
    __syncthreads();
    const double amp2 = hyp[0];
    const int r = tid & 63, cg = tid >> 6, a = bi * 64 + r;
    double aa = 0.0;
    for (int ch = 0; ch < nch; ++ch) aa += apart[(size_t)ch * Np + a];
    double S[AUG_MAX_D + 3];
#pragma unroll
    for (int m = 0; m < AUG_MAX_D + 3; ++m) S[m] = 0.0;
#pragma unroll 1
    for (int c = 0; c < 16; ++c) {
        const int cc = cg * 16 + c, b = bj * 64 + cc;
        if (a < b || a >= N || b >= N) continue;
        const double G = aa * ab[cc] - Kinv[(size_t)b * ldk + a];
        double dl[AUG_MAX_D];
        const double v = aug_entry_dlam(kern, amp2, d, il, &xb[0][cc], 64, &xa[0][r], 64, lb[cc], la[r], dl);
        const double w = (a == b) ? 0.5 * G : G;
#pragma unroll
        for (int m = 0; m < AUG_MAX_D; ++m)
            if (m < d) S[m] = __builtin_fma(w, dl[m], S[m]);
        S[AUG_MAX_D] = __builtin_fma(w, v, S[AUG_MAX_D]);
        if (a == b) S[AUG_MAX_D + (la[r] == 0 ? 1 : 2)] += G;
    }
    for (int m = 0; m < d + 3; ++m) {
        double v = 0.0;
#pragma unroll
        for (int mm = 0; mm < AUG_MAX_D + 3; ++mm)
            if (mm == (m < d ? m : AUG_MAX_D + (m - d))) v = S[mm];
        __syncthreads();
        red[tid] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) out[(size_t)t * (d + 3) + m] = red[0];
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

// The same for 32-wide tiles with the lanes along the candidate columns: a wave writes two whole 256-byte rows of the
// tile per store (one row per thread makes every store touch 64 cache lines).  Rows are staged 8 dimensions at a time.
constexpr int GIBBS_KSTAR_MAX_D = 48;                         // LDS: (2·d·32 + 32 + 2·8·256) doubles
__global__ __launch_bounds__(256) void gibbs_kstar_cols_kernel(const double* __restrict__ X, const double* __restrict__ Lam,
                                                               const double* __restrict__ amp, int d, int N, int Np,
                                                               const double* __restrict__ C, const double* __restrict__ Clam,
                                                               const double* __restrict__ Camp, int Mp, double* __restrict__ out) {
    extern __shared__ double sm[];                           // cx[d][32] | cl[d][32] | ca[32] | xs[8][256] | ls[8][256]
    double* cx = sm;
    double* cl = cx + d * 32;
    double* ca = cl + d * 32;
    double* xs = ca + 32;
    double* ls = xs + 8 * 256;
    const int tid = threadIdx.x, c0 = blockIdx.y * 32, col = tid & 31, rg = tid >> 5, rbase = blockIdx.x * 256;
    out += (size_t)blockIdx.y * Np * 32;
    for (int idx = tid; idx < d * 32; idx += 256) {
        cx[idx] = C[(size_t)(idx >> 5) * Mp + c0 + (idx & 31)];
        cl[idx] = Clam[(size_t)(idx >> 5) * Mp + c0 + (idx & 31)];
    }
    if (tid < 32) ca[tid] = Camp[c0 + tid];
    double pr[32], es[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        pr[i] = 1.0;
        es[i] = 0.0;
    }
    for (int k0 = 0; k0 < d; k0 += 8) {
        const int kc = (d - k0 < 8) ? d - k0 : 8;
        __syncthreads();
        for (int idx = tid; idx < kc * 256; idx += 256) {
            xs[idx] = X[(size_t)(k0 + (idx >> 8)) * Np + rbase + (idx & 255)];
            ls[idx] = Lam[(size_t)(k0 + (idx >> 8)) * Np + rbase + (idx & 255)];
        }
        __syncthreads();
        for (int kk = 0; kk < kc; ++kk) {
            const double xc = cx[(k0 + kk) * 32 + col], lc = cl[(k0 + kk) * 32 + col];
#pragma unroll
            for (int i = 0; i < 32; ++i) gibbs_dim(xc, lc, xs[kk * 256 + rg + 8 * i], ls[kk * 256 + rg + 8 * i], pr[i], es[i]);
        }
    }
    const double ac = ca[col];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int row = rbase + rg + 8 * i;
        const double am = 0.5 * (amp[row] + ac);
        out[(size_t)row * 32 + col] = (row < N) ? am * am * sqrt(pr[i]) * exp(-es[i]) : 0.0;
    }
}

// σ²(x*) = k(x*,x*) − Σv² + 1e-18 with k(x*,x*) = α(x*)²; the substitution kernels left −Σv² in var.
__global__ void gibbs_var_kernel(double* __restrict__ var, const double* __restrict__ Camp, int M) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < M) var[j] = __builtin_fma(Camp[j], Camp[j], var[j]) + PREDICT_JITTER;
}

}  // namespace boss

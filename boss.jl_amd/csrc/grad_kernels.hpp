// grad_kernels.hpp — analytic gradients w.r.t. the candidates (adjoint substitution, accumulation), tracked
// candidates, and the gradient of the log marginal likelihood (triangular inverse by recursive doubling, K⁻¹, tile contraction).
#pragma once
#include "predict_kernels.hpp"

namespace boss {

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
    GU::template run<-1, (PRED_RB / 4) % GU::D == 0>(LT + (size_t)r0 + (size_t)ib * PRED_RB * ldt, ldt, V + (size_t)ib * PRED_RB * 32, 32, PRED_RB, acc);
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
// ∇μ, ∇σ² of a gradient-observation posterior (GradientGaussianProcess, src/models/gradient_gp.jl:334-361 under the ForwardDiff of
// OptimizationAM, src/acquisition_maximizers/optimization.jl:36,89-118).  Rows of the n(1+d) system: (l, point i), l = 0 the value.
// With t = (x* − x_i) ⊘ λ², h = κ'(r)/r, g = h'(r)/r:
//     ∇_{x*} k*_(i,0) = α² h t ,      ∇_{x*} k*_(i,l) = −α² (g t_l t + h e_l / λ_l²)          (l >= 1)
// so per point   ∇μ += α² [ t (h a_i0 − g Σ_l a_il t_l) − h (a_i· ⊘ λ²) ]   and the same with w for −½ ∇σ².
// The derivative rows are evaluated at x_i + 1e-8 when x* ≈ x_i (aug_entry, _build_cross_cov :233).
// One workgroup per 32-candidate slab and row split: lanes along the candidates, eight point subsets; the a entries come from avec
// (row l·n + i), the w entries from the W slabs.  part (gridDim.y > 1): [tile·gridDim.y + y][2 d][32], summed by aug_grad_finalize_kernel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aug_grad_accum_kernel(const double* __restrict__ Wslabs, const double* __restrict__ avec, int Np,
                                                             int n, const double* __restrict__ Xraw, int ldx,
                                                             const double* __restrict__ Craw, int d, int Mp, int M, int kern,
                                                             double amp2, const double* __restrict__ invlam,
                                                             double* __restrict__ dmu, double* __restrict__ dvar,
                                                             double* __restrict__ part) {
    constexpr int BN = 32, DM = AUG_MAX_D;
    __shared__ double red[8][2 * DM][BN];
    const int tid = threadIdx.x, c = tid & 31, rs = tid >> 5;
    const int j = blockIdx.x * BN + c;
    const double* W = Wslabs + (size_t)blockIdx.x * Np * BN + c;
    double xc[DM], il2[DM], il[DM], G1[DM], G2[DM];
    double nc = 0.0;
#pragma unroll
    for (int m = 0; m < DM; ++m) {
        xc[m] = m < d ? Craw[(size_t)m * Mp + j] : 0.0;
        il[m] = m < d ? invlam[m] : 0.0;
        il2[m] = il[m] * il[m];
        nc = __builtin_fma(xc[m], xc[m], nc);
        G1[m] = G2[m] = 0.0;
    }
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int ibeg = blockIdx.y * per, iend = min(n, ibeg + per);
    for (int i = ibeg + rs; i < iend; i += 8) {
        double t[DM], du2 = 0.0, ni = 0.0, r2 = 0.0;
#pragma unroll
        for (int m = 0; m < DM; ++m) {
            const double x = m < d ? Xraw[(size_t)m * ldx + i] : 0.0;
            const double u = xc[m] - x;
            du2 = __builtin_fma(u, u, du2);
            ni = __builtin_fma(x, x, ni);
            t[m] = u;
        }
#pragma unroll
        for (int m = 0; m < DM; ++m) r2 = __builtin_fma(t[m] * il[m], t[m] * il[m], r2);
        const double a0 = avec[i], w0 = W[(size_t)i * BN];
        const double h0 = kappa_prime_over_r_r2(kern, r2);   // the value row: at the points as given
        double rr2 = r2;
        if (du2 <= ISAPPROX_RTOL2 * fmax(nc, ni)) {          // x* ≈ x_i: the derivative rows at x_i + 1e-8
            rr2 = 0.0;
#pragma unroll
            for (int m = 0; m < DM; ++m) {
                if (m < d) t[m] -= MIN_PARAM_VALUE;
                rr2 = __builtin_fma(t[m] * il[m], t[m] * il[m], rr2);
            }
        }
        const double h = kappa_prime_over_r_r2(kern, rr2), g = kappa_second_r2(kern, rr2);
        double sa = 0.0, sw = 0.0, al[DM], wl[DM];
#pragma unroll
        for (int m = 0; m < DM; ++m) {
            al[m] = m < d ? avec[(size_t)(m + 1) * n + i] : 0.0;
            wl[m] = m < d ? W[((size_t)(m + 1) * n + i) * BN] : 0.0;
            const double tm = t[m] * il2[m];
            sa = __builtin_fma(al[m], tm, sa);
            sw = __builtin_fma(wl[m], tm, sw);
        }
        // (the value row's t is the unshifted one: its shift only matters where t ≈ 0 anyway)
        const double ka = -g * sa, kw = -g * sw;
#pragma unroll
        for (int m = 0; m < DM; ++m) {
            const double tm = t[m] * il2[m];
            const double tm0 = (du2 <= ISAPPROX_RTOL2 * fmax(nc, ni)) ? (t[m] + (m < d ? MIN_PARAM_VALUE : 0.0)) * il2[m] : tm;
            G1[m] += tm0 * h0 * a0 + tm * ka - h * al[m] * il2[m];
            G2[m] += tm0 * h0 * w0 + tm * kw - h * wl[m] * il2[m];
        }
    }
#pragma unroll
    for (int m = 0; m < DM; ++m) {
        red[rs][2 * m][c] = G1[m];
        red[rs][2 * m + 1][c] = G2[m];
    }
    __syncthreads();
    for (int slot = rs; slot < 2 * d; slot += 8) {
        double v = 0.0;
        for (int k = 0; k < 8; ++k) v += red[k][slot][c];
        const int m = slot >> 1;
        if (gridDim.y > 1) part[(((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 2 * DM + slot) * BN + c] = v;
        else if (j < M) {
            if (slot & 1) dvar[(size_t)j * d + m] = -2.0 * amp2 * v;
            else dmu[(size_t)j * d + m] = amp2 * v;
        }
    }
}
__global__ __launch_bounds__(32) void aug_grad_finalize_kernel(const double* __restrict__ part, int rsplit, int d, int M, double amp2,
                                                               double* __restrict__ dmu, double* __restrict__ dvar) {
    constexpr int BN = 32, DM = AUG_MAX_D;
    const int c = threadIdx.x, j = blockIdx.x * BN + c;
    if (j >= M) return;
    for (int slot = 0; slot < 2 * d; ++slot) {
        double v = 0.0;
        for (int y = 0; y < rsplit; ++y) v += part[(((size_t)blockIdx.x * rsplit + y) * 2 * DM + slot) * BN + c];
        const int m = slot >> 1;
        if (slot & 1) dvar[(size_t)j * d + m] = -2.0 * amp2 * v;
        else dmu[(size_t)j * d + m] = amp2 * v;
    }
}

// ------------------------------------------------------------------------------------------
// Candidate gradients of a nonstationary posterior (Gibbs kernel, src/models/nonstationary_gp/nonstationary_gp.jl:61-107,153-196):
// the candidate enters k_i = k(x_i, x*) directly and through λ(x*), α(x*).  With q_l = λ_il² + λ*_l², Δ_l = x_il − x*_l:
//     e_l = ∂ln k_i/∂x*_l = 2 Δ_l/q_l ,   c_l = ∂ln k_i/∂λ*_l = ½ (1/λ*_l − 2 λ*_l/q_l) + 2 λ*_l Δ_l²/q_l² ,   s = ∂ln k_i/∂α* = 2/(α_i + α*)
// The kernel leaves, per candidate, the sums Σ_i a_i k_i (s, e, c) and Σ_i w_i k_i (s, e, c) — sums[slot][Mp], slot = which·(2d+1) +
// {0: s, 1+l: e_l, 1+d+l: c_l} — and the host folds the caller's Jacobians ∂λ/∂x, ∂α/∂x in (boss_ngp_predict_grad).
// One workgroup per 32-candidate slab: lanes along the candidates, eight row subsets.  d <= 16.
// ------------------------------------------------------------------------------------------
constexpr int GIBBS_GRAD_MAX_D = 16;
__global__ __launch_bounds__(256) void gibbs_grad_accum_kernel(const double* __restrict__ Wslabs, const double* __restrict__ avec, int Np,
                                                               int N, const double* __restrict__ X, const double* __restrict__ Lam,
                                                               const double* __restrict__ amp, const double* __restrict__ C,
                                                               const double* __restrict__ Clam, const double* __restrict__ Camp, int d,
                                                               int Mp, const unsigned char* __restrict__ discrete,
                                                               double* __restrict__ sums) {
    constexpr int BN = 32, DM = GIBBS_GRAD_MAX_D;
    __shared__ double red[8][BN];
    const int tid = threadIdx.x, c = tid & 31, rs = tid >> 5;
    const int j = blockIdx.x * BN + c;
    const double* W = Wslabs + (size_t)blockIdx.x * Np * BN + c;
    double xc[DM], lc[DM], Ea[DM], Ca[DM], Ew[DM], Cw[DM], Sa = 0.0, Sw = 0.0;
#pragma unroll
    for (int m = 0; m < DM; ++m) {
        xc[m] = m < d ? C[(size_t)m * Mp + j] : 0.0;
        lc[m] = m < d ? Clam[(size_t)m * Mp + j] : 1.0;
        Ea[m] = Ca[m] = Ew[m] = Cw[m] = 0.0;
    }
    const double ac = Camp[j];
    for (int i = rs; i < N; i += 8) {
        double prod = 1.0, esum = 0.0, e[DM], cl[DM];
#pragma unroll
        for (int m = 0; m < DM; ++m) {
            if (m < d) {
                const double x = X[(size_t)m * Np + i], l = Lam[(size_t)m * Np + i];
                const double rq = rcp_refined(__builtin_fma(l, l, lc[m] * lc[m]));
                const double df = x - xc[m];
                prod *= 2.0 * l * lc[m] * rq;
                esum = __builtin_fma(df * df, rq, esum);
                e[m] = (discrete && discrete[m]) ? 0.0 : 2.0 * df * rq;
                cl[m] = 0.5 * (1.0 / lc[m] - 2.0 * lc[m] * rq) + 2.0 * lc[m] * df * df * rq * rq;
            } else {
                e[m] = cl[m] = 0.0;
            }
        }
        const double ai = amp[i], am = 0.5 * (ai + ac);
        const double k = am * am * sqrt(prod) * exp(-esum);
        const double wa = avec[i] * k, ww = W[(size_t)i * BN] * k, s = 1.0 / am;      // 2/(α_i + α*)
        Sa = __builtin_fma(wa, s, Sa);
        Sw = __builtin_fma(ww, s, Sw);
#pragma unroll
        for (int m = 0; m < DM; ++m) {
            Ea[m] = __builtin_fma(wa, e[m], Ea[m]);
            Ca[m] = __builtin_fma(wa, cl[m], Ca[m]);
            Ew[m] = __builtin_fma(ww, e[m], Ew[m]);
            Cw[m] = __builtin_fma(ww, cl[m], Cw[m]);
        }
    }
    // the eight row subsets in order, slot by slot
    const int nslot = 2 * d + 1;
    auto reduce_store = [&](int slot, double v) {
        __syncthreads();
        red[rs][c] = v;
        __syncthreads();
        if (rs == 0) {
            double t = red[0][c];
#pragma unroll
            for (int q = 1; q < 8; ++q) t += red[q][c];
            sums[(size_t)slot * Mp + j] = t;
        }
    };
    reduce_store(0, Sa);
    reduce_store(nslot, Sw);
#pragma unroll
    for (int m = 0; m < DM; ++m) {
        if (m < d) {                                         // (uniform)
            reduce_store(1 + m, Ea[m]);
            reduce_store(1 + d + m, Ca[m]);
            reduce_store(nslot + 1 + m, Ew[m]);
            reduce_store(nslot + 1 + d + m, Cw[m]);
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
                                                        double* __restrict__ U, int ldu, size_t zD2, size_t zW) {
    // blockIdx.z: matrix of a batch (strides zD2 for the diagonal inverses, zW for both work matrices)
    const int b = blockIdx.y, c = blockIdx.x, r = threadIdx.x;
    Dinv2 += (size_t)blockIdx.z * zD2;
    Lw += (size_t)blockIdx.z * zW;
    U += (size_t)blockIdx.z * zW;
    const double v = (r >= c) ? Dinv2[(size_t)b * PRED_RB * PRED_RB + r + (size_t)c * PRED_RB] : 0.0;   // upper half holds scratch
    const size_t o = (size_t)b * PRED_RB;
    Lw[(o + r) + (o + c) * ldw] = v;
    U[(o + c) + (o + r) * ldu] = v;
}

template <class SG, int PHASE>
__global__ __launch_bounds__(256, 2) void linv_level_kernel(const double* __restrict__ Afac, int ld, double* __restrict__ Lw,
                                                            int ldw, double* __restrict__ U, int ldu, int Np, int s, size_t zA,
                                                            size_t zW) {
    Afac += (size_t)blockIdx.z * zA;                         // blockIdx.z: matrix of a batch
    Lw += (size_t)blockIdx.z * zW;
    U += (size_t)blockIdx.z * zW;
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
                                                           double* __restrict__ Kinv, int ldk, size_t zW) {
    LinvT += (size_t)blockIdx.z * zW;                        // blockIdx.z: matrix of a batch
    Kinv += (size_t)blockIdx.z * zW;
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
                                                           double* __restrict__ partial, size_t zW, size_t zA, size_t zP) {
    LinvT += (size_t)blockIdx.z * zW;                        // blockIdx.z: matrix of a batch
    A += (size_t)blockIdx.z * zA;
    partial += (size_t)blockIdx.z * zP;
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
                                                          const double* __restrict__ amp2p, int amp2_stride,
                                                          const double* __restrict__ Kinv, int ldk,
                                                          const double* __restrict__ apart, int nch,
                                                          double* __restrict__ out, size_t zX, size_t zW, size_t zP, size_t zO) {
    // blockIdx.z: matrix of a batch; α² of that matrix at amp2p[z · amp2_stride]
    const double amp2 = amp2p[(size_t)blockIdx.z * amp2_stride];
    Xsc += (size_t)blockIdx.z * zX;
    Kinv += (size_t)blockIdx.z * zW;
    apart += (size_t)blockIdx.z * zP;
    out += (size_t)blockIdx.z * zO;
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
                                                            double* __restrict__ out, size_t zP, size_t zO) {
    parts += (size_t)blockIdx.z * zP;                        // blockIdx.z: matrix of a batch
    out += (size_t)blockIdx.z * zO;
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

// ------------------------------------------------------------------------------------------
// Likelihood gradient of the nonstationary model w.r.t. the latent models' values at the training points (boss_ngp_loglike_grad):
//   ∂ℓ/∂λ_il = Σ_j G_ij K⁰_ij [½ (1/λ_il − 2 λ_il/q_l) + 2 λ_il Δ_l²/q_l²] ,  ∂ℓ/∂α_i = Σ_j G_ij K⁰_ij 2/(α_i + α_j) ,
//   ∂ℓ/∂σ_i = σ_i G_ii ,  ∂ℓ/∂m_i = a_i ,     G = a aᵀ − K⁻¹,  q_l = λ_il² + λ_jl²,  Δ_l = x_il − x_jl.
// K⁻¹ arrives with its lower triangle valid (kinv_syrk_kernel); mirror_lower_kernel fills the upper one so that row i of the
// symmetric matrix is read as column i (coalesced over the 64 rows of a workgroup).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mirror_lower_kernel(double* __restrict__ A, int ld, int Np) {
    __shared__ double t[64][65];
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj >= bi) return;                                    // (strictly lower tiles are copied into their mirror images; diagonal tiles: below)
    const int r = threadIdx.x & 63, cg = threadIdx.x >> 6;
    for (int c = cg; c < 64; c += 4) t[c][r] = A[(size_t)(bj * 64 + c) * ld + bi * 64 + r];   // element (row bi·64 + r, column bj·64 + c)
    __syncthreads();
    for (int c = cg; c < 64; c += 4) A[(size_t)(bi * 64 + c) * ld + bj * 64 + r] = t[r][c];   // element (row bj·64 + r, column bi·64 + c) = (bi·64 + c, bj·64 + r)
}
__global__ __launch_bounds__(256) void mirror_diag_kernel(double* __restrict__ A, int ld) {
    const int b = blockIdx.x, r = threadIdx.x & 63, cg = threadIdx.x >> 6;
    for (int c = cg; c < 64; c += 4)
        if (c > r) A[(size_t)(b * 64 + c) * ld + b * 64 + r] = A[(size_t)(b * 64 + r) * ld + b * 64 + c];
}
// grid (Np/64, JS): rows bi·64 .. +63, columns of split js; thread (r, cg) takes the columns j ≡ cg (mod 4) of the split.
// part[(js·4 + cg)][v][Np]: v = 0..d-1 the λ sums, d the α sum.
__global__ __launch_bounds__(256) void gibbs_llgrad_kernel(const double* __restrict__ X, const double* __restrict__ Lam,
                                                           const double* __restrict__ amp, int d, int N, int Np,
                                                           const double* __restrict__ Kinv, int ldk, const double* __restrict__ apart,
                                                           int nch, double* __restrict__ part) {
    constexpr int DM = GIBBS_GRAD_MAX_D;
    __shared__ double xj[DM][64], lj[DM][64], aj[64], mj[64];
    const int tid = threadIdx.x, r = tid & 63, cg = tid >> 6;
    const int i = blockIdx.x * 64 + r, js = blockIdx.y, JS = gridDim.y;
    const int ntile = Np / 64, t0 = (int)((long long)ntile * js / JS), t1 = (int)((long long)ntile * (js + 1) / JS);
    double xi[DM], li[DM], S[DM + 1];
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        xi[k] = k < d ? X[(size_t)k * Np + i] : 0.0;
        li[k] = k < d ? Lam[(size_t)k * Np + i] : 1.0;
        S[k] = 0.0;
    }
    S[DM] = 0.0;
    double ai = 0.0;
    for (int c = 0; c < nch; ++c) ai += apart[(size_t)c * Np + i];
    const double ami = amp[i];
    for (int tj = t0; tj < t1; ++tj) {
        __syncthreads();
        for (int idx = tid; idx < d * 64; idx += 256) {
            xj[idx >> 6][idx & 63] = X[(size_t)(idx >> 6) * Np + tj * 64 + (idx & 63)];
            lj[idx >> 6][idx & 63] = Lam[(size_t)(idx >> 6) * Np + tj * 64 + (idx & 63)];
        }
        if (tid < 64) {
            double s = 0.0;
            for (int c = 0; c < nch; ++c) s += apart[(size_t)c * Np + tj * 64 + tid];
            aj[tid] = s;
            mj[tid] = amp[tj * 64 + tid];
        }
        __syncthreads();
        for (int c = cg; c < 64; c += 4) {
            const int j = tj * 64 + c;
            if (i >= N || j >= N) continue;
            double pr = 1.0, es = 0.0, D[DM];
#pragma unroll
            for (int k = 0; k < DM; ++k) {
                if (k < d) {
                    const double lx = li[k], ly = lj[k][c];
                    const double q = rcp_refined(__builtin_fma(lx, lx, ly * ly));
                    const double df = xi[k] - xj[k][c];
                    pr *= 2.0 * lx * ly * q;
                    es = __builtin_fma(df * df, q, es);
                    D[k] = 0.5 * (1.0 / lx - 2.0 * lx * q) + 2.0 * lx * df * df * q * q;
                } else {
                    D[k] = 0.0;
                }
            }
            const double am = 0.5 * (ami + mj[c]);
            const double w = (ai * aj[c] - Kinv[(size_t)j * ldk + i]) * am * am * sqrt(pr) * exp(-es);
#pragma unroll
            for (int k = 0; k < DM; ++k) S[k] = __builtin_fma(w, D[k], S[k]);
            S[DM] = __builtin_fma(w, 1.0 / am, S[DM]);               // 2/(α_i + α_j)
        }
    }
    double* P = part + (size_t)(js * 4 + cg) * (d + 1) * Np;
#pragma unroll
    for (int k = 0; k < DM; ++k)
        if (k < d) P[(size_t)k * Np + i] = S[k];
    P[(size_t)d * Np + i] = S[DM];
}
// out: dlam [d][Np] | damp [Np] | dnoise [Np] | dmean [Np], the parts summed in slot order
__global__ __launch_bounds__(256) void gibbs_llgrad_reduce_kernel(const double* __restrict__ part, int nparts, int d, int N, int Np,
                                                                  const double* __restrict__ noise, const double* __restrict__ Kinv, int ldk,
                                                                  const double* __restrict__ apart, int nch, double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    for (int v = 0; v <= d; ++v) {
        double s = 0.0;
        for (int p = 0; p < nparts; ++p) s += part[((size_t)p * (d + 1) + v) * Np + i];
        out[(size_t)v * Np + i] = i < N ? s : 0.0;
    }
    double ai = 0.0;
    for (int c = 0; c < nch; ++c) ai += apart[(size_t)c * Np + i];
    out[(size_t)(d + 1) * Np + i] = i < N ? noise[i] * (ai * ai - Kinv[(size_t)i * ldk + i]) : 0.0;
    out[(size_t)(d + 2) * Np + i] = i < N ? ai : 0.0;
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

}  // namespace boss

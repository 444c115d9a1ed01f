// predict_kernels.hpp — posterior prediction: the fused kernel (K* tile → blocked substitution → Σv², v·z), the
// step-by-step path for few candidates, the resident-inverse paths (one-pass GEMV, inverse GEMMs, rank-one append)
// and the diagonal-block inverses they share.
#pragma once
#include "gram_kernels.hpp"

namespace boss {

// ------------------------------------------------------------------------------------------
// K4-K7 fused prediction.  One workgroup owns BN candidates and walks the row blocks of L:
//     R_i = K*_i − Σ_{j<i} L_ij V_j          (MFMA GEMM, V_j re-read from its own scratch slab)
//     V_i = Dinv_i · R_i                      (MFMA GEMM, R_i resident in LDS)
//     ss += colsum(V_i²) ,  mz += V_i^T z_i
// which is the blocked form of  V = C.U' \ K*  (AbstractGPs var(post(X*))), with
// μ − m(X*) = K*^T a = V^T z  accumulated in the same pass.  K* never touches HBM.
// ------------------------------------------------------------------------------------------
constexpr int PRED_CS_MAX_D = 16;                   // x_dim up to which the workgroup's scaled candidates are staged in LDS
template <class G>
struct PredictLds {
    static constexpr int LDR = G::BN + 16;
    static constexpr int PART = 2 * G::TN * 4;       // per-thread Σv², v·z partials, parked in LDS between row blocks
    static constexpr int BYTES = (G::BM * LDR + 2 * G::WR * G::BN + G::NTHREADS * PART + PRED_CS_MAX_D * G::BN) * 8;
};

// G = GemmDirect<WR,1,TM,TN,D> with RB = WR·TM·16 ∈ {128, 256}: WR waves stacked along the RB rows of a
// substitution step, BN = 16·TN candidates; Dinv holds the dense inverses of the RB×RB diagonal blocks.
// Both GEMMs stream their A operand (L row block / Dinv_i) straight from L2 through a register
// ring; GEMM1's B operand is the workgroup's own V slab (global, candidate-contiguous), GEMM2's
// B operand is the R tile in LDS.  Three barriers per row block, none inside the GEMMs.
// PRE: the right-hand side K* is not evaluated here but was written to the workgroup's V slab beforehand
// (gradient-observation posteriors, aug_kstar_kernel); block ib's rows are consumed before V_ib overwrites them.
// predict_body: the work of one workgroup = candidate tile `tile` (BN candidates) of one posterior, V = its own slab.
template <class G, bool PRE>
__device__ __forceinline__ void predict_body(const double* __restrict__ A, int ld, int Np, int N,
                                             const double* __restrict__ Dinv,
                                             const double* __restrict__ Xsc,
                                             const double* __restrict__ Csc, int d, int Mp, int kern,
                                             double amp2, double* __restrict__ V,
                                             const double* __restrict__ mean_s, int M,
                                             double* __restrict__ mu_out, double* __restrict__ var_out, int dbg_arg, int tile) {
    // timing experiments (skip GEMM1 / GEMM2 / K* / the V store) exist only in -DBOSS_EXPERIMENTS builds; the shipped
    // library compiles the switches away (dbg_arg is ignored)
#ifdef BOSS_EXPERIMENTS
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;
#endif
    static_assert(G::WC == 1 && (G::BM == BLK || G::BM == 2 * BLK), "waves stacked along a 128- or 256-row block");
    constexpr int RB = G::BM;                          // rows per substitution step; Dinv holds RB×RB inverses
    extern __shared__ double lds[];
    constexpr int BN = G::BN, TM = G::TM, TN = G::TN, LDR = PredictLds<G>::LDR;
    double* Rs = lds;
    double* red = Rs + RB * LDR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar wave index (see gemm_f64.hpp)
    const int wr = wave, wc = 0;
    const int c0 = tile * BN;
    const int nblk = Np / RB;

    // per-thread partial sums live in LDS between row blocks (slot-major [slot][tid]: conflict-free).
    // In registers they push the 64-candidate instantiation over 256 VGPRs; hipcc then spills to AGPRs
    // and copies the inline-asm prefetch ring's registers BEFORE their loads have landed.
    double* part = red + 2 * G::WR * G::BN;
#pragma unroll
    for (int u = 0; u < 2 * TN * 4; ++u) part[u * G::NTHREADS + tid] = 0.0;
    // The K* phase sits between the two GEMMs of every row block with nothing to hide behind: its operand loads must not come
    // one coordinate at a time (eight dependent L2 round trips per block cost more than its arithmetic).  The candidates'
    // coordinates — the same for every row block — are staged in LDS once, the rows' coordinates are fetched four at a time (eight would spill).
    double* cs = part + (size_t)G::NTHREADS * PredictLds<G>::PART;
    const bool cs_ok = !PRE && d <= PRED_CS_MAX_D;
    if (cs_ok) {
        for (int idx = tid; idx < d * BN; idx += G::NTHREADS) cs[idx] = Csc[(size_t)(idx / BN) * Mp + c0 + (idx % BN)];
        __syncthreads();
    }

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
        if (cs_ok && !(dbg & 1)) {
            for (int kd0 = 0; kd0 < d; kd0 += 4) {
                double xr[4][TM];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int m = 0; m < TM; ++m)
                        xr[kk][m] = (kd0 + kk < d) ? Xsc[(size_t)(kd0 + kk) * Np + ib * RB + G::row_of(wr, m, lane)] : 0.0;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (kd0 + kk < d) {
#pragma unroll
                        for (int n = 0; n < TN; ++n)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const double xc = cs[(kd0 + kk) * BN + G::col_of(wc, n, i, lane)];
#pragma unroll
                                for (int m = 0; m < TM; ++m) {
                                    const double diff = xr[kk][m] - xc;
                                    r2[m][n][i] = __builtin_fma(diff, diff, r2[m][n][i]);
                                }
                            }
                    }
                }
            }
        } else
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

template <class G, bool PRE = false>
__global__ __launch_bounds__(G::NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void predict_kernel(const double* __restrict__ A, int ld, int Np, int N,
                                                      const double* __restrict__ Dinv,
                                                      const double* __restrict__ Xsc,
                                                      const double* __restrict__ Csc, int d, int Mp, int kern,
                                                      double amp2, double* __restrict__ Vscratch,
                                                      const double* __restrict__ mean_s, int M,
                                                      double* __restrict__ mu_out, double* __restrict__ var_out, int dbg_arg) {
    predict_body<G, PRE>(A, ld, Np, N, Dinv, Xsc, Csc, d, Mp, kern, amp2, Vscratch + (size_t)blockIdx.x * Np * G::BN, mean_s, M, mu_out,
                         var_out, dbg_arg, (int)blockIdx.x);
}

// The same for a SET of equally shaped posteriors in one launch: grid = (candidate tiles, posteriors).  The S hyper-parameter
// samples of a Bayesian-inference model (src/posterior.jl:15-19: one posterior per sample; the acquisition is averaged over them,
// src/acquisitions/expected_improvement.jl:87-90) share the candidates and differ in factor, scaling and amplitude — S launches of
// M/32 workgroups each leave most of the chip idle for small M and pay S ramps and tails for large M.
struct PredSet {
    const double* A;          // factor (+ z row)
    const double* Dinv;       // 256×256 (BN = 32) diagonal-block inverses
    const double* Xsc;        // scaled training points
    const double* Csc;        // the candidates scaled by this posterior's lengthscales
    const double* mean_s;     // prior mean at the candidates, or null
    const double* invlam;     // 1/(λ + 1e-8) of this posterior (scale_cand_set_kernel)
    double* mu;
    double* var;
    double amp2;
};
template <class G>
__global__ __launch_bounds__(G::NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void predict_kernel_set(const PredSet* __restrict__ sets, int ld, int Np, int N,
                                                      int d, int Mp, int kern, double* __restrict__ Vscratch, int M) {
    const PredSet ps = sets[blockIdx.y];                     // (uniform: scalar loads)
    predict_body<G, false>(ps.A, ld, Np, N, ps.Dinv, ps.Xsc, ps.Csc, d, Mp, kern, ps.amp2,
                           Vscratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * Np * G::BN, ps.mean_s, M, ps.mu, ps.var, 0, (int)blockIdx.x);
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
constexpr int WINV_MAX_M_DECL = 4;                            // = WINV_MAX_M (defined with the one-pass kernel below)
constexpr int KSTAR_LDS_EXTRA = 8 * 256;                      // doubles: row coordinates of 8 dimensions × 256 rows
__global__ __launch_bounds__(256) void kstar_rows_kernel(const double* __restrict__ Xsc, int Np, int N,
                                                         const double* __restrict__ Csc, int d, int Mp, int kern,
                                                         double amp2, double* __restrict__ kst, int ncols) {
    // ncols: columns of the 32-wide tile that are needed (the one-to-four-candidates path reads only the first ones)
    extern __shared__ double cs[];                           // [d][32] candidate coordinates | [8][256] row coordinates
    const int c0 = blockIdx.y * 32;                          // candidate tile
    kst += (size_t)blockIdx.y * Np * 32;
    for (int idx = threadIdx.x; idx < d * 32; idx += 256) cs[idx] = Csc[(size_t)(idx >> 5) * Mp + c0 + (idx & 31)];
    if (ncols == 32) {
        // full tile: lane = candidate column, so that a wave writes two whole 256-byte rows of the tile per store
        // (one row per thread made every store touch 64 different cache lines: 244 µs for 128 tiles, now ≈ 40)
        double* xs = cs + d * 32;
        const int col = threadIdx.x & 31, rg = threadIdx.x >> 5, rbase = blockIdx.x * 256;
        double r2[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) r2[i] = 0.0;
        for (int k0 = 0; k0 < d; k0 += 8) {
            const int kc = (d - k0 < 8) ? d - k0 : 8;
            __syncthreads();
            for (int idx = threadIdx.x; idx < kc * 256; idx += 256) xs[idx] = Xsc[(size_t)(k0 + (idx >> 8)) * Np + rbase + (idx & 255)];
            __syncthreads();
            for (int kk = 0; kk < kc; ++kk) {
                const double cv = cs[(k0 + kk) * 32 + col];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const double df = xs[kk * 256 + rg + 8 * i] - cv;
                    r2[i] = __builtin_fma(df, df, r2[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int row = rbase + rg + 8 * i;
            kst[(size_t)row * 32 + col] = (row < N) ? amp2 * kappa_r2(kern, r2[i]) : 0.0;
        }
        return;
    }
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    double r2[WINV_MAX_M_DECL];
#pragma unroll
    for (int c = 0; c < WINV_MAX_M_DECL; ++c) r2[c] = 0.0;
    for (int kd = 0; kd < d; ++kd) {
        const double xr = Xsc[(size_t)kd * Np + row];
#pragma unroll
        for (int c = 0; c < WINV_MAX_M_DECL; ++c) {
            if (c < ncols) {
                const double df = xr - cs[kd * 32 + c];
                r2[c] = __builtin_fma(df, df, r2[c]);
            }
        }
    }
    const bool live = row < N;
#pragma unroll
    for (int c = 0; c < WINV_MAX_M_DECL; ++c)
        if (c < ncols) kst[(size_t)row * 32 + c] = live ? amp2 * kappa_r2(kern, r2[c]) : 0.0;
}

// GU = GemmDirect<4,1,2,2,D>: 128 rows × 32 candidates per workgroup, K = 256
// R[r0 : r0 + BM, tile] −= L[r0 : r0 + BM, 256·isrc : 256·isrc + K] · V[256·isrc : 256·isrc + K, tile]
template <class GU>
__device__ __forceinline__ void few_update_body(const double* __restrict__ A, int ld, int Np, int isrc, int K, int r0,
                                                const double* __restrict__ V, double* __restrict__ R, int tile) {
    static_assert(GU::WC == 1 && GU::BN == 32 && GU::TN == 2, "BM×32 tiles");
    constexpr int TM = GU::TM;
    V += (size_t)tile * Np * 32;
    R += (size_t)tile * Np * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= GU::NWAVES) return;                          // (launched with more waves than the tile uses: few_step_kernel)
    double* Rb = R + (size_t)r0 * 32;
    v4d acc[TM][2];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = GU::row_of(wave, m, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                        // columns 2c, 2c+1 = tiles n = 0, 1
            const v2d v = *reinterpret_cast<const v2d*>(Rb + row * 32 + GU::col_of(0, 0, i, lane));
            acc[m][0][i] = v[0];
            acc[m][1][i] = v[1];
        }
    }
    GU::template run<-1, (PRED_RB / 4) % GU::D == 0>(A + (size_t)r0 + (size_t)isrc * PRED_RB * ld, ld, V + (size_t)isrc * PRED_RB * 32, 32, K, acc);   // (K = 256 or 512: exact ring tail)
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int row = GU::row_of(wave, m, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<v2d*>(Rb + row * 32 + GU::col_of(0, 0, i, lane)) = v2d{acc[m][0][i], acc[m][1][i]};
    }
}

template <class GU>
__global__ __launch_bounds__(GU::NTHREADS) void few_update_kernel(const double* __restrict__ A, int ld, int Np, int ib,
                                                                  const double* __restrict__ V, double* __restrict__ R) {
    few_update_body<GU>(A, ld, Np, ib, PRED_RB, (ib + 1) * PRED_RB + (int)blockIdx.y * GU::BM, V, R, blockIdx.x);   // (tiles fastest in dispatch order)
}

// ------------------------------------------------------------------------------------------
// The same substitution with ONE launch per 256-row step and no dependent pair of GEMMs inside a step.  With
//     W_i = Dinv2_i · L[i, i-1]   and, for even i,   W'_i = Dinv2_i · L[i, i-2]          (few_w_kernel, once per factorisation)
// a step reads
//     odd i :   V_i = Dinv2_i · R_i' − W_i V_{i-1}                         R_i' = K*_i − Σ_{j<=i-2} L[i, j] V_j
//     even i:   V_i = Dinv2_i · R_i' − W'_i V_{i-2} − W_i V_{i-1}          R_i' = K*_i − Σ_{j<=i-3} L[i, j] V_j
// so the residual blocks only have to receive the V's in PAIRS: the launch of an even step i >= 2 also carries the update
// R_j −= L[j, i-2 : i] [V_{i-2}; V_{i-1}] (K = 512) of every block j > i, which needs nothing the same launch computes, and an
// odd step is its 8·tiles step workgroups alone.
//     few_step_kernel(i):   the first 8·tiles workgroups: rows 32p .. 32p+31 of V_i of one tile, a K = 32 (p + 1) + 256 (or 512)
//                                             product split four ways along K over the workgroup's waves (<= 192 MFMAs per
//                                             wave) and summed through LDS
//                           the others:       the pair update, few_update_body on GU::BM-row blocks.  Workgroup n runs on XCD
//                                             n mod 8: whole groups of 8 blocks are dealt block 8k + x -> XCD x (every tile of
//                                             it: each slice of the L panel is pulled into ONE L2), what is left tile by tile.
// The launch-per-GEMM form above spends 16.5 µs of every step in a finish kernel with one workgroup per tile (13.6 µs of MFMA
// issue on one CU) before the update can start, and every launch of the update ≈10 µs in ramp, operand latency and tail.
// Per-step partial sums of Σv², v·z go to part[tile][step][p][64] and are added in a fixed order by few_sum_kernel.
// W lives in slabs of 256 × 512 per block (column-major, ld 256): columns 256.. = W_i, columns 0..255 = W'_i (even i >= 2).
// ------------------------------------------------------------------------------------------
constexpr int FEW_STEP_PARTS = 8;
constexpr size_t FEW_W_SLAB = (size_t)PRED_RB * 2 * PRED_RB;

// blockIdx.y = 2 b + which (which = 1: W_b, 0: W'_b), columns 32·blockIdx.x .. +31 of it
template <class G>
__global__ __launch_bounds__(G::NTHREADS) void few_w_kernel(const double* __restrict__ A, int ld,
                                                            const double* __restrict__ Dinv2, double* __restrict__ W2) {
    constexpr int RB = G::BM, TM = G::TM, TN = G::TN, LDR = PredictLds<G>::LDR;
    static_assert(RB == PRED_RB && TN == 2 && TM % 2 == 0, "256-row blocks, 32 columns per workgroup");
    extern __shared__ double lds[];
    double* Rs = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y >> 1, which = blockIdx.y & 1, c0 = blockIdx.x * 32;
    if (which ? b < 1 : (b < 2 || (b & 1))) return;
    const double* Lb = A + (size_t)b * RB + (size_t)((b - 2 + which) * RB + c0) * ld;
#pragma unroll 8
    for (int q = 0; q < RB * 32 / 256; ++q) {                // column q of the slice, one row per thread
        Rs[tid * LDR + q] = Lb[tid + (size_t)q * ld];
    }
    __syncthreads();
    v4d acc2[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc2[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
    G::run_Blds_tri(Dinv2 + (size_t)b * RB * RB, RB, Rs, LDR, acc2);
    double* Wb = W2 + (size_t)b * FEW_W_SLAB + (size_t)which * RB * RB;
#pragma unroll
    for (int m = 0; m < TM; m += 2) {
        const int row = G::tri_row_of(wave, m, lane);        // even row; tile m + 1 holds the odd one below it
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<v2d*>(Wb + row + (size_t)(c0 + G::col_of(0, n, i, lane)) * RB) = v2d{acc2[m][n][i], acc2[m + 1][n][i]};
    }
}

template <class GU>
__global__ __launch_bounds__(256) void few_step_kernel(const double* __restrict__ A, int ld, int Np, int nb, int ib, int Np_tiles,
                                                       double* __restrict__ R, const double* __restrict__ Dinv2,
                                                       const double* __restrict__ W2, double* __restrict__ V,
                                                       double* __restrict__ part) {
    static_assert(GU::NTHREADS <= 256, "the step part uses 4 waves");
    const int ftiles = Np_tiles;
    if ((int)blockIdx.x >= FEW_STEP_PARTS * ftiles) {        // (even steps from 2 on)
        constexpr int PER = PRED_RB / GU::BM;
        const int n = (int)blockIdx.x - FEW_STEP_PARTS * ftiles;            // (8·tiles is a multiple of 8: n mod 8 is the XCD)
        const int nblk = (nb - 1 - ib) * PER, full = (nblk >> 3) * ftiles;
        const int k = n >> 3, x = n & 7;
        int blk, tile;
        if (k < full) {
            blk = 8 * (k / ftiles) + x;
            tile = k % ftiles;
        } else {
            const int m = (k - full) * 8 + x;
            if (m >= (nblk & 7) * ftiles) return;
            blk = (nblk & ~7) + m / ftiles;
            tile = m % ftiles;
        }
        few_update_body<GU>(A, ld, Np, ib - 2, 2 * PRED_RB, (ib + 1) * PRED_RB + blk * GU::BM, V, R, tile);
        return;
    }
    typedef GemmDirect<1, 1, 2, 2, 8> G1;                    // 32 rows × 32 candidates per wave; its row offset is wave·32: undone below
    __shared__ double red[3][32 * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = (int)blockIdx.x / ftiles, tile = (int)blockIdx.x % ftiles;
    R += (size_t)tile * Np * 32;
    V += (size_t)tile * Np * 32;
    v4d acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
    {   // Dinv2_ib[rows, 0 : 32 (p + 1)) · R_ib: this wave's quarter of the columns
        const int kn = 8 * (p + 1), k0 = wave * kn;
        G1::template run<1>(Dinv2 + (size_t)ib * PRED_RB * PRED_RB + 32 * p - 32 * wave + (size_t)k0 * PRED_RB, PRED_RB,
                            R + ((size_t)ib * PRED_RB + k0) * 32, 32, kn, acc);
    }
    if (ib > 0) {                                            // − W_ib[rows, :] · V_{ib-1}, for an even step − [W'_ib | W_ib][rows, :] · [V_{ib-2}; V_{ib-1}]
        const int two = (ib & 1) == 0 ? 1 : 0;
        const int kn = two ? 128 : 64, k0 = kn * wave, c0 = two ? 0 : PRED_RB;
        G1::template run<-1>(W2 + (size_t)ib * FEW_W_SLAB + 32 * p - 32 * wave + (size_t)(c0 + k0) * PRED_RB, PRED_RB,
                             V + ((size_t)(ib - 1 - two) * PRED_RB + k0) * 32, 32, kn, acc);
    }
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<v2d*>(&red[wave - 1][G1::row_of(0, m, lane) * 32 + G1::col_of(0, 0, i, lane)]) = v2d{acc[m][0][i], acc[m][1][i]};
    }
    __syncthreads();
    if (wave > 0) return;
    double ps[2][4], pz[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) ps[n][i] = pz[n][i] = 0.0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int rl = G1::row_of(0, m, lane);
        const int row = ib * PRED_RB + 32 * p + rl;
        const double zr = A[(size_t)row * ld + Np];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = G1::col_of(0, 0, i, lane);
            v2d v = v2d{acc[m][0][i], acc[m][1][i]};
#pragma unroll
            for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const v2d*>(&red[w][rl * 32 + col]);
            *reinterpret_cast<v2d*>(V + (size_t)row * 32 + col) = v;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                ps[n][i] = __builtin_fma(v[n], v[n], ps[n][i]);
                pz[n][i] = __builtin_fma(v[n], zr, pz[n][i]);
            }
        }
    }
    double* pt = part + (((size_t)tile * nb + ib) * FEW_STEP_PARTS + p) * 64;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double sv = ps[n][i], z = pz[n][i];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                sv += __shfl_xor(sv, off);
                z += __shfl_xor(z, off);
            }
            if ((lane & 15) == 0) {
                const int col = G1::col_of(0, n, i, lane);
                pt[col] = sv;
                pt[32 + col] = z;
            }
        }
}

// μ, σ² from the per-step partial sums: four quarters of the (step, part) list side by side, each in order, then the quarters in order
__global__ __launch_bounds__(256) void few_sum_kernel(const double* __restrict__ part, int nb, const double* __restrict__ mean_s,
                                                      int M, double amp2, double* __restrict__ mu_out,
                                                      double* __restrict__ var_out, int aug) {
    __shared__ double q[4][64];
    const int tid = threadIdx.x, c0 = blockIdx.x * 32, v = tid & 63, g = tid >> 6;
    const int nk = nb * FEW_STEP_PARTS, k0 = g * (nk / 4), k1 = g == 3 ? nk : k0 + nk / 4;
    const double* pt = part + (size_t)blockIdx.x * nk * 64 + v;
    double s = 0.0;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) s += pt[(size_t)k * 64];
    q[g][v] = s;
    __syncthreads();
    if (tid < 32 && c0 + tid < M) {
        const double ss = ((q[0][tid] + q[1][tid]) + q[2][tid]) + q[3][tid];
        const double z = ((q[0][32 + tid] + q[1][32 + tid]) + q[2][32 + tid]) + q[3][32 + tid];
        mu_out[c0 + tid] = (mean_s ? mean_s[c0 + tid] : 0.0) + z;
        var_out[c0 + tid] = aug == 2 ? -ss : aug ? fmax(0.0, amp2 - ss) : amp2 - ss + PREDICT_JITTER;
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
constexpr int WINV_MAX_M = WINV_MAX_M_DECL;
constexpr int FEW_ARGS_MAX_D = 16;                          // x_dim up to which few candidates / one appended point travel in kernel arguments
// The walk of one wave over its two rows k0, k0+1 (columns of U, contiguous in c; row k0 ends at c = k0, row k0+1 at c = k0+1):
// a0[j] += Σ_c U[c, k0] k*_c[j], a1 likewise — per lane, to be summed over the wave by the caller.  16-byte loads, four 128-element
// chunks of both rows in flight per lane.  Shared by winv_gemv_kernel and winv_args_kernel (small_calls.hpp): same sums in the
// same order.  (Measured and not kept: all four waves of a workgroup sharing the column range of its eight rows — 23 against
// 22 µs per pass; the pass is not bound by its longest wave.)
template <int MC>
__device__ __forceinline__ void winv_walk(const double* __restrict__ col0, const double* __restrict__ col1, const double* __restrict__ ks,
                                          int k0, int lane, double (&a0)[MC], double (&a1)[MC]) {
    int base = 0;
    for (; base + 512 <= k0 + 1; base += 512) {              // every element of both rows in [base, base + 512) is inside the triangle
        v2d u0[4], u1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = base + 128 * u + 2 * lane;
            u0[u] = *reinterpret_cast<const v2d*>(col0 + c);
            u1[u] = *reinterpret_cast<const v2d*>(col1 + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = base + 128 * u + 2 * lane;
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                const double q0 = ks[c * MC + j], q1 = ks[(c + 1) * MC + j];
                a0[j] = __builtin_fma(u0[u][0], q0, a0[j]);
                a1[j] = __builtin_fma(u1[u][0], q0, a1[j]);
                a0[j] = __builtin_fma(u0[u][1], q1, a0[j]);
                a1[j] = __builtin_fma(u1[u][1], q1, a1[j]);
            }
        }
    }
    for (int c = base + lane; c <= k0 + 1; c += 64) {         // the rest, element by element (row k0 stops one entry before row k0+1)
        const double u0 = (c <= k0) ? col0[c] : 0.0, u1 = col1[c];
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            const double q = ks[c * MC + j];
            a0[j] = __builtin_fma(u0, q, a0[j]);
            a1[j] = __builtin_fma(u1, q, a1[j]);
        }
    }
}
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
    winv_walk<MC>(col0, col1, ks, k0, lane, a0, a1);
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
    GU::template run<1, (BLK / 4) % GU::D == 0>(Linv + r0, ldl, B, 32, r0 + BM, acc);   // (K a multiple of 128: exact ring tail)
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
    GU::template run<1, (BLK / 4) % GU::D == 0>(Uinv + r0 + (size_t)r0 * ldu, ldu, V + (size_t)r0 * 32, 32, Np - r0, acc);
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
    int base = c0;                                           // (even: 16-byte aligned)
    if (two) {
        if (lane == 0) {                                     // rows c0, c0+1 by hand: entry (c0, c0+1) lies above the diagonal
            a0 = col0[c0] * l[c0] + col0[c0 + 1] * l[c0 + 1];
            a1 = col1[c0 + 1] * l[c0 + 1];
        }
        base = c0 + 2;
        for (; base + 512 <= N0; base += 512) {              // 16-byte loads, four chunks of both columns in flight per lane: 25 -> 14 µs per pass
            v2d u0[4], u1[4], lr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = base + 128 * u + 2 * lane;
                u0[u] = *reinterpret_cast<const v2d*>(col0 + r);
                u1[u] = *reinterpret_cast<const v2d*>(col1 + r);
                lr[u] = *reinterpret_cast<const v2d*>(l + r);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_fma(u0[u][0], lr[u][0], a0);
                a1 = __builtin_fma(u1[u][0], lr[u][0], a1);
                a0 = __builtin_fma(u0[u][1], lr[u][1], a0);
                a1 = __builtin_fma(u1[u][1], lr[u][1], a1);
            }
        }
    }
    for (int r = base + lane; r < N0; r += 64) {
        const double lr = l[r];
        a0 = __builtin_fma(col0[r], lr, a0);
        if (two) a1 = __builtin_fma(col1[r], lr, a1);
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
                                                             double* __restrict__ scal, double* __restrict__ dz, int* __restrict__ info,
                                                             double* __restrict__ host_out = nullptr, unsigned long long res_seq = 0) {
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
        if (host_out) {
            // the caller's answer (log-likelihood, status) is known HERE: it goes to mapped host memory now, and the host returns while the
            // second pass over the inverse factors and the patches are still running (everything that follows is ordered on the stream)
            host_out[0] = scal[0];
            host_out[1] = scal[1];
            reinterpret_cast<int*>(host_out + 2)[0] = *info;
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + 3), res_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// One new observation whose coordinates (raw and scaled), value and prior mean travel in the kernel arguments (boss_gp_append with
// n = 1 on resident inverse factors): written into the handle's point / observation arrays, failed-pivot flag cleared — instead
// of three small uploads and a stream synchronisation in front of the append.
struct AppendPoint {
    int d, N0, Np;
    double y, mean;
    double raw[FEW_ARGS_MAX_D], sc[FEW_ARGS_MAX_D];
};
__global__ __launch_bounds__(64) void append_point_kernel(AppendPoint par, double* __restrict__ Xraw, double* __restrict__ Xsc,
                                                          double* __restrict__ y, double* __restrict__ mean, int* __restrict__ info) {
    const int k = threadIdx.x;
    if (k < par.d) {
        Xraw[(size_t)k * par.Np + par.N0] = par.raw[k];
        Xsc[(size_t)k * par.Np + par.N0] = par.sc[k];
    }
    if (k == 0) {
        y[par.N0] = par.y;
        mean[par.N0] = par.mean;
        if (info) *info = 0;                                 // (null: the fused append pass in front of this kernel has already set it)
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
// One to four candidates per call on resident inverse factors, the reference's own call pattern (`acq.(eachcol(xs))` evaluates one
// point at a time: expected_improvement.jl:75,79; every objective evaluation of an Optimization.jl maximiser is such a call).  The
// call is latency, not work: 21 µs of it is the pass over L⁻ᵀ, the rest used to be two copies, five launches and a stream
// synchronisation.  Here the (host-scaled) candidates and their prior means travel in the kernel ARGUMENTS, K* is evaluated by
// kstar_args_kernel, and winv_finish_host_kernel clips the variances (_clip_var, gaussian_process.jl:186-194) and writes
// {μ, σ², first bad index} and then a sequence number into mapped host memory, which the host polls: three launches, no copy.
// ------------------------------------------------------------------------------------------
struct FewCand {
    int ncols;
    double x[WINV_MAX_M][FEW_ARGS_MAX_D];                    // scaled (and, for DiscreteKernel dims, rounded) coordinates
    double mean[WINV_MAX_M];                                 // prior mean at the candidates
};
__global__ __launch_bounds__(256) void kstar_args_kernel(FewCand par, const double* __restrict__ Xsc, int Np, int N, int d, int kern,
                                                         double amp2, double* __restrict__ kst, int stride = 32) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    double r2[WINV_MAX_M];
#pragma unroll
    for (int c = 0; c < WINV_MAX_M; ++c) r2[c] = 0.0;
    for (int kd = 0; kd < d; ++kd) {
        const double xr = Xsc[(size_t)kd * Np + row];
#pragma unroll
        for (int c = 0; c < WINV_MAX_M; ++c) {
            const double df = xr - par.x[c][kd];
            r2[c] = __builtin_fma(df, df, r2[c]);
        }
    }
    const bool live = row < N;
#pragma unroll
    for (int c = 0; c < WINV_MAX_M; ++c)
        if (c < par.ncols) kst[(size_t)row * stride + c] = live ? amp2 * kappa_r2(kern, r2[c]) : 0.0;
}
// host_out (mapped host memory): [0..4) μ, [4..8) σ² (clipped), [8] first index whose variance is below -1e-8 (or -1), [9] sequence number
__global__ __launch_bounds__(256) void winv_finish_host_kernel(const double* __restrict__ part, int nwg, FewCand par, double amp2,
                                                               double* __restrict__ host_out, unsigned long long seq) {
    __shared__ double red[8][256];
    const int tid = threadIdx.x, M = par.ncols;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    for (int w = tid; w < nwg; w += 256)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += part[(size_t)w * 8 + q];
#pragma unroll
    for (int q = 0; q < 8; ++q) red[q][tid] = acc[q];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {               // fixed-order tree: deterministic (same as winv_finish_kernel)
        if (tid < off)
#pragma unroll
            for (int q = 0; q < 8; ++q) red[q][tid] += red[q][tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        long long bad = -1;
        for (int j = 0; j < M; ++j) {
            const double s = red[2 * j][0], z = red[2 * j + 1][0];
            double v = amp2 - s + PREDICT_JITTER;
            if (!(v >= 0.0)) {                                // (NaN counts as offending, as in clip_var_kernel)
                if (v >= -MAX_NEG_VAR) v = 0.0;
                else if (bad < 0) bad = j;                    // DomainError: the value stays as it is (boss_gp_predict reports it)
            }
            host_out[j] = par.mean[j] + z;
            host_out[4 + j] = v;
        }
        reinterpret_cast<long long*>(host_out)[8] = bad;
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out) + 9, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
                                                            double* __restrict__ Cbase, int ldc, size_t sC, double alpha,
                                                            size_t zA, size_t zB, size_t zC) {
    // blockIdx.y: pair of diagonal blocks (strides sA, sB, sC); blockIdx.z: matrix of a batch (strides zA, zB, zC)
    __shared__ double As[32][33], Bs[32][33];
    const double* A = Abase + (size_t)blockIdx.y * sA + (size_t)blockIdx.z * zA;
    const double* B = Bbase + (size_t)blockIdx.y * sB + (size_t)blockIdx.z * zB;
    double* C = Cbase + (size_t)blockIdx.y * sC + (size_t)blockIdx.z * zC;
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
                                                                 double* __restrict__ Dinv2, size_t z128, size_t z2) {
    const int p = blockIdx.y, c = blockIdx.x;                     // column c of the 256×256 block; blockIdx.z: matrix of a batch
    const double* src = Dinv128 + (size_t)blockIdx.z * z128 + (size_t)(2 * p + (c >> 7)) * BLK * BLK + (size_t)(c & 127) * BLK;
    double* dst = Dinv2 + (size_t)blockIdx.z * z2 + (size_t)p * 4 * BLK * BLK + (size_t)c * 2 * BLK;
    const int r = threadIdx.x;                                    // 0..255
    if (c < BLK) {
        if (r < BLK) dst[r] = src[r];                             // lower-left quadrant is written by the GEMM
    } else {
        dst[r] = (r < BLK) ? 0.0 : src[r - BLK];
    }
}

}  // namespace boss

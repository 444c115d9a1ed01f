// potrf.hpp — blocked right-looking fp64 Cholesky on MI355X, with the right-hand side
// (y - m)^T carried through the factorisation as an extra row block, so that
//     L = chol(K + σ²I)   and   z = L \ (y - m)
// come out of ONE pass (no separate triangular-solve chain), which is all that
// logpdf(FiniteGP, y) = -(N log 2π + logdet C + ||C.U' \ δ||²)/2 needs
// (reference: src/models/gaussian_process.jl:279, algebra in src/models/gradient_gp.jl:325-326,403).
//
// Matrix layout (per batch entry): column-major, ld = Np + RHS_ROWS, Np = N rounded up to 128;
// rows 0..Np-1 hold the lower triangle of K (padding = identity), row Np holds δ^T.
// Step k (128 columns):
//   potrf_diag_kernel   1 workgroup : factor the 128×128 diagonal block in LDS, emit inv(L16) blocks
//   potrf_trsm_kernel   1 wave / 16 rows : rows below  ←  rows · L_kk^{-T}   (registers only, MFMA)
//   potrf_syrk_kernel   128×128 MFMA tiles : trailing  -=  P P^T
#pragma once
#include "gemm_f64.hpp"

namespace boss {

constexpr int LDD = 144;                               // LDS leading dim of the diagonal block
constexpr int DIAG_LDS_BYTES = (BLK * LDD + BLK) * 8;  // block + reciprocal diagonal

// ------------------------------------------------------------------------------------------
// Diagonal block: unblocked 16-column panels (lane = row, pivots broadcast with v_readlane),
// MFMA rank-16 updates of the rest of the block inside LDS.
// inv16 out: for each of the 8 diagonal 16×16 blocks its inverse X (column-major 16×16,
// X(r,c) at c*16+r, zero above the diagonal).
// info: first failing global column + 1 (0 = success) — PosDefException analogue.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void potrf_diag_kernel(double* __restrict__ Abase, int ld, size_t bstride,
                                                         int k, double* __restrict__ inv16base,
                                                         size_t inv16_bstride, int* __restrict__ info) {
    extern __shared__ double smem[];
    double* D = smem;
    double* rdiag = smem + BLK * LDD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* A = Abase + (size_t)blockIdx.z * bstride + (size_t)k * BLK * ((size_t)ld + 1);
    double* inv16 = inv16base + (size_t)blockIdx.z * inv16_bstride + (size_t)k * (8 * 256);

    for (int idx = tid; idx < BLK * (BLK / 2); idx += 256) {
        int c = idx / (BLK / 2), rp = idx % (BLK / 2);
        *reinterpret_cast<v2d*>(D + c * LDD + 2 * rp) = *reinterpret_cast<const v2d*>(A + (size_t)c * ld + 2 * rp);
    }
    __syncthreads();

    int fail = -1;
    for (int jb = 0; jb < 8; ++jb) {
        // ---- panel of 16 columns: every wave factors the 16×16 diagonal block redundantly in
        //      lanes 0..15 and carries 48 rows below it in lanes 16..63 --------------------------
        int row;
        bool active;
        if (lane < 16) {
            row = jb * 16 + lane;
            active = true;
        } else {
            row = (jb + 1) * 16 + wave * 48 + (lane - 16);
            active = row < BLK;
        }
        double x[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            int col = jb * 16 + c;
            int addr = (lane < 16 && c > lane) ? (row * LDD + col) : (col * LDD + row);
            x[c] = active ? D[addr] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double p = readlane_f64(x[j], j);
            if (!(p > 0.0)) {                      // NaN or non-positive pivot: not PD
                if (fail < 0) fail = k * BLK + jb * 16 + j;
                p = 1.0;
            }
            double inv = rsqrt_refined(p);
            if (tid == 0) rdiag[jb * 16 + j] = inv;
            double lj = x[j] * inv;
            x[j] = lj;
#pragma unroll
            for (int c = j + 1; c < 16; ++c) {
                double lc = readlane_f64(lj, c);
                x[c] = __builtin_fma(-lj, lc, x[c]);
            }
        }
        if (active && (lane >= 16 || wave == 0)) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (lane >= 16 || c <= lane) D[(jb * 16 + c) * LDD + row] = x[c];
            }
        }
        __syncthreads();
        // ---- rank-16 update of the remaining lower 16×16 tiles (MFMA, operands from LDS) -----
        const int t = 7 - jb;
        const int T = t * (t + 1) / 2;
        for (int q = wave; q < T; q += 4) {
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= q) ++a;
            int b = q - a * (a + 1) / 2;
            int ti = jb + 1 + a, tj = jb + 1 + b;
            v4d cr;
#pragma unroll
            for (int i = 0; i < 4; ++i) cr[i] = D[(tj * 16 + (lane >> 4) + 4 * i) * LDD + ti * 16 + (lane & 15)];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                int kc = (jb * 16 + 4 * s + (lane >> 4)) * LDD;
                double af = D[kc + tj * 16 + (lane & 15)];
                double bf = D[kc + ti * 16 + (lane & 15)];
                cr = mfma_f64(-af, bf, cr);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) D[(tj * 16 + (lane >> 4) + 4 * i) * LDD + ti * 16 + (lane & 15)] = cr[i];
        }
        __syncthreads();
    }
    if (tid == 0 && fail >= 0) {
        if (info[blockIdx.z] == 0) info[blockIdx.z] = fail + 1;
    }
    // ---- write L (lower part only) -------------------------------------------------------------
    for (int idx = tid; idx < BLK * BLK; idx += 256) {
        int c = idx / BLK, r = idx % BLK;
        if (r >= c) A[(size_t)c * ld + r] = D[c * LDD + r];
    }
    // ---- inverses of the eight 16×16 diagonal blocks: lane = column, forward substitution ------
    if (tid < 128) {
        const int blk = tid >> 4, c = tid & 15, base = blk * 16;
        double xc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int kk = 0; kk < r; ++kk) s = __builtin_fma(-D[(base + kk) * LDD + base + r], xc[kk], s);
            xc[r] = s * rdiag[base + r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) inv16[blk * 256 + c * 16 + r] = xc[r];
    }
}

// ------------------------------------------------------------------------------------------
// One wave solves 16 rows against the 128×128 factored diagonal block, entirely in registers:
//   P^T_jb = inv16_jb · (B^T_jb − Σ_{m<jb} L[jb,m] · P^T_m)
// The f64 MFMA result layout (row = (l>>4)+4i) equals its B-operand layout (k = (l>>4)+4s), so
// finished P^T_m tiles feed the next MFMAs straight from their accumulator registers.
// IDENTITY=true solves for rows of I instead (→ rows of L_kk^{-T}) and stores the transposed
// result as the dense inverse Dinv_k (used by the prediction kernel).
// ------------------------------------------------------------------------------------------
template <bool IDENTITY>
__device__ __forceinline__ void wave_trsm16(const double* __restrict__ Lkk, int ld,
                                            const double* __restrict__ inv16k, v4d (&acc)[8], int lane) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
#pragma unroll
        for (int m = 0; m < jb; ++m) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                double a = Lkk[(size_t)(m * 16 + 4 * s + (lane >> 4)) * ld + jb * 16 + (lane & 15)];
                acc[jb] = mfma_f64(-a, acc[m][s], acc[jb]);
            }
        }
        v4d nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double a = inv16k[jb * 256 + (4 * s + (lane >> 4)) * 16 + (lane & 15)];
            nw = mfma_f64(a, acc[jb][s], nw);
        }
        acc[jb] = nw;
    }
}

__global__ __launch_bounds__(64) void potrf_trsm_kernel(double* __restrict__ Abase, int ld, size_t bstride, int k,
                                                        const double* __restrict__ inv16base,
                                                        size_t inv16_bstride) {
    const int lane = threadIdx.x;
    double* A = Abase + (size_t)blockIdx.z * bstride;
    const double* Lkk = A + (size_t)k * BLK * ((size_t)ld + 1);
    const double* inv16k = inv16base + (size_t)blockIdx.z * inv16_bstride + (size_t)k * (8 * 256);
    double* Brow = A + (size_t)(k + 1) * BLK + (size_t)blockIdx.x * 16 + (size_t)k * BLK * ld;
    v4d acc[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[jb][i] = Brow[(lane & 15) + (size_t)(jb * 16 + (lane >> 4) + 4 * i) * ld];
    wave_trsm16<false>(Lkk, ld, inv16k, acc, lane);
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int i = 0; i < 4; ++i) Brow[(lane & 15) + (size_t)(jb * 16 + (lane >> 4) + 4 * i) * ld] = acc[jb][i];
}

// Dense inverses of all diagonal blocks (grid: 8 row groups × NBLK × batch), off the critical path.
__global__ __launch_bounds__(64) void potrf_dinv_kernel(const double* __restrict__ Abase, int ld, size_t bstride,
                                                        const double* __restrict__ inv16base, size_t inv16_bstride,
                                                        double* __restrict__ dinvbase, size_t dinv_bstride) {
    const int lane = threadIdx.x;
    const int k = blockIdx.y, r0 = blockIdx.x * 16;
    const double* A = Abase + (size_t)blockIdx.z * bstride;
    const double* Lkk = A + (size_t)k * BLK * ((size_t)ld + 1);
    const double* inv16k = inv16base + (size_t)blockIdx.z * inv16_bstride + (size_t)k * (8 * 256);
    double* Dinv = dinvbase + (size_t)blockIdx.z * dinv_bstride + (size_t)k * BLK * BLK;
    v4d acc[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[jb][i] = ((r0 + (lane & 15)) == (jb * 16 + (lane >> 4) + 4 * i)) ? 1.0 : 0.0;
    wave_trsm16<true>(Lkk, ld, inv16k, acc, lane);
    // acc holds P(r, c) = L^{-T}(r, c) at r = r0 + (l&15), c = jb*16 + (l>>4) + 4i;  Dinv(c, r) = P(r, c)
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = r0 + (lane & 15), c = jb * 16 + (lane >> 4) + 4 * i;
            Dinv[(size_t)r * BLK + c] = (c >= r) ? acc[jb][i] : 0.0;
        }
}

// ------------------------------------------------------------------------------------------
// Trailing update  C_ij -= P_i P_j^T  for the lower block triangle behind panel k, plus the
// right-hand-side row block (i == m, only its first 32 rows are live).
// ------------------------------------------------------------------------------------------
typedef GemmNT<2, 2, 4, 4> SyrkG;     // 128×128 tile
typedef GemmNT<1, 4, 2, 2> RhsG;      // 32×128 tile for the δ^T row block
constexpr int SYRK_LDS_BYTES = (SyrkG::LDS_DOUBLES > RhsG::LDS_DOUBLES ? SyrkG::LDS_DOUBLES : RhsG::LDS_DOUBLES) * 8;

template <class G>
__device__ __forceinline__ void syrk_tile(double* __restrict__ A, int ld, int k, int R0, int C0, double* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / G::WC, wc = wave % G::WC;
    const double* Pi = A + R0 + (size_t)k * BLK * ld;
    const double* Pj = A + C0 + (size_t)k * BLK * ld;
    v4d acc[G::TM][G::TN];
#pragma unroll
    for (int m = 0; m < G::TM; ++m)
#pragma unroll
        for (int n = 0; n < G::TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
    G::run(Pi, ld, Pj, ld, BLK, acc, lds);
#pragma unroll
    for (int m = 0; m < G::TM; ++m)
#pragma unroll
        for (int n = 0; n < G::TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                size_t off = (size_t)(R0 + G::row_of(wr, m, lane)) + (size_t)(C0 + G::col_of(wc, n, i, lane)) * ld;
                A[off] -= acc[m][n][i];
            }
}

__global__ __launch_bounds__(256) void potrf_syrk_kernel(double* __restrict__ Abase, int ld, size_t bstride, int k,
                                                         int m) {
    extern __shared__ double lds[];
    double* A = Abase + (size_t)blockIdx.z * bstride;
    const int t = blockIdx.x;
    const int nsq = m * (m + 1) / 2;
    if (t < nsq) {
        int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        int j = t - i * (i + 1) / 2;
        syrk_tile<SyrkG>(A, ld, k, (k + 1 + i) * BLK, (k + 1 + j) * BLK, lds);
    } else {
        int j = t - nsq;
        syrk_tile<RhsG>(A, ld, k, (k + 1 + m) * BLK, (k + 1 + j) * BLK, lds);
    }
}

// logdet = 2 Σ_{i<N} log L_ii ,  zz = Σ_{j<N} z_j²   →  scal[2*b], scal[2*b+1]
__global__ __launch_bounds__(256) void potrf_logdet_kernel(const double* __restrict__ Abase, int ld, size_t bstride,
                                                           int N, int Np, double* __restrict__ scal) {
    const double* A = Abase + (size_t)blockIdx.z * bstride;
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        s0 += log(A[(size_t)i * ld + i]);
        double z = A[(size_t)i * ld + Np];
        s1 += z * z;
    }
    __shared__ double r0[256], r1[256];
    r0[threadIdx.x] = s0;
    r1[threadIdx.x] = s1;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            r0[threadIdx.x] += r0[threadIdx.x + st];
            r1[threadIdx.x] += r1[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scal[2 * blockIdx.z] = 2.0 * r0[0];
        scal[2 * blockIdx.z + 1] = r1[0];
    }
}

}  // namespace boss

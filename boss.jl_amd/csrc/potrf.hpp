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
constexpr int DIAG_LDS_BYTES = (BLK * LDD + 256) * 8;  // block + E tile of the current panel

// ------------------------------------------------------------------------------------------
// Diagonal block: unblocked 16-column panels (lane = row, pivots broadcast with v_readlane),
// MFMA rank-16 updates of the rest of the block inside LDS.
// inv16 out: for each of the 8 diagonal 16×16 blocks its inverse X (column-major 16×16,
// X(r,c) at c*16+r, zero above the diagonal).
// info: first failing global column + 1 (0 = success) — PosDefException analogue.
// ------------------------------------------------------------------------------------------
// Workgroup barrier for data exchanged through LDS only: unlike __syncthreads() it does not wait for
// this wave's outstanding global stores (inv16 goes to global right before barrier 1).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int DIAG_THREADS = 1024;   // 16 waves = 4 per SIMD: single-wave fp64 VALU / LDS / MFMA issue rates are 3-5x below the multi-wave rates

// 16×16 tile in the "transposed C" layout: register i of lane l = element (row = l&15, col = (l>>4)+4i),
// so register t IS the 16×4 micro-panel of columns 4t..4t+3 in MFMA A/B-operand layout.
//
// chol16_with_inverse: factor the symmetric-filled tile S in place (lower part = L16) and carry the
// identity tile E through the same column operations, which leaves E = L16^{-T} (so inv(L16) comes
// out of the factorisation itself).  This runs on ONE wave, and a lone wave pays ≈12 cycles per fp64
// VALU instruction and ≈5 per 32-bit one whether or not they depend on each other — so the code
// minimises the instruction COUNT per column: the column's entries reach the lanes that need them by
// three ds_bpermute (issued before the 1/sqrt(p) refinement, which hides their latency) instead of
// v_readlane/select chains, the update is one unconditional fma per tile with a pre-masked factor,
// and the scaling of the finished columns is deferred to the end of the 4-column micro-panel.  Then
// ONE rank-4 MFMA per tile updates the remaining columns.
__device__ __forceinline__ void chol16_with_inverse(v4d& S, v4d& E, int lane, int col0, int& fail) {
    const int r16 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) E[i] = (r16 == q + 4 * i) ? 1.0 : 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        double rs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double p = readlane_f64(S[t], 16 * j + 4 * t + j);
            double lS = 0.0, lE = 0.0, lc = 0.0;
            if (j < 3) {
                // column j is still UNSCALED (= L(:,j)·sqrt(p)):  X(r,c) -= X(r,j)·S(c,j) / p  for the columns c > j
                lS = __shfl(S[t], 16 * j + r16);            // S(r, j): same row, column j
                lE = __shfl(E[t], 16 * j + r16);
                lc = __shfl(S[t], 16 * j + 4 * t + q);      // S(c, j) for this lane's own column c = 4t+q
            }
            if (!(p > 0.0)) {                      // NaN or non-positive pivot: not PD
                if (fail < 0) fail = col0 + 4 * t + j;
                p = 1.0;
            }
            const double inv = rsqrt_refined(p);
            rs[j] = inv;
            if (j < 3) {
                const double lcm = (q > j) ? lc : 0.0;
                const double m = lcm * (inv * inv);
                S[t] = __builtin_fma(-lS, m, S[t]);
                E[t] = __builtin_fma(-lE, m, E[t]);
            }
        }
        const double sc = (q == 0) ? rs[0] : (q == 1) ? rs[1] : (q == 2) ? rs[2] : rs[3];
        S[t] *= sc;
        E[t] *= sc;
        if (t < 3) {
            // rank-4 update of columns > 4t+3:  X(r,c) -= Σ_k X(r,4t+k) · S(c,4t+k)
            const double am = (r16 > 4 * t + 3) ? -S[t] : 0.0;
            const double bS = S[t], bE = E[t];
            S = mfma_f64(am, bS, S);
            E = mfma_f64(am, bE, E);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Diagonal block (128×128, LDS-resident), 16-column panels, look-ahead inside the block:
//   phase A  wave 0 alone runs the sequential pivot chain on the diagonal 16×16 tile
//            (chol16_with_inverse) — meanwhile waves 1..15 are still applying the PREVIOUS panel's
//            rank-16 update to the rest of the block;
//   phase B  waves 1..7-jb turn their row tile into P = X · inv(L16)^T with 4 MFMAs;
//   phase C  wave 0 updates only the NEXT diagonal tile and goes straight back to phase A,
//            the other waves update everything else.
// inv16 out: for each of the 8 diagonal 16×16 blocks its inverse X (column-major 16×16,
// X(r,c) at c*16+r, zero above the diagonal).
// info: first failing global column + 1 (0 = success) — PosDefException analogue.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DIAG_THREADS) void potrf_diag_kernel(double* __restrict__ Abase, int ld, size_t bstride,
                                                                  int k, double* __restrict__ inv16base,
                                                                  size_t inv16_bstride, int* __restrict__ info) {
    extern __shared__ double smem[];
    double* D = smem;
    double* Es = smem + BLK * LDD;                   // E = L16^{-T} of the current panel, row-major 16×16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: tile decode runs on the SALU
    const int r16 = lane & 15, q = lane >> 4;
    constexpr int NW = DIAG_THREADS / 64;
    double* A = Abase + (size_t)blockIdx.z * bstride + (size_t)k * BLK * ((size_t)ld + 1);
    double* inv16 = inv16base + (size_t)blockIdx.z * inv16_bstride + (size_t)k * (8 * 256);

    for (int idx = tid; idx < BLK * (BLK / 2); idx += DIAG_THREADS) {
        int c = idx / (BLK / 2), rp = idx % (BLK / 2);
        *reinterpret_cast<v2d*>(D + c * LDD + 2 * rp) = *reinterpret_cast<const v2d*>(A + (size_t)c * ld + 2 * rp);
    }
    __syncthreads();

    int fail = -1;
    v4d S = {0.0, 0.0, 0.0, 0.0};
    // loop-invariant per-lane LDS offsets of a diagonal tile relative to its corner (the pivot-chain
    // wave pays ≈5 cycles per integer instruction too): symmetric-fill source, plain element, E slot
    int osym[4], oel[4], oes[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = q + 4 * i;
        osym[i] = (r16 >= c) ? (c * LDD + r16) : (r16 * LDD + c);
        oel[i] = c * LDD + r16;
        oes[i] = r16 * 16 + c;
    }
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) S[i] = D[osym[i]];   // symmetric fill of diagonal tile 0 from its lower part
    }
    for (int jb = 0; jb < 8; ++jb) {
        const int t = 7 - jb;                         // row tiles below the diagonal tile
        // ---------------- phase A ----------------
        if (wave == 0) {
            v4d E;
            chol16_with_inverse(S, E, lane, k * BLK + jb * 16, fail);
            double* Dt = D + jb * 16 * (LDD + 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Dt[oel[i]] = S[i];                    // whole tile: its strict upper part is scratch nobody reads
                Es[oes[i]] = E[i];                    // E(row r16, col c) = inv(L16)(c, r16)
            }
        } else if (jb > 0 && (wave & 3) != 0) {
            // rest of the rank-16 update with panel jb-1 (tile (jb,jb) was done by wave 0 in phase C).
            // Waves 4, 8, 12 share wave 0's SIMD: they stay idle here, so the pivot chain's MFMAs and
            // VALU ops never queue behind update MFMAs (measured: 5450 → 3980 cycles per 16-column chain).
            const int jp = jb - 1, tp = 7 - jp;
            const int T = tp * (tp + 1) / 2;
            constexpr int U = 2;
            constexpr int NUPD = NW - NW / 4;                   // 12 update waves
            const int uw = wave - 1 - (wave >> 2);              // 0..11
            for (int q0 = 1 + uw; q0 < T; q0 += U * NUPD) {     // update waves cover tile indices 1..T-1
                int ti[U], tj[U];
                bool ok[U];
                v4d cr[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int qq = q0 + NUPD * u;
                    ok[u] = qq < T;
                    int a = 0;
                    while ((a + 1) * (a + 2) / 2 <= qq) ++a;
                    const int b = qq - a * (a + 1) / 2;
                    ti[u] = ok[u] ? jp + 1 + a : jp + 2;
                    tj[u] = ok[u] ? jp + 1 + b : jp + 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) cr[u][i] = D[(tj[u] * 16 + q + 4 * i) * LDD + ti[u] * 16 + r16];
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int kc = (jp * 16 + 4 * s4 + q) * LDD + r16;
                    double af[U], bf[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        af[u] = D[kc + tj[u] * 16];
                        bf[u] = D[kc + ti[u] * 16];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) cr[u] = mfma_f64(-af[u], bf[u], cr[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (ok[u]) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) D[(tj[u] * 16 + q + 4 * i) * LDD + ti[u] * 16 + r16] = cr[u][i];
                    }
                }
            }
        }
        lds_barrier();                                // barrier 1: L16/E published, block fully updated through panel jb-1
        if (wave == 4) {                              // idle partner of the pivot-chain wave: inv(L16_jb) to global for it
#pragma unroll
            for (int i = 0; i < 4; ++i) inv16[jb * 256 + lane + 64 * i] = Es[lane + 64 * i];
        }
        if (t == 0) break;
        // ---------------- phase B: row tiles  P = X · inv(L16)^T ----------------
        if (wave == 0) {
            // The pivot-chain wave owns the first row tile P(jb+1, jb) and applies it to the NEXT diagonal
            // tile right away, so the next 16-column chain starts right behind barrier 2 (no separate
            // phase C).  P stays in registers for that update: register s of the MFMA result is
            // P(r, k = q + 4s), a valid k-slice for both operands of S -= P P^T.
            const int tn = jb + 1;
            v4d P = {0.0, 0.0, 0.0, 0.0};
            const double* Dx = D + jb * 16 * LDD + tn * 16;
            const int oa = q * 16 + r16;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double aop = Es[oa + 64 * s4];                 // inv(L16)(c = r16, k = 4s+q)
                const double bop = Dx[oel[s4]];                      // X(r16, k = q+4s)
                P = mfma_f64(aop, bop, P);
            }
            const double* Dn = D + tn * 16 * (LDD + 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) S[i] = Dn[osym[i]];
            double* Dp = D + jb * 16 * LDD + tn * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) Dp[oel[i]] = P[i];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) S = mfma_f64(-P[s4], P[s4], S);
        } else if (wave < t) {
            const int tr = jb + 1 + wave;              // waves 1..t-1 take the row tiles below wave 0's
            v4d P = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double aop = Es[(4 * s4 + q) * 16 + r16];                       // inv(L16)(c = r16, k = 4s+q)
                const double bop = D[(jb * 16 + q + 4 * s4) * LDD + tr * 16 + r16];   // X(r16, k = 4s+q)
                P = mfma_f64(aop, bop, P);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) D[(jb * 16 + q + 4 * i) * LDD + tr * 16 + r16] = P[i];
        }
        lds_barrier();                                // barrier 2: panel jb final
    }
    if (wave == 0 && lane == 0 && fail >= 0) {
        if (info[blockIdx.z] == 0) info[blockIdx.z] = fail + 1;
    }
    // ---- write L: 16-byte stores, skipping the 16×16 tiles strictly above the diagonal (the upper
    //      triangles of the diagonal tiles carry scratch values; nothing reads them) ------------------
    for (int idx = tid; idx < BLK * (BLK / 2); idx += DIAG_THREADS) {
        int c = idx / (BLK / 2), rp = idx % (BLK / 2);
        if ((2 * rp) / 16 >= c / 16)
            *reinterpret_cast<v2d*>(A + (size_t)c * ld + 2 * rp) = *reinterpret_cast<const v2d*>(D + c * LDD + 2 * rp);
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
                                                        size_t inv16_bstride, int row0) {
    // row0: first row solved by 16-row group 0 — (k+1)*BLK in the factorisation (everything below
    // the diagonal block); boss_gp_append solves only the block row it rebuilds, or only the δ^T rows
    const int lane = threadIdx.x;
    double* A = Abase + (size_t)blockIdx.z * bstride;
    const double* Lkk = A + (size_t)k * BLK * ((size_t)ld + 1);
    const double* inv16k = inv16base + (size_t)blockIdx.z * inv16_bstride + (size_t)k * (8 * 256);
    double* Brow = A + (size_t)row0 + (size_t)blockIdx.x * 16 + (size_t)k * BLK * ld;
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
#ifndef BOSS_RHS_D
#define BOSS_RHS_D 4
#endif
typedef GemmDirect<2, 2, 4, 4, 4> SyrkG;   // 128×128 tile, fragments streamed from L2, no LDS
typedef GemmDirect<1, 4, 2, 2, BOSS_RHS_D> RhsG;    // 32×128 tile for the δ^T row block and the column updates
constexpr int SYRK_LDS_BYTES = 0;

// acc is INITIALISED from C (its load latency overlaps the operand prologue), updated with
// acc -= P_i P_j^T, and stored back: the epilogue is pure stores.
template <class G>
__device__ __forceinline__ void syrk_tile(double* __restrict__ A, int ld, int k, int R0, int C0, int K = BLK) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / G::WC, wc = wave % G::WC;
    const double* Pi = A + R0 + (size_t)k * BLK * ld;
    const double* Pj = A + C0 + (size_t)k * BLK * ld;
    double* C = A + R0 + (size_t)C0 * ld;
    v4d acc[G::TM][G::TN];
    // rows row_of(m) and row_of(m+1) (m even) are adjacent: one 16-byte access per pair
#pragma unroll
    for (int m = 0; m < G::TM; m += 2)
#pragma unroll
        for (int n = 0; n < G::TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v2d c2 = *reinterpret_cast<const v2d*>(C + G::row_of(wr, m, lane) + (size_t)G::col_of(wc, n, i, lane) * ld);
                acc[m][n][i] = c2[0];
                acc[m + 1][n][i] = c2[1];
            }
    G::template run<-1>(Pi, ld, Pj, ld, K, acc);
#pragma unroll
    for (int m = 0; m < G::TM; m += 2)
#pragma unroll
        for (int n = 0; n < G::TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v2d c2 = {acc[m][n][i], acc[m + 1][n][i]};
                *reinterpret_cast<v2d*>(C + G::row_of(wr, m, lane) + (size_t)G::col_of(wc, n, i, lane) * ld) = c2;
            }
}

// Trailing update behind panel k over the block triangle that starts at block `first` and has m
// square block rows (+ the δ^T row block at first+m == Np/128):  C_ij -= P_i P_j^T.
//   potrf_syrk_kernel   all of it (first = k+1) or, with look-ahead, everything beyond the next
//                       panel (first = k+2) — runs on the side stream, 128×128 tiles;
//   potrf_colupd_kernel only block column k+1 (the next panel), 32×128 tiles so that this short
//                       kernel on the critical path is one small round.
template <int NPAN>
__global__ __launch_bounds__(256, 2) void potrf_syrk_kernel(double* __restrict__ Abase, int ld, size_t bstride, int k,
                                                            int first, int m, int batch1d) {
    constexpr int npan = NPAN;                              // compile-time K keeps the tile loop inside the register budget
    // npan adjacent panels k..k+npan-1 applied in one pass (K = 128·npan): the batched schedule pairs
    // panels so every trailing tile is read and written half as often.
    // batch1d > 0: the whole batch is ONE 1-D grid (tiles of matrix 0, then of matrix 1, ...) and
    // workgroup ids are permuted so that each XCD — workgroups are dealt round-robin over the 8 XCDs,
    // ≈64 resident per XCD — works on 64 CONSECUTIVE tiles of that list (a couple of block rows of one
    // matrix) instead of every 8th: its L2 then holds a handful of panel blocks rather than all of them.
    const int nsq = m * (m + 1) / 2;
    int t = blockIdx.x, b = blockIdx.z;
    if (batch1d > 0) {
        const int Tb = nsq + m, total = Tb * batch1d;
        const int L = blockIdx.x;
        const int idx = (L < (total & ~511)) ? ((L & ~511) + (L & 7) * 64 + ((L >> 3) & 63)) : L;
        b = idx / Tb;
        t = idx - b * Tb;
    }
    double* A = Abase + (size_t)b * bstride;
    if (t < nsq) {
        int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        int j = t - i * (i + 1) / 2;
        syrk_tile<SyrkG>(A, ld, k, (first + i) * BLK, (first + j) * BLK, npan * BLK);
    } else {
        int j = t - nsq;
        syrk_tile<RhsG>(A, ld, k, (first + m) * BLK, (first + j) * BLK, npan * BLK);
    }
}

__global__ __launch_bounds__(256, 2) void potrf_colupd_kernel(double* __restrict__ Abase, int ld, size_t bstride, int k,
                                                              int m, int ncols, int jfirst, int npan, int tskip = 0) {
    // npan = 2: apply the TWO panels k-1, k (K = 256) — the odd steps of the paired look-ahead schedule
    // 32×128 tiles.  ncols = 1: only block column k+1 (look-ahead: the next panel) — 4 strips per
    // 128-row block (m blocks) + one strip of the δ^T rows.  ncols = m: the whole trailing triangle
    // (used for the last steps, where one small launch beats the two-stream choreography).
    // tskip: leave out the first tskip strips (4 = the diagonal block of the first column handled: the split chain
    // updates that block with potrf_diagupd_kernel on the critical path and the rest of the column here, off it)
    double* A = Abase + (size_t)blockIdx.z * bstride;
    int t = blockIdx.x + tskip;
    int j = jfirst;                                           // first trailing block column handled (0 = column k+1)
    if (ncols > 1) {
        // column j has 4*(m-j)+1 strips
        while (t >= 4 * (m - j) + 1) {
            t -= 4 * (m - j) + 1;
            ++j;
        }
    }
    const int nstr = 4 * (m - j);
    const int R0 = (t < nstr) ? (k + 1 + j) * BLK + t * 32 : (k + 1 + m) * BLK;
    if (npan == 2) syrk_tile<RhsG>(A, ld, k - 1, R0, (k + 1 + j) * BLK, 2 * BLK);
    else syrk_tile<RhsG>(A, ld, k, R0, (k + 1 + j) * BLK);
}

// The diagonal block the next step factorises, brought up to date on the critical path:
//   A[k+1,k+1] -= P[k+1,k] P[k+1,k]^T            (npan = 1)
//   A[k+1,k+1] -= P[k+1,k-1] P[k+1,k-1]^T + P[k+1,k] P[k+1,k]^T   (npan = 2, the odd steps of the paired schedule)
// Ten workgroups, one per 32×32 block of the lower block triangle; the contraction is split over the four waves
// (each wave a quarter of the K = 128·npan columns, 8·npan dependent MFMAs per tile instead of 32·npan), partial
// tiles summed through LDS in a fixed order.  A kernel of ≈2 µs between the panel's first solve and the next diagonal block.
__global__ __launch_bounds__(256) void potrf_diagupd_kernel(double* __restrict__ Abase, int ld, size_t bstride, int k, int npan) {
    __shared__ double part[4][4][4][64];                      // [wave][tile][reg][lane]
    double* A = Abase + (size_t)blockIdx.z * bstride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    int bi = 0, b = blockIdx.x;                               // block (bi, bj), bi >= bj, of the 4×4 grid of 32×32 blocks
    while (b > bi) {
        b -= bi + 1;
        ++bi;
    }
    const int bj = b;
    const int r0 = (k + 1) * BLK;
    const int kcols = BLK * npan / 4;                         // this wave's share of the contraction
    const double* P = A + (size_t)((k + 1 - npan) * BLK + wave * kcols) * ld + r0;   // P(r, c) = P[r + c*ld], columns of this wave
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = v4d{0.0, 0.0, 0.0, 0.0};
    for (int s4 = 0; s4 < kcols / 4; ++s4) {
        const double* Pk = P + (size_t)(4 * s4 + q) * ld;
        double fi[2], fj[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            fi[a] = Pk[32 * bi + 16 * a + r16];               // B operand [k][y = row of C]
            fj[a] = Pk[32 * bj + 16 * a + r16];               // A operand [x = column of C][k]
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[a][c] = mfma_f64(-fj[c], fi[a], acc[a][c]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[wave][2 * a + c][i][lane] = acc[a][c][i];
    __syncthreads();
    // wave w finishes tile w = (a, c): C(row = 32bi + 16a + r16, col = 32bj + 16c + q + 4i)
    const int a = wave >> 1, c = wave & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double* dst = A + (size_t)(r0 + 32 * bj + 16 * c + q + 4 * i) * ld + r0 + 32 * bi + 16 * a + r16;
        *dst += ((part[0][wave][i][lane] + part[1][wave][i][lane]) + part[2][wave][i][lane]) + part[3][wave][i][lane];
    }
}

// Block-row variant for boss_gp_append: behind panel k update ONLY block row kb (4 strips of
// 32×128 per column block k+1..kb, the last one being the diagonal block of the rebuilt rows) and
// the δ^T entries of block column kb:  C_{kb,c} -= W_k L_ck^T ,  δ^T_kb -= z_k W_k^T.
__global__ __launch_bounds__(256, 2) void potrf_rowupd_kernel(double* __restrict__ A, int ld, int k, int kb, int Np) {
    const int t = blockIdx.x;
    const int nstr = 4 * (kb - k);
    if (t < nstr) syrk_tile<RhsG>(A, ld, k, kb * BLK + (t & 3) * 32, (k + 1 + (t >> 2)) * BLK);
    else syrk_tile<RhsG>(A, ld, k, Np, kb * BLK);
}

// logdet = 2 Σ_{i<N} log L_ii ,  zz = Σ_{j<N} z_j²   →  scal[2*b], scal[2*b+1]
constexpr int LOGDET_THREADS = 1024;                          // at the end of the chain: every thread takes N/1024 diagonal entries
__global__ __launch_bounds__(LOGDET_THREADS) void potrf_logdet_kernel(const double* __restrict__ Abase, int ld, size_t bstride,
                                                                      int N, int Np, double* __restrict__ scal) {
    const double* A = Abase + (size_t)blockIdx.z * bstride;
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < N; i += LOGDET_THREADS) {
        s0 += log(A[(size_t)i * ld + i]);
        double z = A[(size_t)i * ld + Np];
        s1 += z * z;
    }
    __shared__ double r0[LOGDET_THREADS], r1[LOGDET_THREADS];
    r0[threadIdx.x] = s0;
    r1[threadIdx.x] = s1;
    __syncthreads();
    for (int st = LOGDET_THREADS / 2; st > 0; st >>= 1) {
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

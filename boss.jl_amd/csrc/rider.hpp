// rider.hpp — the first acquisition of a BO iteration riding along the factorisation that precedes it.
//
// One BO iteration is  estimate_parameters! -> maximize_acquisition  on the new posterior (src/bo.jl:30-48); for SamplingAM / GridAM
// (src/acquisition_maximizers/sampling.jl:43-57, grid.jl:52-65) the candidates do not depend on the posterior, so their forward
// substitution  V = L⁻¹ K*  (the `C.U' \ cov(prior, x, x*)` of AbstractGPs' var, src/models/gaussian_process.jl:169-178) can
// travel WITH the update: block row k of V needs only block rows <= k of L,
//     V_k = L_kk⁻¹ ( K*_k − Σ_{j<k} L_kj V_j ),
// and its work grows with k — the complement of the trailing update's front-loaded profile.  From block ≈ 17 of 32 on the update
// is the ten resident workgroups of the panel chain (chain.hpp) and ≈ 246 CUs are empty; this file fills them.
//
// The candidates are treated like extra ROWS of the matrix (V^T = K*^T L^-T, exactly the arithmetic of the panel solve), kept in the
// prediction path's slab layout  V[strip of 32 candidates][row][32]  (= a column-major (candidates × rows) matrix with ld 32):
//   rider_gate_kernel    one wave on the side stream: ends when the word it watches has reached its value (the panel solve of step k
//                        has STARTED: every column < k of the factor and every z entry < k·128 is final);
//   rider_step_kernel    one launch per block k: S_k workgroups (one per 32 candidates) build the residual of block row k — the
//                        partial sums of the previous launch in fixed order minus the product with V_{k-1} — BEFORE the chain has
//                        finished diagonal block k, wait for it inside the kernel, then run the 128-column triangular solve with the
//                        16×16 inverses the chain published (potrf_trsm_kernel's arithmetic and wave split); beside them the E_k
//                        workgroups accumulate block row k+1 over the blocks that are already final, as up to eight K-chunks per
//                        32-candidate strip (GemmDirect 32×128 strips, the column update's tile): enough workgroups to cover the idle
//                        CUs although the product has only M/32 output strips; chunk 0 starts from the K* tile it evaluates in its
//                        accumulator layout;
//   rider_final_kernel   after the update's last kernel: v·z of the last block, the per-block partials in ascending order -> μ, σ².
// Per block: gate + step on the side stream; nothing on the main stream or in the resident kernels knows about it.  Every sum has a
// fixed order: repeated calls are bit-identical.
#pragma once
#include "chain.hpp"
#include "acq_kernels.hpp"

namespace boss {

constexpr int RIDER_MAX_CH = 8;                              // K-chunks per block row (partial-sum slots)
constexpr int RIDER_LDS_BYTES = 24 * 1024;                   // occupancy limiter (unused dynamic LDS on top of 59 KB static): one workgroup per CU, never beside a strips workgroup
                                                             // (90 KB of 160; at 8 KB = two per CU: 1.25 against 1.21 ms per call at 1024 candidates, 1.57 / 1.52 at 2048)
#ifndef BOSS_RIDER_TN
#define BOSS_RIDER_TN 2                                      // 16-row tiles per wave along the block row.  2: 32 candidates × 32 rows per wave, two groups of four waves
#endif                                                       // split K; 4: 32 × 64 per wave (three operand loads per eight MFMAs instead of two per four), four groups of
#ifndef BOSS_RIDER_D                                         // two, ring depth 8 — measured SLOWER (1.38 against 1.21 ms per call at 1024 candidates, 1.64 / 1.52 at 2048:
#define BOSS_RIDER_D (BOSS_RIDER_TN == 2 ? 16 : 8)           // the step is bound by latencies and launch boundaries, not by operand bytes; tools/rider_probe.py)
#endif                                                       // ring depth: K ranges are handed out in units of 4·D columns
typedef GemmDirect<1, 8 / BOSS_RIDER_TN, 2, BOSS_RIDER_TN, BOSS_RIDER_D> RiderG;   // 32 candidates × 128 rows per group of waves, operands streamed from L2
constexpr int RIDER_THREADS = 512;
constexpr int RIDER_GW = RiderG::WC;                         // waves per group
constexpr int RIDER_NG = 8 / RIDER_GW;                       // groups per workgroup: they split K
constexpr int RIDER_KU = 4 * BOSS_RIDER_D;                   // K unit of a group's share (the exact ring needs whole ring passes)
static_assert((BLK / RIDER_NG) % RIDER_KU == 0, "a K = 128 product is split evenly over the groups");
static_assert(BLK % RIDER_KU == 0 && RIDER_MAX_CH % RIDER_NG == 0, "K units, slots per group");

// one 32×128 strip product on this wave's group (grp = wave / RIDER_GW): GemmDirect derives a row offset of 32·grp from the wave
// index, which the A pointer takes back.  A32: the strip's 32 candidates (lda 32).  K: a multiple of RIDER_KU (or 0).
__device__ __forceinline__ void rider_gemm(const double* __restrict__ A32, const double* __restrict__ B, int ldb, int K, v4d (&acc)[2][RiderG::TN], int grp) {
    RiderG::template run<-1, true>(A32 - 32 * grp, 32, B, ldb, K, acc);
}

#ifdef BOSS_CHAIN_TRACE
// device timeline of the rider (tools/rider_timeline.py): per step earliest start, latest end of the in-kernel wait, latest end of
// the solve part, latest end of the accumulate part; row 63: final kernel start / end
__device__ unsigned long long g_rtrace[64 * 4];
#define RTRACE_MIN(step, slot) do { if (threadIdx.x == 0) atomicMin(&g_rtrace[((step) & 63) * 4 + (slot)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define RTRACE_MAX(step, slot) do { if (threadIdx.x == 0) atomicMax(&g_rtrace[((step) & 63) * 4 + (slot)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#else
#define RTRACE_MIN(step, slot) do { } while (0)
#define RTRACE_MAX(step, slot) do { } while (0)
#endif

// lanes < nw watch words w[lane * stride]; the kernel ends when all of them are >= want (or somebody gave the factorisation up)
__global__ __launch_bounds__(64) void rider_gate_kernel(const unsigned long long* __restrict__ w, int nw, int stride,
                                                        unsigned long long want, int* __restrict__ info, unsigned budget, int code) {
    const int lane = threadIdx.x;
    const PollTimer tm(budget);
    for (int i = 0; i < POLL_CAP; ++i) {
        const unsigned long long v = lane < nw ? ld_word(w + (size_t)lane * stride) : ~0ull;
        if (__all(v >= want)) return;
        const int st = tm.check(i, info);
        if (st == 2) return;                                 // (the update is repeated on a simpler schedule: whatever runs behind this gate is discarded)
        if (st == 1) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (lane == 0) {
        st_info(info, INT_MIN);
        note_giveup(9, code);
    }
}

// part: [slot][strip32][128][32] partial residuals of ONE block row;  V: [strip32][Np][32]
__device__ __forceinline__ size_t rider_part_off(int slot, int nstrips, int strip32) { return ((size_t)slot * nstrips + strip32) * (BLK * 32); }

// The K* tile (32 candidates of `strip` × block row krow) in RiderG's accumulator layout: lane (r16, q) of wave wc holds candidates
// 2 r16, 2 r16 + 1 × rows col_of(wc, n, i, lane).  The coordinates are fetched several dimensions at a time (one memory round trip
// per four or eight, not one per dimension: a workgroup of this kernel is a handful of waves on an otherwise idle CU).
__device__ __forceinline__ void rider_kstar_tile(v4d (&acc)[2][RiderG::TN], const double* __restrict__ Xsc, const double* __restrict__ Csc, int d,
                                                 int Np, int N, int Mp, int kern, double amp2, int krow, int strip, int wc, int lane) {
    constexpr int TN = RiderG::TN, PN = RiderG::PN, DU = 8 / PN;   // dimensions per memory round trip
    const int r16 = lane & 15, q = lane >> 4;
    double r2[2][TN][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) r2[m][n][i] = 0.0;
    const double* cp = Csc + (size_t)strip * 32 + 2 * r16;
    const double* xp = Xsc + (size_t)krow * BLK + wc * (TN * 16) + 2 * q;
    for (int d0 = 0; d0 < d; d0 += DU) {
        v2d cv[DU], xv[DU][PN][4];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int dd = min(d0 + u, d - 1);
            cv[u] = *reinterpret_cast<const v2d*>(cp + (size_t)dd * Mp);
#pragma unroll
            for (int pn = 0; pn < PN; ++pn)
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[u][pn][i] = *reinterpret_cast<const v2d*>(xp + (size_t)dd * Np + 32 * pn + 8 * i);
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            if (d0 + u < d) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < TN; ++n) {
                            const double df = xv[u][n >> 1][i][n & 1] - cv[u][m];
                            r2[m][n][i] = __builtin_fma(df, df, r2[m][n][i]);
                        }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = krow * BLK + RiderG::col_of(wc, n, i, lane);
                acc[m][n][i] = row < N ? amp2 * kappa_r2(kern, r2[m][n][i]) : 0.0;
            }
}

struct RiderStep {
    const double* A;                                         // the factor being built (column-major, ld)
    const double* inv16;                                     // inverses of the diagonal 16×16 tiles, as the chain publishes them
    double* V;                                               // [strip32][Np][32]
    double* part_in;                                         // [slot][strip32][128][32]: partial sums of block row k (left by the previous launch)
    double* part_out;                                        // ... of block row k+1 (written by this launch's E workgroups): the other of two buffers
    double *part_ss, *part_z;                                // [block][Mp]
    const double *Xsc, *Csc;
    const unsigned long long* sig;                           // the context's signal block
    int* info;
    unsigned long long want;                                 // value the watched words must reach before block k is solved
    int ld, Np, N, Mp, d, kern, k, nblk, nstrips, nslots, cb, nchunks;
    int efirst;                                              // E workgroups first in the grid (one per CU before the S workgroups double up)
#ifdef BOSS_EXPERIMENTS
    int exp;                                                 // timing experiments: bit 0 skips the E products, bit 1 the fold products, bit 2 sleeps between E products
#endif
    unsigned budget;
    double amp2;
};

// Step k of the rider, ONE launch on the side stream, ordered behind step k-1 by the stream alone:
//   workgroups [0, nstrips)         S_k for one strip of 32 candidates: the residual of block row k = the partial sums E_{k-1} left
//                                   (slot order) − [V_{k-2}; V_{k-1}]-product with L[k, k-2 : k] (K = 256), staged in LDS; v·z of block
//                                   k-2; THEN the workgroup waits for diagonal block k (k < nblk-1: the eight strips of step k have
//                                   delivered block (k+1, k); last block: the chain's panel word) and solves: V_k, Σv².  Everything in
//                                   front of the wait runs while the chain is still factoring block k.
//   the others (k < nblk-1)         E_k: partial sums of block row k+1 over blocks j <= k-2 (K-chunks of cb blocks; chunk 0 starts
//                                   from the K* tile); S_{k+1} adds the last two blocks.
// What a launch reads is final when it starts: S_{k-1} (previous launch) saw the strips of step k-1 done, i.e. block (k, k-1) is
// out, the column update of step k-2 had started (its critical strips fed them) and with it the panel solve of step k-2 had ended:
// columns <= k-2 of the factor and the z entries of block k-2 are final.  Only diagonal block k itself is waited for in the kernel.
__global__ __launch_bounds__(RIDER_THREADS) void rider_step_kernel(RiderStep p) {
    extern __shared__ double rider_lds_unused[];
    typedef RiderG G;
    __shared__ double Rl[BLK * 32];                          // residual of this strip, [column of the block][candidate]
    __shared__ v4d xs[2][TRSM_NA][64];
    __shared__ int ready[2];
    __shared__ double red[2][2][16], zred[16][32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / RIDER_GW, wc = wave % RIDER_GW;   // group of waves (one 32×128 tile, a share of K), column part inside the tile
    constexpr int TN = G::TN, NG = RIDER_NG;
    const int k = p.k, ld = p.ld, Np = p.Np, nstrips = p.nstrips;
    v4d acc[2][TN];
    auto rl_at = [&](int n, int i) { return reinterpret_cast<v2d*>(Rl + G::col_of(wc, n, i, lane) * 32 + G::row_of(0, 0, lane)); };
    // the groups' tiles summed through Rl in group order 1, 2, …, (group 0 holds its own until the end: the caller adds it last)
    auto fold_groups = [&]() {
#pragma unroll
        for (int g = 1; g < NG; ++g) {
            if (grp == g) {
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v2d o = v2d{acc[0][n][i], acc[1][n][i]};
                        if (g > 1) o += *rl_at(n, i);
                        *rl_at(n, i) = o;
                    }
            }
            __syncthreads();
        }
    };
    RTRACE_MIN(k, 0);
    const int nE = (int)gridDim.x - nstrips;
    if (p.efirst ? (int)blockIdx.x < nE : (int)blockIdx.x >= nstrips) {
        // ---------------- E_k: block row k+1, one (strip, chunk) item per workgroup; chunk ch takes the blocks
        // [ch·cb, min((ch+1)·cb, k-1)), its columns dealt to the groups in units of RIDER_KU (summed through LDS in group order) —
        // strips fastest: the workgroups of a chunk share their slice of L
        // (measured and removed: chunk c on XCD c, so that every slice of L[k+1, ·] crosses the fabric once instead of eight times —
        // 1.33–1.35 against 1.25–1.29 ms per call at 1024 candidates: the last chunk is shorter than the others and its XCD idles)
        const int item = p.efirst ? (int)blockIdx.x : (int)blockIdx.x - nstrips;
        const int strip = item % nstrips, ch = item / nstrips;
        const int krow = k + 1, j0 = ch * p.cb, j1 = min(j0 + p.cb, k - 1);
        const int units = max(j1 - j0, 0) * (BLK / RIDER_KU);
        const int ua = units * grp / NG, uz = units * (grp + 1) / NG;
        if (ch == 0 && grp == 0) rider_kstar_tile(acc, p.Xsc, p.Csc, p.d, Np, p.N, p.Mp, p.kern, p.amp2, krow, strip, wc, lane);
        else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
        }
#ifdef BOSS_EXPERIMENTS
        if (!(p.exp & 1))
#endif
        if (uz > ua) {
            const size_t c0 = (size_t)j0 * BLK + (size_t)ua * RIDER_KU;
            rider_gemm(p.V + (size_t)strip * Np * 32 + c0 * 32, p.A + (size_t)krow * BLK + c0 * ld, ld, (uz - ua) * RIDER_KU, acc, grp);
        }
        if (units > 0) fold_groups();                        // (uniform over the workgroup)
        if (grp != 0) return;
        double* P = p.part_out + rider_part_off(ch, nstrips, strip);
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v2d o = v2d{acc[0][n][i], acc[1][n][i]};
                if (units > 0) o += *rl_at(n, i);
                *reinterpret_cast<v2d*>(P + (size_t)G::col_of(wc, n, i, lane) * 32 + G::row_of(0, 0, lane)) = o;
            }
        RTRACE_MAX(k, 3);
        return;
    }
    // ---------------- S_k
    const int strip = p.efirst ? (int)blockIdx.x - nE : (int)blockIdx.x;
    const int r16 = lane & 15, q = lane >> 4;
    if (tid == 0) {
        *(volatile lds_int_t*)&ready[0] = 0;
        *(volatile lds_int_t*)&ready[1] = 0;
    }
    // (1) residual of block row k in the strip-GEMM layout, the groups side by side: group g takes the partial sums of its
    // RIDER_MAX_CH / NG slots and its share of the columns of blocks k-2, k-1 (k = 1: block 0); the sum goes through Rl in group order
    if (p.nslots == 0) {
        if (grp == 0) {
            rider_kstar_tile(acc, p.Xsc, p.Csc, p.d, Np, p.N, p.Mp, p.kern, p.amp2, k, strip, wc, lane);
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) *rl_at(n, i) = v2d{acc[0][n][i], acc[1][n][i]};
        }
    } else {
        constexpr int SPG = RIDER_MAX_CH / NG;
        const int s0 = SPG * grp;                            // this group's slots: [s0, min(s0 + SPG, nslots))
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
        if (s0 < p.nslots) {
            const double* P0 = p.part_in + rider_part_off(0, nstrips, strip) + G::row_of(0, 0, lane);
            const size_t sstride = rider_part_off(1, nstrips, 0);
            v2d t[SPG][TN][4];
#pragma unroll
            for (int u = 0; u < SPG; ++u) {                  // the group's slots in one memory round trip, added in slot order
                const double* Ps = P0 + (size_t)min(s0 + u, p.nslots - 1) * sstride;
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[u][n][i] = *reinterpret_cast<const v2d*>(Ps + (size_t)G::col_of(wc, n, i, lane) * 32);
            }
#pragma unroll
            for (int u = 0; u < SPG; ++u) {
                if (u == 0) {
#pragma unroll
                    for (int n = 0; n < TN; ++n)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[0][n][i] = t[0][n][i][0];
                            acc[1][n][i] = t[0][n][i][1];
                        }
                } else if (s0 + u < p.nslots) {
#pragma unroll
                    for (int n = 0; n < TN; ++n)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[0][n][i] += t[u][n][i][0];
                            acc[1][n][i] += t[u][n][i][1];
                        }
                }
            }
        }
        const int c0 = max(k - 2, 0) * BLK, share = min(k, 2) * BLK / NG;   // this group's columns: [c0 + grp·share, + share)
#ifdef BOSS_EXPERIMENTS
        if (!(p.exp & 2))
#endif
        rider_gemm(p.V + (size_t)strip * Np * 32 + (size_t)(c0 + grp * share) * 32, p.A + (size_t)k * BLK + (size_t)(c0 + grp * share) * ld, ld, share, acc, grp);
        fold_groups();
        if (grp == 0) {
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) *rl_at(n, i) = *rl_at(n, i) + v2d{acc[0][n][i], acc[1][n][i]};
        }
    }
    // (2) v · z of block k-2 (its z entries are final, see above)
    double* Vst = p.V + (size_t)strip * Np * 32;             // + row * 32 + candidate
    if (k > 1) {
        const int c = tid & 31, g = tid >> 5;                // candidate, group of eight rows
        const double* vp = Vst + (size_t)((k - 2) * BLK + g * 8) * 32 + c;
        const double* zp = p.A + (size_t)((k - 2) * BLK + g * 8) * ld + Np;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) s = __builtin_fma(vp[(size_t)j * 32], zp[(size_t)j * ld], s);
        zred[g][c] = s;
    }
    // (3) diagonal block k is out
    if (wave == 0) {
        const int nw = k + 1 < p.nblk ? 8 : 1;
        const unsigned long long* w = p.sig + (k + 1 < p.nblk ? SIGW_PROG : SIGW_PANEL);
        const int stride = k + 1 < p.nblk ? SIGW_PROG_STRIDE : 1;
        const PollTimer tm(p.budget);
        bool ok = false;
        for (int it = 0; it < POLL_CAP; ++it) {
            const unsigned long long v = lane < nw ? ld_word(w + (size_t)lane * stride) : ~0ull;
            if (__all(v >= p.want)) {
                ok = true;
                break;
            }
            const int st = tm.check(it, p.info);
            if (st == 2) {
                ok = true;                                   // (given up elsewhere: finish without waiting, the results are discarded)
                break;
            }
            if (st == 1) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && lane == 0) {
            st_info(p.info, INT_MIN);
            note_giveup(10, k);
        }
    }
    __syncthreads();
    RTRACE_MAX(k, 1);
    if (k > 1 && tid < 32) {
        double s = zred[0][tid];
#pragma unroll
        for (int g = 1; g < 16; ++g) s += zred[g][tid];
        p.part_z[(size_t)(k - 2) * p.Mp + strip * 32 + tid] = s;
    }
    // (4) the 128-column triangular solve, potrf_trsm_kernel's arithmetic: waves (0, 1) take candidates 0..15, waves (2, 3) 16..31;
    // the first wave of a pair solves column tiles 0..4 and hands them over through LDS, the second finishes tiles 5..7.  The chain's
    // stores are read with agent-scope loads: this kernel started before they were written.
    const int half = (wave >> 1) & 1, role = wave & 1;       // (waves 4..7 sit this part out)
    const double* Lkk = p.A + (size_t)k * BLK * ((size_t)ld + 1);
    const double* inv16k = p.inv16 + (size_t)k * (8 * 256);
    auto lval = [&](int jb, int m, int s) { return ld_sc1(Lkk + (size_t)(m * 16 + 4 * s + q) * ld + jb * 16 + r16); };
    auto ival = [&](int jb, int s) { return ld_sc1(inv16k + jb * 256 + (4 * s + q) * 16 + r16); };
    auto rload = [&](int jb, v4d& t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = Rl[(jb * 16 + q + 4 * i) * 32 + half * 16 + r16];
    };
    double* Vs = Vst + half * 16;
    auto vstore = [&](int jb, const v4d& t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) Vs[(size_t)(k * BLK + jb * 16 + q + 4 * i) * 32 + r16] = t[i];
    };
    double ssq = 0.0;
    if (wave >= 4) {
    } else if (role == 0) {
        v4d a5[TRSM_NA];
        double lv[TRSM_NA][TRSM_NA][4], iv[TRSM_NA][4];
#pragma unroll
        for (int jb = 0; jb < TRSM_NA; ++jb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) iv[jb][s] = ival(jb, s);
#pragma unroll
            for (int m = 0; m < jb; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) lv[jb][m][s] = lval(jb, m, s);
        }
#pragma unroll
        for (int jb = 0; jb < TRSM_NA; ++jb) rload(jb, a5[jb]);
#pragma unroll
        for (int jb = 0; jb < TRSM_NA; ++jb) {
#pragma unroll
            for (int m = 0; m < jb; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) a5[jb] = mfma_f64(-lv[jb][m][s], a5[m][s], a5[jb]);
            v4d nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) nw = mfma_f64(iv[jb][s], a5[jb][s], nw);
            a5[jb] = nw;
            xs[half][jb][lane] = nw;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile lds_int_t*)&ready[half] = jb + 1;
        }
#pragma unroll
        for (int jb = 0; jb < TRSM_NA; ++jb) {
            vstore(jb, a5[jb]);
#pragma unroll
            for (int i = 0; i < 4; ++i) ssq = __builtin_fma(a5[jb][i], a5[jb][i], ssq);
        }
    } else {
        constexpr int NB = 8 - TRSM_NA;
        v4d a3[NB];
        double lu[TRSM_NA][NB][4], lo[NB][NB][4], iv[NB][4];
#pragma unroll
        for (int m = 0; m < TRSM_NA; ++m)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) lu[m][j][s] = lval(TRSM_NA + j, m, s);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int s = 0; s < 4; ++s) iv[j][s] = ival(TRSM_NA + j, s);
#pragma unroll
            for (int m = 0; m < j; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) lo[j][m][s] = lval(TRSM_NA + j, TRSM_NA + m, s);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) rload(TRSM_NA + j, a3[j]);
#pragma unroll
        for (int m = 0; m < TRSM_NA; ++m) {
#pragma unroll 1
            while (*(volatile lds_int_t*)&ready[half] <= m) __builtin_amdgcn_s_sleep(1);   // (its pair wave never waits for this one)
            asm volatile("" ::: "memory");
            const v4d x = xs[half][m][lane];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NB; ++j) a3[j] = mfma_f64(-lu[m][j][s], x[s], a3[j]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int m = 0; m < j; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) a3[j] = mfma_f64(-lo[j][m][s], a3[m][s], a3[j]);
            v4d nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) nw = mfma_f64(iv[j][s], a3[j][s], nw);
            a3[j] = nw;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            vstore(TRSM_NA + j, a3[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) ssq = __builtin_fma(a3[j][i], a3[j][i], ssq);
        }
    }
    // Σ v² of this block per candidate: the four column groups of a wave, then the two waves of the pair
    ssq += __shfl_xor(ssq, 16);
    ssq += __shfl_xor(ssq, 32);
    if (wave < 4 && lane < 16) red[half][role][lane] = ssq;
    __syncthreads();
    if (tid < 32) p.part_ss[(size_t)k * p.Mp + strip * 32 + tid] = red[tid >> 4][0][tid & 15] + red[tid >> 4][1][tid & 15];
    RTRACE_MAX(k, 2);
}

// The rider's tail in ONE launch behind the last step: μ, σ² (unclipped, as the prediction kernels leave them), the EI × feasibility
// epilogue of acq_epilogue_kernel (expected_improvement.jl:68-101) and the first-index arg-max.  One workgroup per 32 candidates,
// eight threads per candidate — thread (candidate c, g) takes every eighth block of the per-block partial sums BEFORE the
// workgroup waits for the update's last kernel (the z entries of the last two blocks are final then), afterwards the g-th sixteen
// rows of those blocks' v·z.  Each workgroup leaves its (max, index) pair write-through; the one that counts itself in last reduces
// the pairs and answers the host through mapped memory (value, index, the factorisation's flag, then the sequence word it polls).
struct RiderFinal {
    const double *A, *V, *part_ss, *part_z, *mean_s;
    double *mu, *var, *acq;                                  // device arrays (M)
    const unsigned char* mask;
    const int* info;
    double* host_res;                                        // mapped: {max, arg-max, sequence word, flag}
    const unsigned long long* gate;                          // sig[SIGW_GATE] >= want: the update's last kernel has started
    unsigned long long want, done_after, res_seq;
    unsigned long long* done;                                // device counter of finished workgroups (grows by the grid size per call)
    double* pairs;                                           // [workgroup][2]
    int ld, Np, nblk, M, Mp;
    unsigned budget;
    double amp2;
    EiPar par;
};
__global__ __launch_bounds__(256) void rider_final_kernel(RiderFinal p) {
    RTRACE_MIN(63, 0);
    __shared__ double sz[2][8][32], sp[2][8][32], bvs[32];
    __shared__ long bis[32];
    __shared__ int lastflag;
    constexpr long NONE = 0x7fffffffffffffffL;
    const int tid = threadIdx.x, cl = tid & 31, g = tid >> 5, c = blockIdx.x * 32 + cl;
    const int nblk = p.nblk, Mp = p.Mp, ld = p.ld;
    double ss = 0.0, zz = 0.0;
    for (int k = g; k < nblk; k += 8) ss += p.part_ss[(size_t)k * Mp + c];
    for (int k = g; k < nblk - 2; k += 8) zz += p.part_z[(size_t)k * Mp + c];
    sp[0][g][cl] = ss;
    sp[1][g][cl] = zz;
    if (tid < 64) {                                          // (one wave)
        const PollTimer tm(p.budget);
        bool ok = false;
        for (int it = 0; it < POLL_CAP; ++it) {
            if (ld_word(p.gate) >= p.want) {
                ok = true;
                break;
            }
            const int st = tm.check(it, p.info);
            if (st == 2) {
                ok = true;                                   // (given up elsewhere: the host discards what follows)
                break;
            }
            if (st == 1) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && tid == 0) {
            st_info(const_cast<int*>(p.info), INT_MIN);
            note_giveup(11, 0);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the z entries were written while this kernel was already running
    const double* vp = p.V + (size_t)blockIdx.x * p.Np * 32 + cl;
#pragma unroll
    for (int b = 0; b < 2; ++b) {                             // blocks nblk-2, nblk-1 (the steps behind them never ran)
        const int kb = nblk - 2 + b;
        const double* vq = vp + (size_t)(kb * BLK + g * 16) * 32;
        const double* zq = p.A + (size_t)(kb * BLK + g * 16) * ld + p.Np;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s = __builtin_fma(vq[(size_t)j * 32], ld_sc1(zq + (size_t)j * ld), s);
        sz[b][g][cl] = s;
    }
    __syncthreads();
    if (g == 0) {
        double bv = -INFINITY;
        long bi = NONE;
        if (c < p.M) {
            double s = sp[0][0][cl], z = sp[1][0][cl], z0 = sz[0][0][cl], z1 = sz[1][0][cl];
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                s += sp[0][q][cl];
                z += sp[1][q][cl];
                z0 += sz[0][q][cl];
                z1 += sz[1][q][cl];
            }
            z = (z + z0) + z1;
            const double mu_c = (p.mean_s ? p.mean_s[c] : 0.0) + z, var_c = p.amp2 - s + PREDICT_JITTER;
            p.mu[c] = mu_c;
            p.var[c] = var_c;
            double a = ei_value_one(mu_c, var_c, p.par);
            if (p.mask && !p.mask[c]) a = 0.0;
            p.acq[c] = a;
            bv = a;
            bi = c;
        }
        bvs[cl] = bv;
        bis[cl] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        double bv = bvs[0];
        long bi = bis[0];
        for (int q = 1; q < 32; ++q)
            if (bis[q] != NONE && (bi == NONE || better(bvs[q], bis[q], bv, bi))) {
                bv = bvs[q];
                bi = bis[q];
            }
        st_sc1(p.pairs + 2 * (size_t)blockIdx.x, bv);
        st_sc1(p.pairs + 2 * (size_t)blockIdx.x + 1, __longlong_as_double((long long)bi));
        drain_stores();
        const unsigned long long before = __hip_atomic_fetch_add(as_global(p.done), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *(volatile lds_int_t*)&lastflag = (before + 1 == p.done_after) ? 1 : 0;
    }
    __syncthreads();
    RTRACE_MAX(63, 2);
    if (*(volatile lds_int_t*)&lastflag == 0 || tid >= 64) return;
    // ---- the last workgroup: first-index arg-max over the workgroups' pairs (one wave; at most 256 pairs)
    // the factorisation's flag as it stands behind the rider's last wait (a wait of the rider that gave up after the update's own
    // kernels had passed leaves INT_MIN there, and the host discards the rider's numbers): fetched beside the pairs
    const int info_now = tid == 63 ? ld_info_fresh(p.info) : 0;
    double bv = -INFINITY;
    long bi = NONE;
    for (int w = tid; w < (int)gridDim.x; w += 64) {
        const double v = ld_sc1(p.pairs + 2 * (size_t)w);
        const long i = (long)__double_as_longlong(ld_sc1(p.pairs + 2 * (size_t)w + 1));
        if (i != NONE && (bi == NONE || better(v, i, bv, bi))) {
            bv = v;
            bi = i;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const long oi = ((long)__shfl_xor((int)(bi >> 32), off) << 32) | (unsigned)__shfl_xor((int)(bi & 0xffffffffL), off);
        if (oi != NONE && (bi == NONE || better(ov, oi, bv, bi))) {
            bv = ov;
            bi = oi;
        }
    }
    const int info63 = __shfl(info_now, 63);
    if (tid == 0) {
        p.host_res[0] = bv;
        reinterpret_cast<long*>(p.host_res)[1] = bi;
        reinterpret_cast<int*>(p.host_res + 3)[0] = info63;
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.host_res) + 2, p.res_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    RTRACE_MAX(63, 3);
}

}  // namespace boss

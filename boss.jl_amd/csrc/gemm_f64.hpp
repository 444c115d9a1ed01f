// gemm_f64.hpp — fp64 MFMA tile core: C(BM×BN) = Σ_k A(r,k)·B(c,k)   ("NT": both operands
// are column-major with the contraction index as the COLUMN index: A(r,k) = A[r + k*lda],
// B(c,k) = B[c + k*ldb]).  256 threads = 4 waves arranged WR×WC; each wave owns TM×TN
// 16×16 MFMA tiles.  Operands are staged global → registers → LDS (double-buffered, KB=16
// deep), fragments are read back with conflict-free ds_read_b64 (row stride padded by 16
// doubles = 32 banks, see MI355X_MICROARCH.md §LDS).
//
// Accumulator orientation ("transposed trick"): the B fragment is fed as the MFMA's A
// operand and vice versa, so result register i of lane l is
//     C(row = m*16 + (l&15), col = n*16 + (l>>4) + 4*i)      (within the wave's sub-tile)
// i.e. consecutive lanes hold consecutive ROWS of a column-major C: global stores/loads of C
// are 128-byte contiguous per 16 lanes.
#pragma once
#include "common.hpp"

namespace boss {

template <int WR_, int WC_, int TM_, int TN_>
struct GemmNT {
    static constexpr int WR = WR_, WC = WC_, TM = TM_, TN = TN_;
    static_assert(WR * WC == 4, "4 waves per workgroup");
    static constexpr int BM = WR * TM * 16, BN = WC * TN * 16, KB = 16;
    static_assert(BM % 32 == 0 && BN % 32 == 0, "tile dims must be multiples of 32");
    static constexpr int SA = BM + 16, SB = BN + 16;          // LDS row strides (doubles)
    static constexpr int STAGE = KB * (SA + SB);              // doubles per pipeline stage
    static constexpr int LDS_DOUBLES = 2 * STAGE;
    static constexpr int NA = BM / 32, NB = BN / 32;          // 16-byte chunks / thread / stage

    struct Stage {
        v2d a[NA];
        v2d b[NB];
    };

    __device__ static __forceinline__ void gload(Stage& s, const double* __restrict__ A, int lda,
                                                 const double* __restrict__ B, int ldb, int k0, int tid) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            int q = tid + 256 * p;
            int rp = q % (BM / 2), kk = q / (BM / 2);
            s.a[p] = *reinterpret_cast<const v2d*>(A + (size_t)(k0 + kk) * lda + 2 * rp);
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            int q = tid + 256 * p;
            int cp = q % (BN / 2), kk = q / (BN / 2);
            s.b[p] = *reinterpret_cast<const v2d*>(B + (size_t)(k0 + kk) * ldb + 2 * cp);
        }
    }

    __device__ static __forceinline__ void sstore(const Stage& s, double* __restrict__ buf, int tid) {
        double* As = buf;
        double* Bs = buf + KB * SA;
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            int q = tid + 256 * p;
            int rp = q % (BM / 2), kk = q / (BM / 2);
            *reinterpret_cast<v2d*>(As + kk * SA + 2 * rp) = s.a[p];
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            int q = tid + 256 * p;
            int cp = q % (BN / 2), kk = q / (BN / 2);
            *reinterpret_cast<v2d*>(Bs + kk * SB + 2 * cp) = s.b[p];
        }
    }

    // 4 k-substeps of 4 on one staged KB=16 slab.
    __device__ static __forceinline__ void compute(const double* __restrict__ buf, v4d (&acc)[TM][TN],
                                                   int wr, int wc, int lane) {
        const double* As = buf + wr * (TM * 16) + (lane & 15);
        const double* Bs = buf + KB * SA + wc * (TN * 16) + (lane & 15);
        const int kq = lane >> 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = 4 * s + kq;
            double af[TM], bf[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) af[m] = As[kk * SA + m * 16];
#pragma unroll
            for (int n = 0; n < TN; ++n) bf[n] = Bs[kk * SB + n * 16];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[m][n] = mfma_f64(bf[n], af[m], acc[m][n]);
        }
    }

    // acc += A(:,0:K)·B(:,0:K)^T.  K must be a multiple of 16 (K == 0 is a no-op).
    // Every thread of the workgroup must call this (it contains barriers).
    __device__ static __forceinline__ void run(const double* __restrict__ A, int lda,
                                               const double* __restrict__ B, int ldb, int K,
                                               v4d (&acc)[TM][TN], double* __restrict__ lds) {
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        const int wr = wave / WC, wc = wave % WC;
        const int KT = K / KB;
        if (KT == 0) return;
        Stage st;
        gload(st, A, lda, B, ldb, 0, tid);
        __syncthreads();                       // previous users of `lds` are done
        sstore(st, lds, tid);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const bool more = (kt + 1 < KT);
            if (more) gload(st, A, lda, B, ldb, (kt + 1) * KB, tid);
            compute(lds + (kt & 1) * STAGE, acc, wr, wc, lane);
            if (more) {
                sstore(st, lds + ((kt + 1) & 1) * STAGE, tid);
                __syncthreads();
            }
        }
    }

    // Same, but the B operand is already resident in LDS as Bl[k*ldbl + c] (k = 0..K-1,
    // c = 0..BN-1, ldbl ≡ 16 mod 32 keeps fragment reads conflict-free); only A is staged.
    __device__ static __forceinline__ void run_Blds(const double* __restrict__ A, int lda,
                                                    const double* __restrict__ Bl, int ldbl, int K,
                                                    v4d (&acc)[TM][TN], double* __restrict__ lds) {
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        const int wr = wave / WC, wc = wave % WC;
        const int KT = K / KB;
        if (KT == 0) return;
        constexpr int ASTAGE = KB * SA;
        v2d a[NA];
        auto ga = [&](int k0) {
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                int q = tid + 256 * p;
                int rp = q % (BM / 2), kk = q / (BM / 2);
                a[p] = *reinterpret_cast<const v2d*>(A + (size_t)(k0 + kk) * lda + 2 * rp);
            }
        };
        auto sa = [&](double* buf) {
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                int q = tid + 256 * p;
                int rp = q % (BM / 2), kk = q / (BM / 2);
                *reinterpret_cast<v2d*>(buf + kk * SA + 2 * rp) = a[p];
            }
        };
        ga(0);
        __syncthreads();
        sa(lds);
        __syncthreads();
        const int kq = lane >> 4;
        for (int kt = 0; kt < KT; ++kt) {
            const bool more = (kt + 1 < KT);
            if (more) ga((kt + 1) * KB);
            const double* As = lds + (kt & 1) * ASTAGE + wr * (TM * 16) + (lane & 15);
            const double* Bs = Bl + (size_t)(kt * KB) * ldbl + wc * (TN * 16) + (lane & 15);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kk = 4 * s + kq;
                double af[TM], bf[TN];
#pragma unroll
                for (int m = 0; m < TM; ++m) af[m] = As[kk * SA + m * 16];
#pragma unroll
                for (int n = 0; n < TN; ++n) bf[n] = Bs[kk * ldbl + n * 16];
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n) acc[m][n] = mfma_f64(bf[n], af[m], acc[m][n]);
            }
            if (more) {
                sa(lds + ((kt + 1) & 1) * ASTAGE);
                __syncthreads();
            }
        }
    }

    // element coordinates of accumulator register (m, n, i) of this thread inside the BM×BN tile
    __device__ static __forceinline__ int row_of(int wr, int m, int lane) { return wr * (TM * 16) + m * 16 + (lane & 15); }
    __device__ static __forceinline__ int col_of(int wc, int n, int i, int lane) { return wc * (TN * 16) + n * 16 + (lane >> 4) + 4 * i; }
};

// ------------------------------------------------------------------------------------------
// GemmDirect — same contraction as GemmNT, but the MFMA fragments are streamed straight from
// global/L2 into a D-deep register ring: no LDS, no barriers, waves drift freely.  fp64 MFMA is
// slow relative to memory (64 cycles per 2 KFLOP), so L2 feeds it comfortably: measured 67 TF at
// one 128x128 workgroup per CU and 70-72 TF at two (tools/gemm_probe2.hip) vs 56 / 67 TF LDS-staged.
// A(r,k) = A[r + k*lda], B(c,k) = B[c + k*ldb]; SIGN = -1 accumulates  acc -= A·B^T.
//
// Row/column interleave: the wave's MFMA tiles are paired — tile 2p holds the EVEN rows and tile
// 2p+1 the ODD rows of a 32-row group (same for columns) — so ONE 16-byte load per lane feeds two
// fragments (half the load instructions, 16-byte L2 accesses), and C is read/written 16 bytes per
// lane.  Everything outside goes through row_of / col_of, so the permutation is invisible.
//   accumulator register (m, n, i) of lane l  <->  C(row_of(wr, m, l), col_of(wc, n, i, l))
// ------------------------------------------------------------------------------------------
// Ring loads are inline asm so that hipcc can neither sink them towards their use (it does, to cut
// register pressure, which collapses the prefetch distance) nor insert its own conservative
// s_waitcnt; the waits are hand-counted (loads complete in order: waiting for vmcnt(n) retires
// everything but the n most recent vector-memory operations).
// REQUIREMENTS of the hand-counted scheme (each one was a real bug during bring-up):
//   * uniform (scalar) control flow around the ring: under an EXEC-masked loop a fully masked-off
//     wave still executes the instruction stream, its loads do not issue and do not bump vmcnt —
//     the wave index is therefore taken through readfirstlane;
//   * no register spills in kernels that use it (agpr_count == 0, no scratch): hipcc would copy a
//     ring register before its load has landed (tests/test_abi_and_host.py checks the metadata).
#ifdef BOSS_RING_PLAIN   // debugging aid: compiler-managed loads and waits
__device__ __forceinline__ void gld16(v2d& dst, const double* p) { dst = *reinterpret_cast<const v2d*>(p); }
template <int IMM>
__device__ __forceinline__ void gld16s(v2d& dst, const double* base, unsigned voff) {
    dst = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(base) + voff + IMM);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {}
#else
__device__ __forceinline__ void gld16(v2d& dst, const double* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// SGPR base + 32-bit per-lane byte offset + immediate: the wave-uniform part of the address (which k
// slice) lives on the scalar unit, the per-lane part is loop-invariant — no VALU address arithmetic
// between the MFMAs.
template <int IMM>
__device__ __forceinline__ void gld16s(v2d& dst, const double* base, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#endif

template <int WR_, int WC_, int TM_, int TN_, int D_>
struct GemmDirect {
    static constexpr int WR = WR_, WC = WC_, TM = TM_, TN = TN_, D = D_;
    static constexpr int NWAVES = WR * WC, NTHREADS = 64 * WR * WC;   // 4 or 8 waves per workgroup
    static_assert(TM % 2 == 0 && TN % 2 == 0, "tiles are paired");
    static constexpr int BM = WR * TM * 16, BN = WC * TN * 16;
    static constexpr int PM = TM / 2, PN = TN / 2;

    __device__ static __forceinline__ int row_of(int wr, int m, int lane) {
        return wr * (TM * 16) + (m >> 1) * 32 + 2 * (lane & 15) + (m & 1);
    }
    __device__ static __forceinline__ int col_of(int wc, int n, int i, int lane) {
        return wc * (TN * 16) + (n >> 1) * 32 + 2 * ((lane >> 4) + 4 * i) + (n & 1);
    }

    // slot S of a ring that is no longer refilled: wait until slots 0 .. S have landed (loads complete in order), consume it, go on
    template <int SIGN, int S>
    __device__ static __forceinline__ void ring_tail(v2d (&af)[D][PM], v2d (&bf)[D][PN], v4d (&acc)[TM][TN]) {
        if constexpr (S < D) {
            wait_vmcnt<(PM + PN) * (D - 1 - S)>();
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    const double b = bf[S][t >> 1][t & 1];
                    acc[m][t] = mfma_f64(SIGN < 0 ? -b : b, af[S][m >> 1][m & 1], acc[m][t]);
                }
            __builtin_amdgcn_sched_barrier(0);
            ring_tail<SIGN, S + 1>(af, bf, acc);
        }
    }

    // EXACT: the caller guarantees that K/4 is a multiple of D (and >= D): see the ring tail below
    template <int SIGN, bool EXACT = false>
    __device__ static __forceinline__ void run(const double* __restrict__ A, int lda, const double* __restrict__ B,
                                               int ldb, int K, v4d (&acc)[TM][TN]) {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: uniform control flow around the ring
        const int wr = wave / WC, wc = wave % WC;
        // per-lane BYTE offsets (loop-invariant, < 2^31: (lane>>4)·ld·8 ≤ 3·ld·8)
        const unsigned aoff = 8u * (unsigned)(wr * (TM * 16) + 2 * (lane & 15) + (lane >> 4) * lda);
        const unsigned boff = 8u * (unsigned)(wc * (TN * 16) + 2 * (lane & 15) + (lane >> 4) * ldb);
        const int n = K / 4;
        if (n == 0) return;
        v2d af[D][PM], bf[D][PN];
        static_assert((PM + PN) * (D - 1) <= 63, "vmcnt is a 6-bit counter");
        static_assert(PM <= 4 && PN <= 4, "immediate offsets are written out for up to 4 pairs");
#define BOSS_LOAD_A(slot, Ak)                                      \
    do {                                                           \
        gld16s<0>(af[slot][0], Ak, aoff);                          \
        if (PM > 1) gld16s<256>(af[slot][PM > 1 ? 1 : 0], Ak, aoff); \
        if (PM > 2) gld16s<512>(af[slot][PM > 2 ? 2 : 0], Ak, aoff); \
        if (PM > 3) gld16s<768>(af[slot][PM > 3 ? 3 : 0], Ak, aoff); \
    } while (0)
#define BOSS_LOAD_B(slot, Bk)                                      \
    do {                                                           \
        gld16s<0>(bf[slot][0], Bk, boff);                          \
        if (PN > 1) gld16s<256>(bf[slot][PN > 1 ? 1 : 0], Bk, boff); \
        if (PN > 2) gld16s<512>(bf[slot][PN > 2 ? 2 : 0], Bk, boff); \
        if (PN > 3) gld16s<768>(bf[slot][PN > 3 ? 3 : 0], Bk, boff); \
    } while (0)
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int ks = s < n ? s : n - 1;
            const double* Ak = A + (size_t)(4 * ks) * lda;
            const double* Bk = B + (size_t)(4 * ks) * ldb;
            BOSS_LOAD_A(s, Ak);
            BOSS_LOAD_B(s, Bk);
        }
        // Main loop: whole ring passes, branch-free.  Before slot s is consumed exactly
        // (PM+PN)(D-1) younger ring loads are in flight (every pass re-issues every slot, clamped).
        // Row pair p's A fragment is re-loaded right after the MFMAs that read it, so the load issue
        // sits in the shadow of the next pair's MFMAs; the B fragments go last (every MFMA reads them).
        // EXACT — n a multiple of D (short K: 128 … 512 with D = 4, 8, 16; the trailing updates of potrf.hpp): the last D substeps are consumed AS THEY LAND — one pass less in the
        // loop (whose reloads would all be clamped duplicates of the last slice: a quarter of the loads at K = 128, D = 8), then a sweep
        // over the ring with the wait count falling by one slot per substep instead of a full drain in front of the remainder.
        constexpr bool exact = EXACT;
        const int npass = exact ? n / D - 1 : n / D;
        for (int g = 0; g < npass; ++g) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                wait_vmcnt<(PM + PN) * (D - 1)>();
                int ks = (g + 1) * D + s;
                ks = ks < n ? ks : n - 1;
                const double* Ak = A + (size_t)(4 * ks) * lda;
                const double* Bk = B + (size_t)(4 * ks) * ldb;
#pragma unroll
                for (int p = 0; p < PM; ++p) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                        for (int t = 0; t < TN; ++t) {
                            const double b = bf[s][t >> 1][t & 1];
                            acc[2 * p + mm][t] = mfma_f64(SIGN < 0 ? -b : b, af[s][p][mm], acc[2 * p + mm][t]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    if (p == 0) gld16s<0>(af[s][0], Ak, aoff);
                    if (p == 1) gld16s<256>(af[s][PM > 1 ? 1 : 0], Ak, aoff);
                    if (p == 2) gld16s<512>(af[s][PM > 2 ? 2 : 0], Ak, aoff);
                    if (p == 3) gld16s<768>(af[s][PM > 3 ? 3 : 0], Ak, aoff);
                    if (p == PM - 1) BOSS_LOAD_B(s, Bk);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (exact) {
            ring_tail<SIGN, 0>(af, bf, acc);
            return;
        }
        wait_vmcnt<0>();                        // ring drained: the (clamped) reloads of the last pass
        const int rem = n - npass * D;          // < D k-substeps left; their operands are already in the ring
#pragma unroll
        for (int s = 0; s < D; ++s) {
            if (s < rem) {
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const double b = bf[s][t >> 1][t & 1];
                        acc[m][t] = mfma_f64(SIGN < 0 ? -b : b, af[s][m >> 1][m & 1], acc[m][t]);
                    }
            }
        }
    }

    // acc += A(:,0:K)·Bl(0:K,:) with A streamed from global (ring of D) and the B operand resident in
    // LDS as Bl[k*ldbl + c] (ldbl even).  No barriers inside; K may differ per wave (triangular A).
    __device__ static __forceinline__ void run_Blds(const double* __restrict__ A, int lda,
                                                    const double* __restrict__ Bl, int ldbl, int K,
                                                    v4d (&acc)[TM][TN]) {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: uniform control flow around the ring
        const int wr = wave / WC, wc = wave % WC;
        const unsigned aoff = 8u * (unsigned)(wr * (TM * 16) + 2 * (lane & 15) + (lane >> 4) * lda);
        const double* Bs = Bl + wc * (TN * 16) + 2 * (lane & 15) + (lane >> 4) * ldbl;
        const int n = K / 4;
        if (n == 0) return;
        v2d af[D][PM];
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int ks = s < n ? s : n - 1;
            const double* Ak = A + (size_t)(4 * ks) * lda;
            BOSS_LOAD_A(s, Ak);
        }
        const int npass = n / D;
        for (int g = 0; g < npass; ++g) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                v2d bf[PN];
#pragma unroll
                for (int t = 0; t < PN; ++t) bf[t] = *reinterpret_cast<const v2d*>(Bs + (4 * (g * D + s)) * ldbl + t * 32);
                wait_vmcnt<PM * (D - 1)>();
                int ks = (g + 1) * D + s;
                ks = ks < n ? ks : n - 1;
                const double* Ak = A + (size_t)(4 * ks) * lda;
#pragma unroll
                for (int p = 0; p < PM; ++p) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                        for (int t = 0; t < TN; ++t)
                            acc[2 * p + mm][t] = mfma_f64(bf[t >> 1][t & 1], af[s][p][mm], acc[2 * p + mm][t]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (p == 0) gld16s<0>(af[s][0], Ak, aoff);
                    if (p == 1) gld16s<256>(af[s][PM > 1 ? 1 : 0], Ak, aoff);
                    if (p == 2) gld16s<512>(af[s][PM > 2 ? 2 : 0], Ak, aoff);
                    if (p == 3) gld16s<768>(af[s][PM > 3 ? 3 : 0], Ak, aoff);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        wait_vmcnt<0>();
        const int rem = n - npass * D;
#pragma unroll
        for (int s = 0; s < D; ++s) {
            if (s < rem) {
                v2d bf[PN];
#pragma unroll
                for (int t = 0; t < PN; ++t) bf[t] = *reinterpret_cast<const v2d*>(Bs + (4 * (npass * D + s)) * ldbl + t * 32);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int t = 0; t < TN; ++t) acc[m][t] = mfma_f64(bf[t >> 1][t & 1], af[s][m >> 1][m & 1], acc[m][t]);
            }
        }
    }

    // Triangular-balanced variant of run_Blds for PM == 2, WC == 1 and a LOWER-TRIANGULAR A block of
    // BM = 8·32 rows: this wave's two row pairs are the 32-row groups g0 = wave and g1 = 7 − wave, and
    // pair p only multiplies the k < 32 (g_p + 1) columns that are non-zero in its rows.  Every wave
    // then issues 72 four-MFMA stages (contiguous 64-row slices would give wave 3 128 of them).
    // acc rows: tiles 0,1 ↔ group g0, tiles 2,3 ↔ group g1 (see tri_row_of).
    __device__ static __forceinline__ int tri_group(int wave, int p) { return p == 0 ? wave : 7 - wave; }
    __device__ static __forceinline__ int tri_row_of(int wave, int m, int lane) {
        return 32 * tri_group(wave, m >> 1) + 2 * (lane & 15) + (m & 1);
    }
    __device__ static __forceinline__ void run_Blds_tri(const double* __restrict__ A, int lda,
                                                        const double* __restrict__ Bl, int ldbl,
                                                        v4d (&acc)[TM][TN]) {
        static_assert(PM == 2 && WC == 1 && WR == 4 && BM == 256, "balanced pairing is written for 4 waves × 2 row pairs");
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int g0 = wave, g1 = 7 - wave;
        const unsigned off0 = 8u * (unsigned)(32 * g0 + 2 * (lane & 15) + (lane >> 4) * lda);
        const unsigned off1 = 8u * (unsigned)(32 * g1 + 2 * (lane & 15) + (lane >> 4) * lda);
        const double* Bs = Bl + 2 * (lane & 15) + (lane >> 4) * ldbl;
        const int n0 = 8 * (g0 + 1), n1 = 8 * (g1 + 1);          // k-substeps (of 4 columns) per pair
        {   // ---- phase 1: k-substeps [0, n0), both pairs
            v2d af[D][2];
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const int ks = s < n0 ? s : n0 - 1;
                const double* Ak = A + (size_t)(4 * ks) * lda;
                gld16s<0>(af[s][0], Ak, off0);
                gld16s<0>(af[s][1], Ak, off1);
            }
            const int npass = n0 / D;
            for (int g = 0; g < npass; ++g) {
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    v2d bf[PN];
#pragma unroll
                    for (int t = 0; t < PN; ++t) bf[t] = *reinterpret_cast<const v2d*>(Bs + (4 * (g * D + s)) * ldbl + t * 32);
                    wait_vmcnt<2 * (D - 1)>();
                    int ks = (g + 1) * D + s;
                    ks = ks < n0 ? ks : n0 - 1;
                    const double* Ak = A + (size_t)(4 * ks) * lda;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                            for (int t = 0; t < TN; ++t)
                                acc[2 * p + mm][t] = mfma_f64(bf[t >> 1][t & 1], af[s][p][mm], acc[2 * p + mm][t]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (p == 0) gld16s<0>(af[s][0], Ak, off0);
                        else gld16s<0>(af[s][1], Ak, off1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            wait_vmcnt<0>();
            const int rem = n0 - npass * D;
#pragma unroll
            for (int s = 0; s < D; ++s) {
                if (s < rem) {
                    v2d bf[PN];
#pragma unroll
                    for (int t = 0; t < PN; ++t) bf[t] = *reinterpret_cast<const v2d*>(Bs + (4 * (npass * D + s)) * ldbl + t * 32);
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int t = 0; t < TN; ++t) acc[m][t] = mfma_f64(bf[t >> 1][t & 1], af[s][m >> 1][m & 1], acc[m][t]);
                }
            }
        }
        {   // ---- phase 2: k-substeps [n0, n1), pair 1 only (rows of group g1 reach further right)
            const int n2 = n1 - n0;
            if (n2 <= 0) return;
            const double* A2 = A + (size_t)(4 * n0) * lda;
            const double* Bs2 = Bs + (size_t)(4 * n0) * ldbl;
            v2d af[D];
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const int ks = s < n2 ? s : n2 - 1;
                gld16s<0>(af[s], A2 + (size_t)(4 * ks) * lda, off1);
            }
            const int npass = n2 / D;
            for (int g = 0; g < npass; ++g) {
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    v2d bf[PN];
#pragma unroll
                    for (int t = 0; t < PN; ++t) bf[t] = *reinterpret_cast<const v2d*>(Bs2 + (4 * (g * D + s)) * ldbl + t * 32);
                    wait_vmcnt<D - 1>();
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                        for (int t = 0; t < TN; ++t) acc[2 + mm][t] = mfma_f64(bf[t >> 1][t & 1], af[s][mm], acc[2 + mm][t]);
                    __builtin_amdgcn_sched_barrier(0);
                    int ks = (g + 1) * D + s;
                    ks = ks < n2 ? ks : n2 - 1;
                    gld16s<0>(af[s], A2 + (size_t)(4 * ks) * lda, off1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            wait_vmcnt<0>();
            const int rem = n2 - npass * D;
#pragma unroll
            for (int s = 0; s < D; ++s) {
                if (s < rem) {
                    v2d bf[PN];
#pragma unroll
                    for (int t = 0; t < PN; ++t) bf[t] = *reinterpret_cast<const v2d*>(Bs2 + (4 * (npass * D + s)) * ldbl + t * 32);
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                        for (int t = 0; t < TN; ++t) acc[2 + mm][t] = mfma_f64(bf[t >> 1][t & 1], af[s][mm], acc[2 + mm][t]);
                }
            }
        }
    }
};

}  // namespace boss

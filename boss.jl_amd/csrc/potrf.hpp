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

#ifdef BOSS_CHAIN_TRACE
__device__ unsigned long long g_cutrace[64 * 4];            // column-update kernel: earliest entry / latest exit per step
__device__ unsigned long long g_ptrace[64 * 32];            // inside the published diagonal block: per panel, wave 0 / wave 15 stamps
#define PTRACE(col0, jb, slot) do { if ((threadIdx.x & 63) == 0) g_ptrace[(((col0) / BLK) & 63) * 32 + (slot) * 8 + (jb)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PTRACE(col0, jb, slot) do { } while (0)
#endif

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


// 16×16 tile in the "transposed C" layout: register i of lane l = element (row = l&15, col = (l>>4)+4i),
// so register t IS the 16×4 micro-panel of columns 4t..4t+3 in MFMA A/B-operand layout.
//
// chol16_unscaled: the sequential pivot chain of one diagonal tile, WITHOUT square roots.  It runs on ONE wave, and a
// lone wave pays ≈12 cycles per fp64 VALU instruction and ≈5 per 32-bit one whether or not they depend on each other —
// so the code minimises the instruction COUNT on this wave and leaves everything that can wait to other waves:
//   * the tile is factored in the UNSCALED form S' = L16·diag(√p) (column c of L16 times its pivot's root): eliminating
//     column j from the later columns needs only 1/p_j (v_rcp_f64 + one cubic correction: 4 instructions; 1/√p cost 6
//     plus a squaring), and the diagonal of S' holds the pivots themselves, S'(c,c) = p_c;
//   * the identity tile E rides along through the same column operations and ends as E' = L16^{-T}·diag(√p);
//   * the column's entries reach the lanes that need them by three ds_bpermute (issued before the reciprocal's
//     refinement, which hides their latency), the update is one unconditional fma per tile with a pre-masked factor;
//   * after each 4-column micro-panel ONE rank-4 MFMA per tile updates the remaining columns, its A operand scaled by
//     1/p of its column (ipsel) instead of both operands by 1/√p.
// The 1/√p scaling of L16, of L16^{-1} and of the row tiles below is applied by the waves that consume S' and E'
// (they derive it from the diagonal of S'), so no square root is ever computed on the chain.
// ipsel[t] of lane l returns 1/p of column 4t + (l>>4).
__device__ __forceinline__ void chol16_unscaled(v4d& S, v4d& E, double (&ipsel)[4], int lane, int col0, int& fail) {
    const int r16 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) E[i] = (r16 == q + 4 * i) ? 1.0 : 0.0;
    bool bad = false;
    int badcol = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ipsel[t] = 1.0;
        // The chain is LATENCY-bound (≈9 dependent instructions per column at ≈23 cycles each when every pivot is read back
        // from the updated tile): the next pivot is therefore advanced as a wave-uniform scalar recurrence
        //     p_{j+1} = S(j+1,j+1) − S(j+1,j)·(S(j+1,j)/p_j)
        // from two entries read BEFORE column j is eliminated — the same two operations the owning lane performs, so it
        // equals the tile's diagonal entry bit for bit — and the reciprocal of p_{j+1} starts while the vector update of
        // column j is still in flight.  Dependent path per column: mul, fma, rcp, 3 × fma.
        double p = readlane_f64(S[t], 4 * t);                       // S(4t, 4t): lane (r16 = 4t, q = 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double dn = 0.0, ln = 0.0, lS = 0.0, lE = 0.0, lc = 0.0;
            if (j < 3) {
                dn = readlane_f64(S[t], 16 * (j + 1) + 4 * t + j + 1);      // S(c+1, c+1), c = 4t+j, eliminations < j applied
                ln = readlane_f64(S[t], 16 * j + 4 * t + j + 1);            // S(c+1, c)
                // X(r,c') -= X(r,c)·S(c',c) / p  for the columns c' > c of this micro-panel
                lS = __shfl(S[t], 16 * j + r16);            // S(r, c): same row, column j of the micro-panel
                lE = __shfl(E[t], 16 * j + r16);
                lc = __shfl(S[t], 16 * j + 4 * t + q);      // S(c', c) for this lane's own column c' = 4t+q
            }
            if (!(p > 0.0) && !bad) {              // NaN or non-positive pivot: not PD (reported; the numbers that follow are garbage)
                bad = true;
                badcol = 4 * t + j;
            }
            const double ip = rcp_refined(p);
            ipsel[t] = (q == j) ? ip : ipsel[t];
            if (j < 3) {
                p = __builtin_fma(-ln, ln * ip, dn);                // next pivot, ahead of the tile update
                const double lcm = (q > j) ? lc : 0.0;
                const double m = lcm * ip;
                S[t] = __builtin_fma(-lS, m, S[t]);
                E[t] = __builtin_fma(-lE, m, E[t]);
            }
        }
        if (t < 3) {
            // rank-4 update of columns > 4t+3:  X(r,c) -= Σ_k X(r,4t+k) · S(c,4t+k) / p_{4t+k}
            const double am = (r16 > 4 * t + 3) ? -S[t] * ipsel[t] : 0.0;
            const double bS = S[t], bE = E[t];
            S = mfma_f64(am, bS, S);
            E = mfma_f64(am, bE, E);
        }
    }
    if (bad && fail < 0) fail = col0 + badcol;
}

// Lane-private staging of a 16×16 tile held in the layout above: lane (r16, q) owns four consecutive doubles
// (its columns q, q+4, q+8, q+12) at ((r16·4 + q)·4): two 16-byte LDS writes per tile on the chain wave instead of
// four 8-byte ones, and element (row R, col C) is read back from ((R·4 + (C&3))·4 + (C>>2)) — conflict-free for the
// MFMA operand pattern (row = 4s + (l>>4), col = l&15) of the consumers.
__device__ __forceinline__ int lp_index(int R, int C) { return ((R * 4 + (C & 3)) * 4) + (C >> 2); }

// ------------------------------------------------------------------------------------------
// Hand-offs between workgroups that run at the same time (the persistent panel chain and its followers, below).
// Form used (MI355X_MICROARCH.md, "Valid forms"): EVERY handed-off byte is stored with an agent-scope (sc1, write-through)
// store, the storing wave drains its stores (s_waitcnt vmcnt(0)) and then one lane stores / raises the sequence word (sc1);
// the consumer polls the word with sc1 loads and reads the bytes with sc1 loads only after its poll has matched (other waves
// of its workgroup: after a barrier the polling wave joins).  Sequence words only ever grow (one base per factorisation).
// ------------------------------------------------------------------------------------------
// (explicit global address space: on a generic pointer these builtins become flat_ instructions, which count on lgkmcnt as
// well — a publishing wave would then stall at the next LDS-only barrier until its stores have landed — and flat sc1 loads are
// not among the measured-valid hand-off forms)
typedef __attribute__((address_space(3))) int lds_int_t;
typedef __attribute__((address_space(1))) unsigned long long gmem_u64;
typedef __attribute__((address_space(1))) int gmem_i32;
__device__ __forceinline__ gmem_u64* as_global(const void* p) { return (gmem_u64*)(unsigned long long)p; }
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(as_global(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(as_global(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned long long ld_word(const unsigned long long* w) {
    return __hip_atomic_load(as_global(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_info(const int* p) {
    return __hip_atomic_load((gmem_i32*)(unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_info(int* p, int v) {
    __hip_atomic_store((gmem_i32*)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The flag as memory holds it NOW, whatever this XCD's L2 has cached: an agent-scope read-modify-write executes at the memory side.
// An sc1 LOAD is served by the XCD's L2, and the flag's line is not a hand-off line — the update's first kernel clears it with a
// plain store, the log-det kernel writes its neighbours — so an L2 may keep a copy from before another XCD stored INT_MIN there.
// A resident workgroup that polled the flag with loads then never saw the mark and sat out the full budget of EVERY later wait
// (63 steps × 8 panels × the budget: the fallback of an N = 8192 update took minutes).  Used every 64th poll only.
// (inline asm: LLVM folds an idempotent `atomicrmw or p, 0` back into the very load this replaces)
__device__ __forceinline__ int ld_info_fresh(const int* p) {
    int r;
    const int zero = 0;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"((gmem_i32*)(unsigned long long)p), "v"(zero) : "memory");
    return r;
}
// 16 bytes per lane, write-through (there is no 16-byte atomic builtin).  The s_nop covers the wait state a VALU write to the
// data registers needs behind a wide store: the compiler's hazard recogniser does not look inside inline asm.
__device__ __forceinline__ void st_sc1_x4(double* p, v2d v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(as_global(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void raise_word(unsigned long long* w, unsigned long long v) {
    __hip_atomic_store(as_global(w), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// word indices inside the context's signal block (each group on a 128-byte line of its own)
constexpr int SIGW_GATE = 0, SIGW_BULK = 3, SIGW_PANEL = 16, SIGW_WDONE = 32, SIGW_CRIT = 48, SIGW_NEAR = 8, SIGW_PROG = 64, SIGW_PROG_STRIDE = 16,
              SIGW_UP = SIGW_PROG + 8 * SIGW_PROG_STRIDE,      // (the eight strips' progress words: a line each — eight CUs store them, one wave polls all)
                                                               // SIGW_UP: workgroups of the resident kernels that have started (chain_ready_kernel)
              SIGW_DBG = SIGW_UP + 16,                    // BOSS_DEBUG_WATCH: [0] first give-up code, [1] give-ups, [2] last code, [4] waits ended by the mark, [5] by the clock
              SIG_WORDS = SIGW_DBG + 16;
// who gave up first (diagnostics: BOSS_CHAIN_VERBOSE prints it when an update falls back): code * 1000 + detail
__device__ int g_giveup = 0;
#ifdef BOSS_DEBUG_WATCH_BUILD
__device__ unsigned long long* g_dbg = nullptr;             // diagnostics build (tools/): the host-mapped signal block, for the watcher thread
#endif
__device__ __forceinline__ void note_giveup(int code, int detail) {
    atomicCAS(&g_giveup, 0, code * 1000 + (detail & 511));
#ifdef BOSS_DEBUG_WATCH_BUILD
    unsigned long long* d = g_dbg;
    if (d) {
        atomicCAS(d + SIGW_DBG, 0ull, (unsigned long long)(code * 1000 + (detail & 511)));
        atomicAdd(d + SIGW_DBG + 1, 1ull);
        d[SIGW_DBG + 2] = (unsigned long long)(code * 1000 + (detail & 511));
    }
#endif
}
// Bounded waits.  Every wait of the cross-kernel protocols gives up when `budget` ticks of the constant 100 MHz clock (s_memrealtime)
// have passed since the wait began: the host derives the budget from the size of the system (≈ 20× the expected update time, at
// least 20 ms: poll_budget_ticks, host_factor.inc), so that a caller whose device cannot run the kernels side by side (another
// process holding the CUs, a tool that serialises the queues) loses tens of milliseconds ONCE, not a second per wait.  The clock is
// read every 64th poll; POLL_CAP bounds the loop count whatever the clock does.  A wait that gives up marks the factorisation
// (info = INT_MIN: gp_finish repeats it on the next simpler schedule) and every other waiter notices the mark within 64 polls.
constexpr int POLL_CAP = 1 << 26;
constexpr int POLL_CAP_LDS = 1 << 22;                      // spins on an LDS word a sibling wave of the same workgroup posts (that wave's own waits are clocked)
struct PollTimer {
    unsigned long long t0;
    unsigned budget;
    __device__ __forceinline__ explicit PollTimer(unsigned b) : t0(__builtin_amdgcn_s_memrealtime()), budget(b) {}
    __device__ __forceinline__ bool expired() const { return __builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)budget; }
    // every 64th poll: 1 = out of time, 2 = somebody else gave up already, 0 = keep polling
    __device__ __forceinline__ int check(int i, const int* info) const {
        if ((i & 63) != 63) return 0;
        if (ld_info(info) == INT_MIN) return 2;
        // (the memory-side read is an atomic on ONE word: every 64th poll from the hundreds of panel-solve workgroups of a step it
        // serialised behind itself and cost the N = 4096 update 0.1 ms — every 1024th poll only, i.e. in waits of a millisecond and more)
        if ((i & 1023) == 1023) {
            const int v = ld_info_fresh(info);
#ifdef BOSS_DEBUG_WATCH_BUILD
            if (v == INT_MIN && g_dbg && (threadIdx.x & 63) == 0) atomicAdd(g_dbg + SIGW_DBG + 4, 1ull);
#endif
            if (v == INT_MIN) return 2;
        }
        const bool out = expired();
#ifdef BOSS_DEBUG_WATCH_BUILD
        if (out && g_dbg && (threadIdx.x & 63) == 0) atomicAdd(g_dbg + SIGW_DBG + 5, 1ull);
#endif
        return out ? 1 : 0;
    }
};
// One wave waits until *w >= v (wave-uniform).  false: gave up (timeout, or another waiter already marked the factorisation).
__device__ __forceinline__ bool poll_ge(const unsigned long long* w, unsigned long long v, int* info, unsigned budget) {
    const PollTimer tm(budget);
    for (int i = 0; i < POLL_CAP; ++i) {
        if (ld_word(w) >= v) {
            asm volatile("" ::: "memory");
            return true;
        }
        const int st = tm.check(i, info);
        if (st == 2) return false;
        if (st == 1) break;
        __builtin_amdgcn_s_sleep(1);
    }
    if ((threadIdx.x & 63) == 0) {
        st_info(info, INT_MIN);
        note_giveup(1, (int)(v & 511));
    }
    return false;
}

// ------------------------------------------------------------------------------------------
// Diagonal block (128×128, LDS-resident), 16-column panels, look-ahead inside the block:
//   phase A  wave 0 alone runs the sequential pivot chain on the diagonal 16×16 tile (chol16_unscaled) — meanwhile the
//            update waves are still applying the PREVIOUS panel's rank-16 update to the rest of the block;
//   phase B  waves 1..7-jb turn the row tiles into P = X · inv(L16)^T (4 MFMAs with the unscaled E', then the 1/√p column
//            scaling); wave 0 forms the first row tile unscaled, applies it to the NEXT diagonal tile (weights 1/p) and
//            goes straight back to phase A; wave 15 scales S' and E' into L16 and inv(L16) and writes both to global.
// The block lives in LDS as its 36 lower 16×16 tiles, tile-major (tile (i, j), i >= j, at (i(i+1)/2 + j)·256, column-major
// inside): 74 KB instead of the 147 KB of a padded square, and only the lower tiles travel between global memory and LDS.
// (Measured and rejected: 8 waves so that the kernel fits beside one bulk-update workgroup on every CU — co-residency with a
// bulk wave on the pivot chain's SIMD stretches the kernel from 22 to 60 µs; it is better off waiting for an idle CU.)
// inv16 out: for each of the 8 diagonal 16×16 blocks its inverse X (column-major 16×16,
// X(r,c) at c*16+r, zero above the diagonal).
// info: first failing global column + 1 (0 = success) — PosDefException analogue.
// ------------------------------------------------------------------------------------------
constexpr int DIAG_THREADS = 1024;   // 16 waves = 4 per SIMD: single-wave fp64 VALU / LDS / MFMA issue rates are 3-5x below the multi-wave rates
constexpr int DIAG_TILES = 36;
constexpr int DIAG_STAGE = 2 * 256 + 2;                // doubles: unscaled E' and S' of the current panel (lane-private layout) + a flag word
constexpr int DIAG_LDS_BYTES = (DIAG_TILES * 256 + DIAG_STAGE) * 8;
__device__ __forceinline__ int dtile(int i, int j) { return (i * (i + 1) / 2 + j) * 256; }   // LDS offset of lower tile (i, j)

// nsub: number of leading 16-column panels that hold observations (8 = the whole block); the panels behind them are
// identity padding and are neither factored nor touched.
// KEEP: the scaled diagonal tiles and their inverses additionally stay in LDS (tile slot (jb, jb), and Is + jb·256 in the
// inv16 format) for a caller that goes on to solve with the factor inside the same kernel (small_fit_kernel).
// TILE0_GLOBAL: the pivot-chain wave takes diagonal tile 0 straight from global memory (the caller did not wait for the
// workgroup's LDS fill: the chain starts while the other waves are still loading; barrier 1 of the first panel orders the fill).
// PUB: the block is factored by the persistent chain (potrf_chain_kernel) while followers on other CUs consume it panel by
// panel: everything that leaves for global memory goes out with sc1 stores from wave 15 (idle in phase B, not on the pivot
// chain's SIMD), which then drains its stores and raises pubword to pubseq0 + jb + 1 = "panel jb of this block is out: the
// inverse of its diagonal tile and its row tiles" (the last panel has no row tiles: its word follows barrier 1 directly).  A failed
// pivot is written to info by the pivot-chain wave itself before that panel's barrier (nobody is stream-ordered behind this kernel).
template <bool KEEP = false, bool TILE0_GLOBAL = false, bool PUB = false>
__device__ __forceinline__ void diag_block_factor(double* __restrict__ smem, double* __restrict__ A, int ld,
                                                  double* __restrict__ inv16, int col0, int nsub, int& fail,
                                                  double* __restrict__ Is = nullptr, unsigned long long* pubword = nullptr,
                                                  unsigned long long pubseq0 = 0, int* info = nullptr, int tidtok = 0) {
    double* D = smem;
    double* Es = smem + DIAG_TILES * 256;            // E' = L16^{-T}·diag(√p) of the current panel, lane-private layout
    double* Ss = Es + 256;                           // S' = L16·diag(√p), lane-private layout
    // panel whose first row tile wave 0 has finished reading (explicit LDS address space: a volatile access through a generic
    // pointer is a flat_ instruction, and the wave that spins on it then waits on vmcnt as well)
    volatile lds_int_t* xread = (volatile lds_int_t*)(unsigned)(unsigned long long)(Ss + 256);
    // (tidtok: an opaque zero the persistent chain kernel passes so that the per-lane offsets below are recomputed per block instead
    // of being hoisted out of its block loop, where they would stay live — and spill — through the follower phase)
    // tidtok >= 0x10000: the caller passes the thread index itself (+ 0x10000), rebuilt from scalar state — the chain kernel cuts
    // every vector live range between its phases and would otherwise fetch the index from scratch at the start of each block
    const int tid = tidtok >= 0x10000 ? tidtok - 0x10000 : (int)threadIdx.x + tidtok, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: tile decode runs on the SALU
    const int r16 = lane & 15, q = lane >> 4;
    constexpr int NW = DIAG_THREADS / 64;
    if (tid == 0) *xread = 0;
    v4d S = {0.0, 0.0, 0.0, 0.0};
    // loop-invariant per-lane offsets inside a 16×16 tile (the pivot-chain wave pays ≈5 cycles per integer instruction
    // too): symmetric-fill source of a diagonal tile, plain element (row r16, col q + 4i)
    int osym[4], oel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = q + 4 * i;
        osym[i] = (r16 >= c) ? (c * 16 + r16) : (r16 * 16 + c);
        oel[i] = c * 16 + r16;
    }
    // consumers' read offsets into the staged tiles: E'(row = 4s + q, col = r16) and the pivots S'(c, c), c = q + 4i
    int oe[4], od[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oe[i] = lp_index(4 * i + q, r16);
        od[i] = lp_index(q + 4 * i, q + 4 * i);
    }
    const int olp = (r16 * 4 + q) * 4;               // this lane's own four doubles of a staged tile
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                    // symmetric fill of diagonal tile 0 from its lower part
            if constexpr (TILE0_GLOBAL) {
                const int c = q + 4 * i;
                S[i] = (r16 >= c) ? A[(size_t)c * ld + r16] : A[(size_t)r16 * ld + c];
            } else {
                S[i] = D[osym[i]];
            }
        }
    }
    for (int jb = 0; jb < nsub; ++jb) {
        const int t = nsub - 1 - jb;                  // row tiles below the diagonal tile
        double ipsel[4];
        // ---------------- phase A ----------------
        if (wave == 0) {
            v4d E;
            if constexpr (PUB) PTRACE(col0, jb, 0);
            chol16_unscaled(S, E, ipsel, lane, col0 + jb * 16, fail);
            if constexpr (PUB) PTRACE(col0, jb, 1);
            *reinterpret_cast<v4d*>(Es + olp) = E;
            *reinterpret_cast<v4d*>(Ss + olp) = S;
            if constexpr (PUB) {
                if (fail >= 0) {                              // (rare) reported before this panel is published
                    if (lane == 0 && ld_info(info) == 0) st_info(info, fail + 1);
                    drain_stores();
                }
            }
        } else if (jb > 0 && (wave & 3) != 0 && !(PUB && wave == 15)) {
            // rest of the rank-16 update with panel jb-1 (tile (jb,jb) was done by wave 0 in phase B).
            // Waves 4, 8, 12 share wave 0's SIMD: they stay idle here, so the pivot chain's MFMAs and
            // VALU ops never queue behind update MFMAs (measured: 5450 → 3980 cycles per 16-column chain).
            const int jp = jb - 1, tp = nsub - 1 - jp;
            const int T = tp * (tp + 1) / 2;
            constexpr int U = 2;                                // (U = 3 would cover panel 0's 27 tiles in one round, but the extra registers slow every round: measured)
            constexpr int NUPD = NW - NW / 4 - (PUB ? 1 : 0);   // 12 update waves (PUB: 11 — wave 15 only publishes)
            const int uw = wave - 1 - (wave >> 2);              // 0..11
            for (int q0 = 1 + uw; q0 < T; q0 += U * NUPD) {     // update waves cover tile indices 1..T-1
                int ti[U], tj[U];
                bool ok[U];
                v4d cr[U];
                double* ct[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int qq = q0 + NUPD * u;
                    ok[u] = qq < T;
                    int a = 0;
                    while ((a + 1) * (a + 2) / 2 <= qq) ++a;
                    const int b = qq - a * (a + 1) / 2;
                    ti[u] = ok[u] ? jp + 1 + a : jp + 2;
                    tj[u] = ok[u] ? jp + 1 + b : jp + 2;
                    ct[u] = D + dtile(ti[u], tj[u]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) cr[u][i] = ct[u][oel[i]];
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    double af[U], bf[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        af[u] = D[dtile(tj[u], jp) + oel[s4]];       // P(tj rows, k = q + 4s)
                        bf[u] = D[dtile(ti[u], jp) + oel[s4]];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) cr[u] = mfma_f64(-af[u], bf[u], cr[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (ok[u]) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) ct[u][oel[i]] = cr[u][i];
                    }
                }
            }
        }
        lds_barrier();                                // barrier 1: S'/E' published, block fully updated through panel jb-1
        if (wave == 15) {
            // (a wave that has no row tile in phase B and does not share the pivot chain's SIMD) the scaled results of this panel's diagonal tile go to global from here —
            // L16 = S'·diag(1/√p) into the factor, inv(L16) = (E'·diag(1/√p))^T into inv16 (for the panel solves)
            const v4d Sv = *reinterpret_cast<const v4d*>(Ss + olp), Ev = *reinterpret_cast<const v4d*>(Es + olp);
            double* At = A + (size_t)(jb * 16) * ((size_t)ld + 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double rs = rsqrt_refined(Ss[od[i]]);          // 1/√p of column c = q + 4i
                const int c = q + 4 * i;
                if constexpr (PUB) {
                    if (r16 >= c) st_sc1(At + (size_t)c * ld + r16, Sv[i] * rs);
                    st_sc1(inv16 + jb * 256 + r16 * 16 + c, Ev[i] * rs);
                } else {
                    if (r16 >= c) At[(size_t)c * ld + r16] = Sv[i] * rs;
                    inv16[jb * 256 + r16 * 16 + c] = Ev[i] * rs;        // inv(L16)(c, r16) = E(r16, c), stored at r16*16 + c
                }
                if constexpr (KEEP) {
                    D[dtile(jb, jb) + oel[i]] = (r16 >= c) ? Sv[i] * rs : 0.0;
                    Is[jb * 256 + r16 * 16 + c] = Ev[i] * rs;
                }
            }
            if constexpr (PUB) {
                if (t == 0) {                                 // last panel: nothing else to publish
                    drain_stores();
                    if (lane == 0) raise_word(pubword, pubseq0 + jb + 1);
                }
            }
        }
        if (t == 0) break;
        // ---------------- phase B: row tiles  P = X · inv(L16)^T ----------------
        if (wave == 0) {
            // The pivot-chain wave forms the first row tile itself, UNSCALED (P' = X·E'^T = P·diag(√p)), and applies it to the
            // NEXT diagonal tile right away with the weights 1/p (P' diag(1/p) P'^T = P P^T), so the next 16-column chain starts
            // right behind barrier 2.  Register s of the MFMA result is P'(r, k = q + 4s): a valid k-slice for both operands.
            // (The scaled tile that the rest of the block needs is written by wave 1.)
            const int tn = jb + 1;
            v4d P = {0.0, 0.0, 0.0, 0.0};
            const double* Dx = D + dtile(tn, jb);
            double aop[4], bop[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                aop[s4] = Es[oe[s4]];                                // E'(k = 4s+q, c = r16)
                bop[s4] = Dx[oel[s4]];                               // X(r16, k = q+4s)
            }
            const double* Dn = D + dtile(tn, tn);
#pragma unroll
            for (int i = 0; i < 4; ++i) S[i] = Dn[osym[i]];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) P = mfma_f64(aop[s4], bop[s4], P);      // (waits for the reads above)
            if (lane == 0) *xread = jb + 1;            // wave 1 may now overwrite X(jb+1, jb) with the scaled tile
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) S = mfma_f64(-P[s4] * ipsel[s4], P[s4], S);
        } else if ((wave & 3) != 0 && wave - (wave >> 2) <= t) {
            // row tiles jb+1..nsub-1 on the waves 1,2,3,5,6,7,9 — none on the pivot chain's SIMD (waves 4, 8, 12 stay idle: wave 0's
            // own MFMAs and LDS traffic of this phase are the block's critical path)
            const int wi = wave - (wave >> 2);         // 1..7
            const int tr = jb + wi;
            v4d P = {0.0, 0.0, 0.0, 0.0};
            double rs[4];
            double* Xt = D + dtile(tr, jb);
#pragma unroll
            for (int i = 0; i < 4; ++i) rs[i] = Ss[od[i]];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double aop = Es[oe[s4]];                       // E'(k = 4s+q, c = r16)
                const double bop = Xt[oel[s4]];                      // X(r16, k = 4s+q)
                P = mfma_f64(aop, bop, P);
            }
            double pv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[i] = P[i] * rsqrt_refined(rs[i]);
            if (wi == 1) {
                // this tile is also the operand wave 0 is reading for the next diagonal tile: do not overwrite it before
                // wave 0 has its copy (wave 0 raises the flag right after its first MFMAs; it never waits for this wave)
                while (*xread < jb + 1) __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) Xt[oel[i]] = pv[i];
        }
        lds_barrier();                                // barrier 2: panel jb final
        if constexpr (PUB) {
            if (wave == 15) {
                for (int tr = jb + 1; tr < nsub; ++tr) {
                    const double* src = D + dtile(tr, jb);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int e = lane + 64 * h, c = e >> 3, rp = e & 7;
                        st_sc1_x4(A + (size_t)(jb * 16 + c) * ld + tr * 16 + 2 * rp, *reinterpret_cast<const v2d*>(src + c * 16 + 2 * rp));
                    }
                }
                // panel jb is out: its inverse / diagonal tile (stored after barrier 1) and its row tiles.  (This wave takes no part
                // in the next phase A's update, so the wait costs the block nothing.)
                PTRACE(col0, jb, 2);
                drain_stores();
                PTRACE(col0, jb, 3);
                if (lane == 0) raise_word(pubword, pubseq0 + jb + 1);
            }
        }
        if constexpr (TILE0_GLOBAL) {
            // the finished row tiles of this panel leave for global memory now, from a wave that has nothing else to do (12: idle
            // in both phases), instead of in a copy loop behind the last panel
            if (wave == 12) {
                for (int tr = jb + 1; tr < nsub; ++tr) {
                    const double* src = D + dtile(tr, jb);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int e = lane + 64 * h, c = e >> 3, rp = e & 7;
                        *reinterpret_cast<v2d*>(A + (size_t)(jb * 16 + c) * ld + tr * 16 + 2 * rp) =
                            *reinterpret_cast<const v2d*>(src + c * 16 + 2 * rp);
                    }
                }
            }
        }
    }
}

// Gate in front of a kernel on another stream: one lane polls a sequence word until it reaches v (the producer stream's
// next kernel stores it at its entry, see potrf_diag_kernel), sleeping ≈0.4 µs between polls so that it does not disturb the
// waves it shares a SIMD with.  The spin is bounded (PollTimer); a timeout marks the factorisation as failed (info = INT_MIN)
// instead of letting the gated kernel run on unfinished operands.
__global__ __launch_bounds__(64) void potrf_gate_kernel(unsigned long long* __restrict__ sig, unsigned long long v,
                                                        int* __restrict__ info, unsigned budget) {
    if (threadIdx.x != 0) return;
    const PollTimer tm(budget);
    for (int i = 0; i < POLL_CAP; ++i) {
        if (__hip_atomic_load(sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v) return;
        if ((i & 63) == 63 && tm.expired()) break;
        __builtin_amdgcn_s_sleep(16);
    }
    info[0] = INT_MIN;
    note_giveup(3, (int)(v & 511));
}

// The opposite direction (side stream -> chain): a one-lane kernel enqueued right behind the bulk update stores the next sequence
// number of ITS word when it starts (= the bulk update has completed and released its stores), and the chain kernel that needs
// the updated columns polls that word at its entry — an event took 11-13 µs to get from one stream to the other, which the chain
// started to wait for once its own kernels had become faster than the first bulk updates.
__global__ __launch_bounds__(64) void potrf_publish_kernel(unsigned long long* __restrict__ word, unsigned long long v) {
    if (threadIdx.x == 0) __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Every thread of the waiting kernel calls this before it touches the operands (w == nullptr: nothing to wait for).
// Normally the word is already there: one load.  ONE thread per workgroup polls, sleeping ≈0.4 µs between polls — with the
// resident chain the column updates arrive early and wait tens of microseconds, and a thousand waves polling one line in memory
// slowed the bulk update they were waiting for; the others wait at the barrier.  A workgroup that had to wait re-acquires (its
// kernel started before the producer ended).
__device__ __forceinline__ void wait_word(unsigned long long* w, unsigned long long v, int* info, unsigned budget) {
    if (!w) return;
    __shared__ int waited_s;
    if (threadIdx.x == 0) {
        bool ok = false, waited = false;
        const PollTimer tm(budget);
        for (int i = 0; i < POLL_CAP; ++i) {
            if (ld_word(w) >= v) {
                ok = true;
                break;
            }
            waited = true;
            if (tm.check(i, info)) break;
            __builtin_amdgcn_s_sleep(16);
        }
        if (!ok) {
            st_info(info, INT_MIN);
            note_giveup(2, (int)(v & 511));
        }
        *(volatile lds_int_t*)&waited_s = waited ? 1 : 0;
    }
    __syncthreads();
    if (*(volatile lds_int_t*)&waited_s) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// sig / sigval: when sig is non-null the kernel stores sigval there at its entry — every earlier kernel of its stream has
// finished by then, so a gate kernel on another stream can release work that depends on them (the look-ahead
// schedule's bulk update) without an event on this stream.
__global__ __launch_bounds__(DIAG_THREADS) void potrf_diag_kernel(double* __restrict__ Abase, int ld, size_t bstride,
                                                                  int k, double* __restrict__ inv16base,
                                                                  size_t inv16_bstride, int* __restrict__ info,
                                                                  unsigned long long* sig, unsigned long long sigval) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    if (sig && tid == 0 && blockIdx.z == 0) __hip_atomic_store(sig, sigval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double* A = Abase + (size_t)blockIdx.z * bstride + (size_t)k * BLK * ((size_t)ld + 1);
    double* inv16 = inv16base + (size_t)blockIdx.z * inv16_bstride + (size_t)k * (8 * 256);
    // lower tiles global -> LDS, a wave per tile (36 tiles over 16 waves), 16-byte accesses: lane l moves the pairs l and l + 64
    // (wave 0 = the pivot chain takes tile (0,0) from global memory itself and starts at once; the other 15 waves fill the
    // LDS with tiles 1..35, ordered before their first use by the first barrier inside diag_block_factor)
    const int lane = tid & 63, wv = tid >> 6;
    if (wv != 0) {
        for (int tl = wv; tl < DIAG_TILES; tl += DIAG_THREADS / 64 - 1) {
            int ti = 0;
            while ((ti + 1) * (ti + 2) / 2 <= tl) ++ti;
            const int tj = tl - ti * (ti + 1) / 2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = lane + 64 * h, c = e >> 3, rp = e & 7;
                *reinterpret_cast<v2d*>(smem + tl * 256 + c * 16 + 2 * rp) =
                    *reinterpret_cast<const v2d*>(A + (size_t)(tj * 16 + c) * ld + ti * 16 + 2 * rp);
            }
        }
    }
    int fail = -1;
    diag_block_factor<false, true>(smem, A, ld, inv16, k * BLK, 8, fail);
    if (tid == 0 && fail >= 0) {
        if (info[blockIdx.z] == 0) info[blockIdx.z] = fail + 1;
    }
    // (the tiles below the diagonal went out panel by panel from wave 12, the diagonal tiles from wave 15)
}

// ------------------------------------------------------------------------------------------
// Whole posterior update in ONE launch for N <= 128 observations (BASELINE config 1: the reference's own example runs
// with 3..20 points, examples/example.jl:117-162): scaled points, Gram tiles straight into LDS, the diagonal-block
// factorisation above restricted to the ceil(N/16) panels that hold observations, z = L \ (y - m) by tile-wise
// substitution, logdet and z^T z — and the results written into mapped host memory, so an update costs one kernel launch
// and one stream synchronisation (the general path: 8 launches, two copies).  Replaces posterior_gp / logpdf
// (src/models/gaussian_process.jl:199-211,269-280) for small data sets; outputs are the same arrays every other entry
// point reads (factor, z row, inv16, scaled points, resident hyper-parameters).
// ------------------------------------------------------------------------------------------
constexpr int SMALL_MAX_N = BLK, SMALL_MAX_D = 32;
struct SmallFitPar {
    int d, N, Np, ld, kern;
    double amp2, sig2;
    double invlam[SMALL_MAX_D];
};
constexpr int SMALL_LDS_DOUBLES = DIAG_TILES * 256 + DIAG_STAGE + 2 + 8 * 256 + SMALL_MAX_D * SMALL_MAX_N + 2 * SMALL_MAX_N + 8 + SMALL_MAX_D;
constexpr int SMALL_LDS_BYTES = SMALL_LDS_DOUBLES * 8;

// The body shared by the one-handle kernel (hyper-parameters in the kernel arguments) and the batched one (S sets of
// hyper-parameters on the same data, one workgroup per set: boss_gp_loglike_batch at the reference's own sizes).
// ilam: 1/λ in LDS (filled by the caller, visible after the first barrier below); par_dev / host_res / mean may be null.
__device__ __forceinline__ void small_fit_body(double* __restrict__ smem, int d, int N, int Np, int ld, int kern, double amp2,
                                               double sig2, const double* __restrict__ ilam, const double* __restrict__ Xraw,
                                               double* __restrict__ Xsc, const double* __restrict__ y,
                                               const double* __restrict__ mean, double* __restrict__ A,
                                               double* __restrict__ inv16, double* __restrict__ par_dev,
                                               double* __restrict__ scal, int* __restrict__ info,
                                               double* __restrict__ host_res, unsigned long long res_seq = 0) {
    double* Is = smem + DIAG_TILES * 256 + DIAG_STAGE + 2;   // scaled inverses of the diagonal tiles (inv16 format), 8 × 256
    double* xs = Is + 8 * 256;                               // scaled points [d][128]
    double* rhs = xs + SMALL_MAX_D * SMALL_MAX_N;            // y - m (then overwritten tile by tile with z)
    double* red = rhs + 2 * SMALL_MAX_N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nsub = (N + 15) / 16;
    __syncthreads();                                         // ilam is in place
    // resident copies of the hyper-parameters for the entry points that follow (prediction, append, gradients)
    if (par_dev) {
        if (tid < d) par_dev[tid] = ilam[tid];
        if (tid == 0) {
            par_dev[d] = amp2;
            par_dev[d + 1] = sig2;
        }
    }
    for (int idx = tid; idx < d * Np; idx += DIAG_THREADS) {
        const int k = idx / Np, j = idx - k * Np;
        const double v = Xraw[idx] * ilam[k];                // ARDTransform(1 ./ λ), gaussian_process.jl:243
        Xsc[idx] = v;
        if (j < SMALL_MAX_N) xs[k * SMALL_MAX_N + j] = v;
    }
    if (tid < SMALL_MAX_N) rhs[tid] = (tid < N) ? y[tid] - (mean ? mean[tid] : 0.0) : 0.0;
    __syncthreads();
    // Gram tiles K = α² κ(r) + σ² I, padding rows / columns of the last panel = identity
    const int ntl = nsub * (nsub + 1) / 2;
    for (int idx = tid; idx < ntl * 256; idx += DIAG_THREADS) {
        const int tl = idx >> 8, e = idx & 255;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= tl) ++ti;
        const int tj = tl - ti * (ti + 1) / 2;
        const int r = e & 15, c = e >> 4, i = ti * 16 + r, j = tj * 16 + c;
        double v = 0.0;
        if (i >= j) {
            if (i < N && j < N) {
                double r2 = 0.0;
                for (int k = 0; k < d; ++k) {
                    const double diff = xs[k * SMALL_MAX_N + i] - xs[k * SMALL_MAX_N + j];
                    r2 = __builtin_fma(diff, diff, r2);
                }
                v = amp2 * kappa_r2(kern, r2) + ((i == j) ? sig2 : 0.0);
            } else {
                v = (i == j) ? 1.0 : 0.0;
            }
        }
        smem[tl * 256 + c * 16 + r] = v;
    }
    __syncthreads();
    int fail = -1;
    diag_block_factor<true>(smem, A, ld, inv16, 0, nsub, fail, Is);
    __syncthreads();
    if (wv == 0) {
        // z = L \ (y - m), tile by tile: z_jb = inv(L16_jb) (δ_jb − Σ_{m<jb} L(jb, m) z_m); lane l carries the running sums of
        // rows l and l + 64
        const int r16 = lane & 15, q = lane >> 4;
        double acc0 = 0.0, acc1 = 0.0;
        for (int jb = 0; jb < nsub; ++jb) {
            // v = δ_jb − acc_jb
            const int row0 = 16 * jb;
            if (lane >= (row0 & 63) && lane < (row0 & 63) + 16) rhs[row0 + (lane - (row0 & 63))] -= (row0 < 64) ? acc0 : acc1;
            // z(r) = Σ_c X(r, c) v(c),  X(r, c) at c*16 + r: lane (r16, q) sums c = q, q+4, q+8, q+12, then the four quarters add up
            double zs = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = q + 4 * i;
                zs = __builtin_fma(Is[jb * 256 + c * 16 + r16], rhs[row0 + c], zs);
            }
            zs += __shfl_xor(zs, 16);
            zs += __shfl_xor(zs, 32);
            if (q == 0) rhs[SMALL_MAX_N + row0 + r16] = zs;                       // z lives behind the right-hand side
            // acc(row) += Σ_c L(row, 16jb + c) z(c) for the rows below this panel
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = lane + 64 * h, ti = row >> 4;
                if (ti > jb && ti < nsub) {
                    const double* Lt = smem + dtile(ti, jb) + (row & 15);
                    double a = 0.0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) a = __builtin_fma(Lt[c * 16], rhs[SMALL_MAX_N + row0 + c], a);
                    if (h == 0) acc0 += a;
                    else acc1 += a;
                }
            }
        }
        // logdet = 2 Σ log L_ii, zz = Σ z_i²  (i < N)
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = lane + 64 * h;
            if (i < N) {
                s0 += log(smem[dtile(i >> 4, i >> 4) + (i & 15) * 17]);
                const double z = rhs[SMALL_MAX_N + i];
                s1 = __builtin_fma(z, z, s1);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s0 += __shfl_xor(s0, off);
            s1 += __shfl_xor(s1, off);
        }
        if (lane == 0) {
            const int inf = fail >= 0 ? fail + 1 : 0;
            scal[0] = 2.0 * s0;
            scal[1] = s1;
            info[0] = inf;
            if (host_res) {
                host_res[0] = 2.0 * s0;
                host_res[1] = s1;
                reinterpret_cast<int*>(host_res + 2)[0] = inf;
                // (polled by gp_finish; the factor tiles and the z row follow below — every consumer of those is stream-ordered)
                __threadfence_system();
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_res + 3), res_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    // factor tiles (diagonal ones included, zero above the diagonal) and the z row
    for (int tl = wv; tl < ntl; tl += DIAG_THREADS / 64) {
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= tl) ++ti;
        const int tj = tl - ti * (ti + 1) / 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = lane + 64 * h, c = e >> 3, rp = e & 7;
            *reinterpret_cast<v2d*>(A + (size_t)(tj * 16 + c) * ld + ti * 16 + 2 * rp) =
                *reinterpret_cast<const v2d*>(smem + tl * 256 + c * 16 + 2 * rp);
        }
    }
    for (int c = tid; c < Np; c += DIAG_THREADS) {
        A[(size_t)c * ld + Np] = (c < 16 * nsub) ? rhs[SMALL_MAX_N + c] : 0.0;
        if (c >= 16 * nsub) A[(size_t)c * ld + c] = 1.0;                          // identity padding of the factor
    }
    // identity inverses of the padding panels (the panel solves of an append and the block inverses read them)
    const int npan = Np / 16;
    for (int idx = tid; idx < (npan - nsub) * 256; idx += DIAG_THREADS) {
        const int pnl = nsub + (idx >> 8), e = idx & 255;
        inv16[pnl * 256 + e] = ((e & 15) == (e >> 4)) ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(DIAG_THREADS) void small_fit_kernel(SmallFitPar par, const double* __restrict__ Xraw,
                                                                 double* __restrict__ Xsc, const double* __restrict__ y,
                                                                 const double* __restrict__ mean, double* __restrict__ A,
                                                                 double* __restrict__ inv16, double* __restrict__ par_dev,
                                                                 double* __restrict__ scal, int* __restrict__ info,
                                                                 double* __restrict__ host_res, unsigned long long res_seq) {
    extern __shared__ double smem[];
    double* ilam = smem + SMALL_LDS_DOUBLES - SMALL_MAX_D;
    if (threadIdx.x < par.d) ilam[threadIdx.x] = par.invlam[threadIdx.x];
    small_fit_body(smem, par.d, par.N, par.Np, par.ld, par.kern, par.amp2, par.sig2, ilam, Xraw, Xsc, y, mean, A, inv16, par_dev, scal,
                   info, host_res, res_seq);
}

// One workgroup per hyper-parameter set (blockIdx.x): 1/λ of set b at invlam[b·d ..], {α², σ²} at hyp[2b ..]; the sets' factor
// arrays, scaled points, inverses, results and prior means lie at the given strides (mean: null, shared (stride 0) or per set).
__global__ __launch_bounds__(DIAG_THREADS) void small_fit_batch_kernel(int d, int N, int Np, int ld, int kern,
                                                                       const double* __restrict__ invlam,
                                                                       const double* __restrict__ hyp,
                                                                       const double* __restrict__ Xraw, double* __restrict__ Xsc,
                                                                       size_t sX, const double* __restrict__ y,
                                                                       const double* __restrict__ mean, size_t sMean,
                                                                       double* __restrict__ A, size_t sA, double* __restrict__ inv16,
                                                                       size_t sInv, double* __restrict__ scal, int* __restrict__ info) {
    extern __shared__ double smem[];
    const int b = blockIdx.x;
    double* ilam = smem + SMALL_LDS_DOUBLES - SMALL_MAX_D;
    if (threadIdx.x < d) ilam[threadIdx.x] = invlam[(size_t)b * d + threadIdx.x];
    small_fit_body(smem, d, N, Np, ld, kern, hyp[2 * b], hyp[2 * b + 1], ilam, Xraw, Xsc + (size_t)b * sX, y,
                   mean ? mean + (size_t)b * sMean : nullptr, A + (size_t)b * sA, inv16 + (size_t)b * sInv, nullptr, scal + 2 * b,
                   info + b, nullptr);
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

// Panel solve, TWO waves per 16-row group.  One wave alone walks 176 MFMAs that mostly depend on each other (≈6 µs, the
// longest stage of the panel chain after the diagonal block).  Here wave 0 solves the first TRSM_NA column tiles and hands
// each finished tile to wave 1 through LDS (same lane mapping on both sides: the accumulator layout IS the B-operand
// layout); wave 1 folds them into the remaining tiles as they arrive — consecutive MFMAs on different accumulators — and
// finishes its own small triangle: ≈88 MFMA times on the critical path.  Every tile still receives its contributions in
// ascending column order, so the result is bit-identical to the one-wave form (wave_trsm16, kept for the block inverses).
constexpr int TRSM_NA = 5;
constexpr int TRSM_THREADS = 128;
__global__ __launch_bounds__(TRSM_THREADS) void potrf_trsm_kernel(double* __restrict__ Abase, int ld, size_t bstride, int k,
                                                                  const double* __restrict__ inv16base,
                                                                  size_t inv16_bstride, int row0) {
    // row0: first row solved by 16-row group 0 — (k+1)*BLK in the factorisation (everything below
    // the diagonal block); boss_gp_append solves only the block row it rebuilds, or only the δ^T rows
    __builtin_amdgcn_s_setprio(3);           // a chain kernel: its waves win issue arbitration over co-resident bulk-update waves
    __shared__ v4d xs[TRSM_NA][64];
    __shared__ int ready;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* A = Abase + (size_t)blockIdx.z * bstride;
    const double* Lkk = A + (size_t)k * BLK * ((size_t)ld + 1);
    const double* inv16k = inv16base + (size_t)blockIdx.z * inv16_bstride + (size_t)k * (8 * 256);
    double* Brow = A + (size_t)row0 + (size_t)blockIdx.x * 16 + (size_t)k * BLK * ld;
    if (threadIdx.x == 0) *(volatile lds_int_t*)&ready = 0;
    __syncthreads();
    // Every operand is fetched up front (one memory round trip per wave; the hand-off spins below are compiler barriers, loads
    // left behind them would each expose their latency): lv(jb, m, s) = L(row = jb*16 + (lane&15), col = m*16 + 4s + (lane>>4)),
    // iv(jb, s) = inv16_jb(4s + (lane>>4), lane&15).
    auto lval = [&](int jb, int m, int s) { return Lkk[(size_t)(m * 16 + 4 * s + (lane >> 4)) * ld + jb * 16 + (lane & 15)]; };
    auto ival = [&](int jb, int s) { return inv16k[jb * 256 + (4 * s + (lane >> 4)) * 16 + (lane & 15)]; };
    if (wave == 0) {
        v4d acc[TRSM_NA];
        double lv[TRSM_NA][TRSM_NA][4], iv[TRSM_NA][4];
#pragma unroll
        for (int jb = 0; jb < TRSM_NA; ++jb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[jb][i] = Brow[(lane & 15) + (size_t)(jb * 16 + (lane >> 4) + 4 * i) * ld];
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
        for (int jb = 0; jb < TRSM_NA; ++jb) {
#pragma unroll
            for (int m = 0; m < jb; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[jb] = mfma_f64(-lv[jb][m][s], acc[m][s], acc[jb]);
            v4d nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) nw = mfma_f64(iv[jb][s], acc[jb][s], nw);
            acc[jb] = nw;
            xs[jb][lane] = nw;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile lds_int_t*)&ready = jb + 1;
        }
#pragma unroll
        for (int jb = 0; jb < TRSM_NA; ++jb)
#pragma unroll
            for (int i = 0; i < 4; ++i) Brow[(lane & 15) + (size_t)(jb * 16 + (lane >> 4) + 4 * i) * ld] = acc[jb][i];
    } else {
        constexpr int NB = 8 - TRSM_NA;
        v4d acc[NB];
        double lu[TRSM_NA][NB][4], lo[NB][NB][4], iv[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = Brow[(lane & 15) + (size_t)((TRSM_NA + j) * 16 + (lane >> 4) + 4 * i) * ld];
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
        for (int m = 0; m < TRSM_NA; ++m) {
#pragma unroll 1
            while (*(volatile lds_int_t*)&ready <= m) __builtin_amdgcn_s_sleep(1);   // (wave 0 of this workgroup: it never waits for this wave)
            asm volatile("" ::: "memory");
            const v4d x = xs[m][lane];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = mfma_f64(-lu[m][j][s], x[s], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int m = 0; m < j; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[j] = mfma_f64(-lo[j][m][s], acc[m][s], acc[j]);
            v4d nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) nw = mfma_f64(iv[j][s], acc[j][s], nw);
            acc[j] = nw;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) Brow[(lane & 15) + (size_t)((TRSM_NA + j) * 16 + (lane >> 4) + 4 * i) * ld] = acc[j][i];
    }
}

// ------------------------------------------------------------------------------------------
// Gradient of the log marginal likelihood for N <= 128 observations in ONE launch (the reference's own regime: its model
// fitters call data_loglike under AD with tens of points, src/model_fitters/optimization.jl:146-164).  The general path
// (grad_kernels.hpp: block inverses, L⁻¹ by recursive doubling, K⁻¹ as a GEMM, tile partials, reduction — ten launches and
// a copy) costs 0.14 ms at N = 20, six times the single-launch fit in front of it.  Here one workgroup does all of it:
//   waves 0-7   L⁻¹ by the register-resident substitution of the block inverses (wave_trsm16<true>), tile-wise into LDS
//   all         a = L⁻ᵀ z
//   16 waves    K⁻¹ tile by tile with MFMAs straight into registers, and on those registers the pair sums
//                   S_m = Σ_{i>j} (a_i a_j − K⁻¹_ij) α² h(r_ij) Δu²_ij,m ,  tr K⁻¹ ,  ‖a‖²
// — the Σ-vector llgrad_finalize turns into ∂ℓ/∂(λ, α, σ) — written to mapped host memory.  Same sums as llgrad_tile_kernel
// (every pair once, diagonal terms once); the order of the floating-point additions differs, the tolerance of the tests does not.
// ------------------------------------------------------------------------------------------
constexpr int SMALL_LLG_LDS_DOUBLES = DIAG_TILES * 256 + SMALL_MAX_D * SMALL_MAX_N + SMALL_MAX_N + 16 * (SMALL_MAX_D + 2);
constexpr int SMALL_LLG_LDS_BYTES = SMALL_LLG_LDS_DOUBLES * 8;
// blockIdx.x: set of a batch — factor, inverses and scaled points at strides sA / sInv / sX, α² at amp2p[b·amp2_stride] (null: the
// scalar argument), the Σ-vector to out + b·out_stride (one handle: mapped host memory, a batch: device memory).
__global__ __launch_bounds__(DIAG_THREADS) void small_llgrad_kernel(const double* __restrict__ A, int ld, int Np, int N, int d,
                                                                    int kern, double amp2, const double* __restrict__ inv16,
                                                                    const double* __restrict__ Xsc, int ldx,
                                                                    double* __restrict__ host_out, size_t sA, size_t sInv, size_t sX,
                                                                    const double* __restrict__ amp2p, int amp2_stride, int out_stride,
                                                                    unsigned long long res_seq) {
    extern __shared__ double smem[];
    A += (size_t)blockIdx.x * sA;
    inv16 += (size_t)blockIdx.x * sInv;
    Xsc += (size_t)blockIdx.x * sX;
    host_out += (size_t)blockIdx.x * out_stride;
    if (amp2p) amp2 = amp2p[(size_t)blockIdx.x * amp2_stride];
    double* Li = smem;                                   // L⁻¹, lower 16×16 tiles (i, j) at dtile(i, j), ROW-major inside: [row][col]
    double* xs = Li + DIAG_TILES * 256;                  // scaled points [d][128]
    double* av = xs + SMALL_MAX_D * SMALL_MAX_N;         // a = L⁻ᵀ z
    double* red = av + SMALL_MAX_N;                      // per-wave partial sums [16][d + 2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsub = (N + 15) / 16;
    for (int idx = tid; idx < d * SMALL_MAX_N; idx += DIAG_THREADS) {
        const int k = idx / SMALL_MAX_N, j = idx - k * SMALL_MAX_N;
        xs[idx] = (j < N) ? Xsc[(size_t)k * ldx + j] : 0.0;
    }
    if (tid >= DIAG_THREADS - SMALL_MAX_N) {             // z (row Np of the factor array) parked in `red` until a is formed
        const int k = tid - (DIAG_THREADS - SMALL_MAX_N);
        red[k] = (k < N) ? A[(size_t)k * ld + Np] : 0.0;
    }
    if (wave < 8) {
        // rows r0..r0+15 of L⁻ᵀ (= columns of L⁻¹): P(r, c) = L⁻ᵀ(r, c), r = r0 + (l&15), c = jb·16 + (l>>4) + 4i
        const int r0 = wave * 16;
        v4d acc[8];
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[jb][i] = ((r0 + (lane & 15)) == (jb * 16 + (lane >> 4) + 4 * i)) ? 1.0 : 0.0;
        if (wave < nsub) wave_trsm16<true>(A, ld, inv16, acc, lane);
        // L⁻¹(k = c, i = r) = P(r, c) for k >= i: tile (kb = jb, ib = wave), element [k & 15][i & 15]
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
            if (jb < wave || jb >= nsub || wave >= nsub) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = jb * 16 + (lane >> 4) + 4 * i, r = r0 + (lane & 15);
                Li[dtile(jb, wave) + ((lane >> 4) + 4 * i) * 16 + (lane & 15)] = (k >= r) ? acc[jb][i] : 0.0;
            }
        }
    }
    __syncthreads();
    if (tid < SMALL_MAX_N) {
        // a_i = Σ_{k >= i} L⁻¹(k, i) z_k,  z_k in row Np of the factor array
        const int i = tid, ib = i >> 4;
        double s = 0.0;
        if (i < N)
            for (int k = i; k < N; ++k) s = __builtin_fma(Li[dtile(k >> 4, ib) + (k & 15) * 16 + (i & 15)], red[k], s);
        av[i] = s;
    }
    __syncthreads();
    double S[SMALL_MAX_D];
#pragma unroll
    for (int m = 0; m < SMALL_MAX_D; ++m) S[m] = 0.0;
    double tr = 0.0, aa = 0.0;
    const int T = nsub * (nsub + 1) / 2;
    for (int tl = wave; tl < T; tl += DIAG_THREADS / 64) {
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= tl) ++bi;
        const int bj = tl - bi * (bi + 1) / 2;
        // K⁻¹(i, j) = Σ_{k >= i} L⁻¹(k, i) L⁻¹(k, j): MFMA rows M <-> i, columns N <-> j, contraction over the tiles kb >= bi
        v4d kin = {0.0, 0.0, 0.0, 0.0};
        for (int kb = bi; kb < nsub; ++kb) {
            const double* Ti = Li + dtile(kb, bi);
            const double* Tj = Li + dtile(kb, bj);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int o = (4 * s4 + (lane >> 4)) * 16 + (lane & 15);     // element [k = 4s + (l>>4)][col = l&15]
                kin = mfma_f64(Ti[o], Tj[o], kin);
            }
        }
        // lane l, register r: K⁻¹(i = bi·16 + (l>>4) + 4r, j = bj·16 + (l&15))
        const int j = bj * 16 + (lane & 15);
        const double aj = av[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = bi * 16 + (lane >> 4) + 4 * r;
            if (i >= N || j >= N || i < j) continue;
            if (i == j) {
                tr += kin[r];
                aa = __builtin_fma(aj, aj, aa);
                continue;
            }
            double r2 = 0.0;                               // (the differences are formed twice: 1024 threads leave no room for a second array)
            for (int m = 0; m < d; ++m) {
                const double df = xs[m * SMALL_MAX_N + i] - xs[m * SMALL_MAX_N + j];
                r2 += df * df;
            }
            const double q = (av[i] * aj - kin[r]) * amp2 * kappa_prime_over_r_r2(kern, r2);
#pragma unroll
            for (int m = 0; m < SMALL_MAX_D; ++m) {
                if (m < d) {
                    const double df = xs[m * SMALL_MAX_N + i] - xs[m * SMALL_MAX_N + j];
                    S[m] = __builtin_fma(q, df * df, S[m]);
                }
            }
        }
    }
    // wave sums (fixed butterfly order), then the 16 wave partials in wave order
    const int nv = d + 2;
    for (int m = 0; m < nv; ++m) {
        double v = (m == d) ? tr : aa;
#pragma unroll
        for (int mm = 0; mm < SMALL_MAX_D; ++mm)
            if (mm == m) v = S[mm];
        if (m >= d) v = (m == d) ? tr : aa;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[wave * (SMALL_MAX_D + 2) + m] = v;
    }
    __syncthreads();
    if (tid < nv) {
        double v = 0.0;
        for (int w = 0; w < DIAG_THREADS / 64; ++w) v += red[w * (SMALL_MAX_D + 2) + tid];
        host_out[tid] = v;
        if (res_seq) __threadfence_system();                 // one handle: the host polls the word below (mapped memory)
    }
    if (res_seq) {
        __syncthreads();
        if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out) - 1, res_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------
// Posterior moments AND their gradients w.r.t. the candidates for N <= 128 observations in one launch (what one step of a
// gradient-based acquisition maximiser asks for, src/acquisition_maximizers/optimization.jl:59-77, at the reference's own data
// sizes).  The general path (prediction kernel, adjoint substitution, gradient accumulation, finalisation: six launches) costs
// 0.15 ms per call whatever M is.  Here every wave takes one candidate at a time against L⁻¹ held in LDS (packed lower columns,
// from the block inverse Dinv: Dinv[i·128 + k] = L⁻¹(k, i)):
//   k*_i = α² κ(r_i),  v = L⁻¹ k*  (column sweeps: lanes = rows),  μ = m + vᵀz,  σ² = α² − vᵀv      (gaussian_process.jl:168-184)
//   w = L⁻ᵀ v  (lane i: the dot of column i with v),  a = L⁻ᵀ z (once per workgroup)
//   ∂μ/∂x_m = (1/λ_m) Σ_i a_i α² h(r_i)(u*_m − u_i,m),   ∂σ²/∂x_m = −(2/λ_m) Σ_i w_i α² h(r_i)(u*_m − u_i,m)
// with the conventions of grad_finalize_kernel (discrete dimensions: gradient 0 and rounded coordinate; prior-mean gradient
// added to ∂μ).  Variances leave unclipped, like from the prediction kernel (the caller's clip kernel follows).
// ------------------------------------------------------------------------------------------
constexpr int SPG_THREADS = 256, SPG_WAVES = SPG_THREADS / 64;
constexpr int SPG_LDS_DOUBLES = SMALL_MAX_N * (SMALL_MAX_N + 1) / 2 + SMALL_MAX_D * SMALL_MAX_N + 2 * SMALL_MAX_N +
                                SPG_WAVES * (2 * SMALL_MAX_N + SMALL_MAX_D);
constexpr int SPG_LDS_BYTES = SPG_LDS_DOUBLES * 8;
// GRAD = false: the moments alone (boss_gp_predict / boss_acq_ei at these sizes: scaling, K*, substitution and moments in one launch).
template <bool GRAD>
__global__ __launch_bounds__(SPG_THREADS) void small_predict_grad_kernel(const double* __restrict__ Dinv, const double* __restrict__ A,
                                                                         int ld, int Np, int N, int d, int kern, double amp2,
                                                                         const double* __restrict__ Xsc, int ldx,
                                                                         const double* __restrict__ Craw, int Mp, int M,
                                                                         const double* __restrict__ invlam,
                                                                         const unsigned char* __restrict__ discrete,
                                                                         const double* __restrict__ mean_s,
                                                                         const double* __restrict__ mean_grad,
                                                                         double* __restrict__ mu, double* __restrict__ var,
                                                                         double* __restrict__ dmu, double* __restrict__ dvar) {
    extern __shared__ double smem[];
    double* Lc = smem;                                        // L⁻¹, packed lower columns: column i at coff(i), rows i..N-1
    double* xs = Lc + SMALL_MAX_N * (SMALL_MAX_N + 1) / 2;    // scaled training points [d][128]
    double* zs = xs + SMALL_MAX_D * SMALL_MAX_N;              // z
    double* as = zs + SMALL_MAX_N;                            // a = L⁻ᵀ z
    double* wbuf = as + SMALL_MAX_N;                          // per wave: k* / v [128], α² h [128], u* [32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto coff = [N](int i) { return i * N - (i * (i - 1)) / 2; };
    for (int idx = tid; idx < N * SMALL_MAX_N; idx += SPG_THREADS) {
        const int i = idx / SMALL_MAX_N, k = idx - i * SMALL_MAX_N;
        if (k >= i && k < N) Lc[coff(i) + (k - i)] = Dinv[(size_t)i * BLK + k];
    }
    for (int idx = tid; idx < d * SMALL_MAX_N; idx += SPG_THREADS) {
        const int m = idx / SMALL_MAX_N, i = idx - m * SMALL_MAX_N;
        xs[idx] = (i < N) ? Xsc[(size_t)m * ldx + i] : 0.0;
    }
    if (tid < SMALL_MAX_N) zs[tid] = (tid < N) ? A[(size_t)tid * ld + Np] : 0.0;
    __syncthreads();
    if (GRAD && tid < SMALL_MAX_N) {
        double s = 0.0;
        if (tid < N) {
            const double* col = Lc + coff(tid);
            for (int k = tid; k < N; ++k) s = __builtin_fma(col[k - tid], zs[k], s);
        }
        as[tid] = s;
    }
    __syncthreads();
    double* kv = wbuf + wave * (2 * SMALL_MAX_N + SMALL_MAX_D);   // k*, then v
    double* hv = kv + SMALL_MAX_N;                               // α² h(r_i)
    double* us = hv + SMALL_MAX_N;                               // u* of this wave's candidate
    for (int j = blockIdx.x * SPG_WAVES + wave; j < M; j += gridDim.x * SPG_WAVES) {
        if (lane < d) {
            double c = Craw[(size_t)lane * Mp + j];
            if (discrete && discrete[lane]) c = rint(c);
            us[lane] = c * invlam[lane];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // one wave: its LDS operations execute in order
        // rows i = lane, lane + 64
        double v0 = 0.0, v1 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = lane + 64 * h;
            double r2 = 0.0;
            for (int m = 0; m < d; ++m) {
                const double df = us[m] - xs[m * SMALL_MAX_N + i];
                r2 = __builtin_fma(df, df, r2);
            }
            kv[i] = (i < N) ? amp2 * kappa_r2(kern, r2) : 0.0;
            if (GRAD) hv[i] = (i < N) ? amp2 * kappa_prime_over_r_r2(kern, r2) : 0.0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // v = L⁻¹ k*: sweep the columns, lane = row
        for (int i = 0; i < N; ++i) {
            const double ki = kv[i];
            const double* col = Lc + coff(i) - i;            // col[k] = L⁻¹(k, i), k >= i
            if (lane >= i) v0 = __builtin_fma(col[lane], ki, v0);
            if (lane + 64 >= i && lane + 64 < N) v1 = __builtin_fma(col[lane + 64], ki, v1);
        }
        if (lane >= N) v0 = 0.0;
        double sm = v0 * zs[lane] + v1 * zs[lane + 64], sv = v0 * v0 + v1 * v1;
        if (!GRAD) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                sm += __shfl_xor(sm, off);
                sv += __shfl_xor(sv, off);
            }
            if (lane == 0) {
                mu[j] = (mean_s ? mean_s[j] : 0.0) + sm;
                var[j] = amp2 - sv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every lane has read k* before v overwrites it
        kv[lane] = v0;
        kv[lane + 64] = v1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // w_i = Σ_{k >= i} L⁻¹(k, i) v_k
        double w0 = 0.0, w1 = 0.0;
        if (lane < N) {
            const double* col = Lc + coff(lane) - lane;
            for (int k = lane; k < N; ++k) w0 = __builtin_fma(col[k], kv[k], w0);
        }
        if (lane + 64 < N) {
            const double* col = Lc + coff(lane + 64) - (lane + 64);
            for (int k = lane + 64; k < N; ++k) w1 = __builtin_fma(col[k], kv[k], w1);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sm += __shfl_xor(sm, off);
            sv += __shfl_xor(sv, off);
        }
        // gradient sums over the rows
        const double ah0 = as[lane] * hv[lane], ah1 = as[lane + 64] * hv[lane + 64];
        const double wh0 = w0 * hv[lane], wh1 = w1 * hv[lane + 64];
        double s1 = ah0 + ah1, s2 = wh0 + wh1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
        }
        if (lane == 0) {
            mu[j] = (mean_s ? mean_s[j] : 0.0) + sm;
            var[j] = amp2 - sv;
        }
        for (int m = 0; m < d; ++m) {
            const double x0 = xs[m * SMALL_MAX_N + lane], x1 = xs[m * SMALL_MAX_N + lane + 64];
            double t1 = ah0 * x0 + ah1 * x1, t2 = wh0 * x0 + wh1 * x1;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                t1 += __shfl_xor(t1, off);
                t2 += __shfl_xor(t2, off);
            }
            if (lane == 0) {
                const double u = us[m], il = invlam[m];
                const bool disc = discrete && discrete[m];
                dmu[(size_t)j * d + m] = (disc ? 0.0 : (u * s1 - t1) * il) + (mean_grad ? mean_grad[(size_t)j * d + m] : 0.0);
                dvar[(size_t)j * d + m] = disc ? 0.0 : -2.0 * (u * s2 - t2) * il;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // us / kv are rewritten for the next candidate
    }
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
#define BOSS_RHS_D 4      // ring depth of the strip GEMM.  It must DIVIDE the 32 k-substeps of a K = 128 strip: the ring's tail is then exact (gemm_f64.hpp: the last D substeps are consumed as they land — no clamped reloads, no full drain in front of a remainder).  ms per N = 4096 update with the exact tail, round 4: D = 2: 1.199, 4: 1.135–1.142, 8: 1.140–1.143, 16: 1.248; D = 12 (round 3's depth, drained tail): 1.161
#endif
typedef GemmDirect<2, 2, 4, 4, 4> SyrkG;   // 128×128 tile, fragments streamed from L2, no LDS
typedef GemmDirect<1, 4, 2, 2, BOSS_RHS_D> RhsG;    // 32×128 tile for the δ^T row block and the column updates
constexpr int SYRK_LDS_BYTES = 0;

// acc is INITIALISED from C (its load latency overlaps the operand prologue), updated with
// acc -= P_i P_j^T, and stored back: the epilogue is pure stores.
// SC1: the tile is stored write-through (a resident kernel on another CU reads it while this kernel is still running)
template <class G, bool SC1 = false>
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
    G::template run<-1, (BLK / 4) % G::D == 0>(Pi, ld, Pj, ld, K, acc);   // (K is a multiple of BLK: exact ring tail where D divides 32)
#pragma unroll
    for (int m = 0; m < G::TM; m += 2)
#pragma unroll
        for (int n = 0; n < G::TN; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v2d c2 = {acc[m][n][i], acc[m + 1][n][i]};
                double* dst = C + G::row_of(wr, m, lane) + (size_t)G::col_of(wc, n, i, lane) * ld;
                if constexpr (SC1) st_sc1_x4(dst, c2);
                else *reinterpret_cast<v2d*>(dst) = c2;
            }
}

// Trailing update behind panel k over the block triangle that starts at block `first` and has m
// square block rows (+ the δ^T row block at first+m == Np/128):  C_ij -= P_i P_j^T.
//   potrf_syrk_kernel   all of it (first = k+1) or, with look-ahead, everything beyond the next
//                       panel (first = k+2) — runs on the side stream, 128×128 tiles;
//   potrf_colupd_kernel only block column k+1 (the next panel), 32×128 tiles so that this short
//                       kernel on the critical path is one small round.
typedef GemmDirect<2, 2, 2, 4, 4> SyrkHalfG;   // 64×128 half tile: workgroups half as long (look-ahead: the chain's kernels get CUs sooner)

template <int NPAN, bool HALF = false>
__global__ __launch_bounds__(256, 2) void potrf_syrk_kernel(double* __restrict__ Abase, int ld, size_t bstride, int k,
                                                            int first, int m, int batch1d, int skip00 = 0) {
    if (skip00 && blockIdx.x == 0) return;                  // tile (first, first) is updated by the column-update kernel (chain schedule)
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
        if constexpr (HALF) syrk_tile<SyrkHalfG>(A, ld, k, (first + i) * BLK + 64 * blockIdx.y, (first + j) * BLK, npan * BLK);
        else syrk_tile<SyrkG>(A, ld, k, (first + i) * BLK, (first + j) * BLK, npan * BLK);
    } else {
        if (HALF && blockIdx.y != 0) return;
        int j = t - nsq;
        syrk_tile<RhsG>(A, ld, k, (first + m) * BLK, (first + j) * BLK, npan * BLK);
    }
}

// Workgroup -> strip of potrf_colupd_kernel (host-callable: tests/test_abi_and_host.py walks every launch form of the schedule
// through boss_debug_colupd_decode and checks that each 32×128 strip is taken by exactly ONE workgroup — syrk_tile is a plain
// read-modify-write, two workgroups on one strip would apply the panel twice).
//   R0 < 0: nothing to do;  crit: one of the eight critical strips of the chain schedule.
struct ColupdWork {
    int R0, C0, crit;
};
__host__ __device__ inline ColupdWork colupd_decode(int t, int G, int k, int m, int ncols, int jfirst, int skipdiag, int xblk, bool critical) {
    if (m < 2) critical = false;                             // no block row k+2: the swap below would not be a permutation (m == 1, a grid
                                                             // of 5: every workgroup landed on the δ^T strip of the last block column)
    if (critical) {
        // dispatch order: the eight critical strips first — tile (k+2, k+1) = strips 4..7 of this list, tile (k+2, k+2) = the extra
        // strips at its end (even steps) or the first four strips of the second column — swapped with strips 0..3 (the left-out
        // diagonal tile: nothing to do) and 4..7
        const int d0 = (xblk == k + 2) ? G - 4 : 4 * m + 1;
        if (t < 4) t = 4 + t;
        else if (t < 8) t = d0 + (t - 4);
        else if (t >= d0 && t < d0 + 4) t = t - d0;
    }
    if (xblk >= 0 && t >= G - 4) {
        ColupdWork w = {xblk * BLK + (t - (G - 4)) * 32, xblk * BLK, (critical && xblk == k + 2) ? 1 : 0};
        return w;
    }
    if (skipdiag && t < 4) {                                  // column k+1's strips 0..3 = its diagonal tile
        ColupdWork w = {-1, -1, 0};
        return w;
    }
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
    ColupdWork w = {R0, (k + 1 + j) * BLK, (critical && t < nstr && R0 / BLK == k + 2 && j <= 1) ? 1 : 0};   // tile (k+2, k+1) or (k+2, k+2)
    return w;
}

// skipdiag: the four strips of tile (k+1, k+1) are left out — the persistent chain applies panel k to that tile itself (chain.hpp);
// xblk >= 0: four extra strips update the diagonal tile (xblk, xblk) with the same panel(s), so that every diagonal tile has
// received every panel but the last one a whole step before its own factorisation (even steps: xblk = k+2; odd steps: k+3,
// which the bulk update then leaves out).
__global__ __launch_bounds__(256, 2) void potrf_colupd_kernel(double* __restrict__ Abase, int ld, size_t bstride, int k,
                                                              int m, int ncols, int jfirst, int npan,
                                                              unsigned long long* wword, unsigned long long wval, int* info,
                                                              int skipdiag, int xblk, unsigned long long* critw, unsigned budget) {
#ifdef BOSS_CHAIN_TRACE
    if (threadIdx.x == 0) atomicMin(&g_cutrace[(k & 63) * 4 + 0], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    struct TraceEnd { int k; __device__ ~TraceEnd() { if (threadIdx.x == 0) atomicMax(&g_cutrace[(k & 63) * 4 + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } } trace_end_{k};
#endif
    wait_word(wword, wval, info, budget);                    // (look-ahead schedule: the bulk update that wrote these columns before)
    // npan = 2: apply the TWO panels k-1, k (K = 256) — the odd steps of the paired look-ahead schedule
    // 32×128 tiles.  ncols = 1: only block column k+1 (look-ahead: the next panel) — 4 strips per
    // 128-row block (m blocks) + one strip of the δ^T rows.  ncols = m: the whole trailing triangle
    // (used for the last steps, where one small launch beats the two-stream choreography).
    __builtin_amdgcn_s_setprio(3);           // a chain kernel: its waves win issue arbitration over co-resident bulk-update waves
    double* A = Abase + (size_t)blockIdx.z * bstride;
    if (m < 2) critw = nullptr;                              // no block row k+2: no critical strips (see colupd_decode)
    const ColupdWork w = colupd_decode((int)blockIdx.x, (int)gridDim.x, k, m, ncols, jfirst, skipdiag, xblk, critw != nullptr);
    if (w.R0 < 0) return;                                     // a left-out strip (tile (k+1, k+1) under the chain schedule)
    // critical strips — tiles (k+2, k+1) and (k+2, k+2), what the resident chain needs first for step k+1 — store write-through
    // and count themselves in (eight per step); the resident kernels start on those tiles while the rest of this launch is still running
    if (w.crit) {
        if (npan == 2) syrk_tile<RhsG, true>(A, ld, k - 1, w.R0, w.C0, 2 * BLK);
        else syrk_tile<RhsG, true>(A, ld, k, w.R0, w.C0);
        drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(as_global(critw), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (npan == 2) syrk_tile<RhsG>(A, ld, k - 1, w.R0, w.C0, 2 * BLK);
    else syrk_tile<RhsG>(A, ld, k, w.R0, w.C0);
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
// host_out (or null; single matrix): mapped host memory that receives {logdet, zᵀz, info} directly — the update then ends
// with this kernel instead of a device-to-host copy behind it.  sig / sigval: as in potrf_diag_kernel.
__global__ __launch_bounds__(LOGDET_THREADS) void potrf_logdet_kernel(const double* __restrict__ Abase, int ld, size_t bstride,
                                                                      int N, int Np, double* __restrict__ scal,
                                                                      double* __restrict__ host_out, const int* __restrict__ info,
                                                                      unsigned long long* sig, unsigned long long sigval,
                                                                      unsigned long long res_seq) {
    if (sig && threadIdx.x == 0 && blockIdx.z == 0) __hip_atomic_store(sig, sigval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        if (host_out && blockIdx.z == 0) {
            host_out[0] = 2.0 * r0[0];
            host_out[1] = r1[0];
            reinterpret_cast<int*>(host_out + 2)[0] = info[0];
            // the host polls this word instead of waiting for the stream's completion signal (gp_finish): results first, then the number
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + 3), res_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace boss

// chain.hpp — the panel chain of the blocked Cholesky as ONE resident kernel, with followers in lockstep.
//
// The N = 4096 update is bound by the sequential chain  diag_k -> solve_k -> column update_k -> diag_{k+1}  (32 steps): as
// separate kernels each step costs the diagonal block (19 µs) PLUS the head of the panel solve, the head of the column update
// and three kernel boundaries (40-49 µs per step in all, profiles/r02_chain_timeline_final.log).  Here the diagonal blocks are
// factored by a persistent kernel of two workgroups that take the blocks alternately:
//   C(b)    factor diagonal block b (diag_block_factor<PUB>): every 16-column panel is published as soon as it is final;
//   F(b+1)  meanwhile the OTHER workgroup holds diagonal block b+1 in registers (36 lower tiles over 16 waves) and applies
//           P[b+1, b] P[b+1, b]^T to it panel by panel, as the panel solve of block row b+1 — the first eight 16-row strips of
//           potrf_follow_kernel, on other CUs, in lockstep with C(b) — delivers the 16×16 tiles.  When C(b) has published its
//           last panel, block b+1 is complete a few microseconds later (two hand-offs) and its workgroup goes straight into C(b+1).
// The step's critical path is the diagonal block plus two hand-offs; the rest of the panel solve, the column update and the bulk
// of the trailing update stay ordinary kernels on the library's streams, ordered among themselves by stream order and gates and
// against the chain by sequence words.
// Replaces the schedule around AbstractGPs.posterior's cholesky (src/models/gaussian_process.jl:199-211) for single matrices.
#pragma once
#include "potrf.hpp"

namespace boss {

#ifdef BOSS_CHAIN_TRACE
// device timeline of the resident chain (tools/chain_trace3.py builds a copy of the library with it): 100 MHz stamps
__device__ unsigned long long g_ctrace[64 * 16];
#define CTRACE(step, slot) do { if ((threadIdx.x & 63) == 0) g_ctrace[((step) & 63) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CTRACE_MIN(step, slot) do { if (threadIdx.x == 0) atomicMin(&g_ctrace[((step) & 63) * 16 + (slot)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define CTRACE_MAX(step, slot) do { if (threadIdx.x == 0) atomicMax(&g_ctrace[((step) & 63) * 16 + (slot)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#else
#define CTRACE(step, slot) do { } while (0)
#define CTRACE_MIN(step, slot) do { } while (0)
#define CTRACE_MAX(step, slot) do { } while (0)
#endif

// Nothing may stay in a vector register across this point.  The two phases of the chain kernel are inlined into one loop body;
// left to itself the register allocator keeps per-lane state of one phase alive through the other and spills inside both
// (follower rounds waiting on scratch loads: 10 µs each); as separate non-inlined functions each call saves and restores
// 40-48 callee-saved registers through scratch on the step's critical path.  Cutting every live range at the phase boundary
// costs one reload of the thread index per phase.
#define CHAIN_CUT_VGPRS() asm volatile("" ::: "memory", "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111")   // (v112-v127 stay: a zero accumulator the compiler keeps for every MFMA chain — cut too, it went through scratch at every cut —, SGPR spill lanes, the thread index)
constexpr int STRIPS_LDS_BYTES = 90 * 1024;                  // occupancy limiter of potrf_strips_kernel (see chain_begin)
constexpr int CHAIN_CTRL = DIAG_TILES * 256 + DIAG_STAGE;   // doubles: one control word behind the diagonal-block kernel's LDS
constexpr int CHAIN_LDS_BYTES = (CHAIN_CTRL + 2) * 8;       // two control words (ints): go-ahead of the block, posted panel count

// sig[SIGW_WDONE] >= base + k + 1 : follow kernel k has STARTED, i.e. everything enqueued before it on the main stream has
//                                   completed (column k of the matrix carries every panel < k; tile (k+1, k+1) every panel < k)
// sig[SIGW_PANEL] >= base + 8 b + jb + 1 : panel jb of diagonal block b is out (inverse of its diagonal tile, its row tiles)
// sig[SIGW_PROG + SIGW_PROG_STRIDE s] >= base + 8 k + jb + 1 : strip s (of the eight strips of block row k+1) has stored its tile of panel jb of step k
//
// The two phases are separate (non-inlined) functions: inlined into one loop body the register allocator keeps the factorisation's
// per-lane state alive through the follower phase and spills it (the follower rounds then wait on scratch loads: 10 µs each).

// F(b): diagonal block b (b >= 1) in registers, P[b, b-1] P[b, b-1]^T applied panel by panel as the strips deliver, result left in
// the LDS tile area.  Every wave works on its own: it polls the strips' progress words itself and takes its MFMA operands
// straight from global memory (sc1) — no staging through LDS, no workgroup barrier before the last tile is written (a round of
// "one wave polls, barrier, everybody loads, barrier, MFMAs, barrier" took 3-4 µs, on the critical path after C(b-1)'s last panel).
// A wave owns up to three tiles of ONE tile row (they share the row operand: 16 operand loads per panel instead of 24); when it
// has fallen behind it takes two panels per round trip, and the next poll travels with the operand loads.
// A wait that gives up marks the factorisation (info = INT_MIN) and the wave carries on without waiting: every loop of the
// kernel stays bounded and every barrier is reached by all waves.
__device__ const unsigned char chain_tile_tab[16][3] = {   // wave -> tile row, first tile column, number of tiles (36 lower tiles; wave 15: none, it polls)
    {7, 0, 3}, {7, 3, 3}, {7, 6, 2}, {6, 0, 3}, {6, 3, 2}, {6, 5, 2}, {5, 0, 3}, {5, 3, 3},
    {4, 0, 3}, {4, 3, 2}, {3, 0, 2}, {3, 2, 2}, {2, 0, 3}, {1, 0, 2}, {0, 0, 1}, {0, 0, 0}};
__device__ __forceinline__ void chain_follow_phase(double* __restrict__ A, int ld, int b, int* __restrict__ info,
                                                             unsigned long long* __restrict__ sig, unsigned long long base, int wave_s,
                                                             unsigned budget) {
    extern __shared__ double smem[];
    int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // (as in chain_factor_phase)
    asm volatile("" : "+v"(lane));
    const int wave = wave_s, tid = wave * 64 + lane;
    const int r16 = lane & 15, q = lane >> 4;
    const int ti = chain_tile_tab[wave][0], tj0 = chain_tile_tab[wave][1], nt = chain_tile_tab[wave][2];
    constexpr int U = 3;
    // register i of lane (r16, q) = element (row r16, column q + 4i) of a tile (the layout of the in-block update)
    const double* Ab = A + (size_t)b * BLK * ((size_t)ld + 1) + (size_t)q * ld + ti * 16 + r16;
    v4d cr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int tj = min(tj0 + u, ti);                                // (slots beyond nt shadow a valid tile and are dropped at the end)
#pragma unroll
        for (int i = 0; i < 4; ++i) cr[u][i] = ld_sc1(Ab + (size_t)(tj * 16 + 4 * i) * ld);
    }
    const unsigned long long pbase = base + 8ull * (b - 1);
    // this lane's element (row r16, k = q) of the tiles of P[b, b-1] (128 rows × 128 columns): + 16 t for row tile t, + (16 jb + 4 s) ld for k
    const double* Pl = A + (size_t)b * BLK + (size_t)(b - 1) * BLK * ld + (size_t)q * ld + r16;
    // Wave 15 (one tile) polls the strips' progress words for the whole workgroup and posts the count in LDS; the others spin on
    // that LDS word (sixteen waves polling the same lines in memory delayed each other's polls — and the strips' stores — by 2 µs).
    volatile lds_int_t* seenw = (volatile lds_int_t*)(unsigned)(unsigned long long)(smem + CHAIN_CTRL) + 1;
    auto poll_strips = [&]() {                                         // panels all eight strips of block row b have delivered
        const unsigned long long v = lane < 8 ? ld_word(sig + SIGW_PROG + SIGW_PROG_STRIDE * lane) : ~0ull;
        const unsigned long long cnt = (v > pbase) ? v - pbase : 0ull;
        int mn = cnt > 8ull ? 8 : (int)cnt;
        mn = min(mn, __shfl_xor(mn, 1));
        mn = min(mn, __shfl_xor(mn, 2));
        mn = min(mn, __shfl_xor(mn, 4));
        return __builtin_amdgcn_readfirstlane(mn);
    };
    int seen = 0, jb = 0;
    bool dead = false;
    if (wave == 15) {
        // the polling wave: posts every advance until the last panel is out (or the wait gives up: 99 releases everybody)
        if (lane == 0) CTRACE(b, 1);
        const PollTimer tm(budget);                              // (for all eight panels of the block: a whole step of the chain)
        for (int it = 0; it < POLL_CAP && seen < 8; ++it) {
            const int mn = poll_strips();
            asm volatile("" ::: "memory");
            if (mn > seen) {
                seen = mn;
                if (lane == 0) *seenw = seen;
            }
            if (tm.check(it, info)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (seen < 8) {
            if (lane == 0) {
                st_info(info, INT_MIN);
                note_giveup(4, b);
                *seenw = 99;
            }
        }
        if (lane == 0) CTRACE(b, 2);
        jb = 8;
    }
#pragma unroll 1
    while (jb < 8) {
        if (!dead && seen <= jb) {
            bool got = false;
            for (int it = 0; it < POLL_CAP_LDS; ++it) {           // (an LDS word of this workgroup: the polling wave posts 99 when it gives up)
                const int v = *seenw;
                if (v > jb) {
                    seen = v;
                    got = v <= 8;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!got) dead = true;
        }
        const bool two = (dead || seen >= jb + 2) && jb + 2 <= 8;
        const double* Pj = Pl + (size_t)(jb * 16) * ld;
        double bf[2][4], af[2][U][4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            bf[0][s4] = ld_sc1(Pj + (size_t)(4 * s4) * ld + ti * 16);
#pragma unroll
            for (int u = 0; u < U; ++u) af[0][u][s4] = ld_sc1(Pj + (size_t)(4 * s4) * ld + min(tj0 + u, ti) * 16);
        }
        if (two) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                bf[1][s4] = ld_sc1(Pj + (size_t)(16 + 4 * s4) * ld + ti * 16);
#pragma unroll
                for (int u = 0; u < U; ++u) af[1][u][s4] = ld_sc1(Pj + (size_t)(16 + 4 * s4) * ld + min(tj0 + u, ti) * 16);
            }
        }
#ifdef BOSS_CHAIN_TRACE
        if (wave == 0 && jb + (two ? 2 : 1) == 8) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CTRACE(b, 11);
        }
#endif
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int u = 0; u < U; ++u) cr[u] = mfma_f64(-af[0][u][s4], bf[0][s4], cr[u]);
        if (two) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int u = 0; u < U; ++u) cr[u] = mfma_f64(-af[1][u][s4], bf[1][s4], cr[u]);
        }
        jb += two ? 2 : 1;
        if (!dead) {
            const int v = *seenw;
            if (v > seen && v <= 8) seen = v;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (u < nt) {
            double* ct = smem + (ti * (ti + 1) / 2 + tj0 + u) * 256 + q * 16 + r16;
#pragma unroll
            for (int i = 0; i < 4; ++i) ct[64 * i] = cr[u][i];
        }
    }
    if (wave == 0) CTRACE(b, 12);
}

// C(b): the block in the LDS tile area is factored and published panel by panel.
__device__ __forceinline__ void chain_factor_phase(double* __restrict__ Ab, int ld, double* __restrict__ inv16b, int col0,
                                                             unsigned long long* __restrict__ pubword, unsigned long long seq0,
                                                             int* __restrict__ info, int wave_s) {
    extern __shared__ double smem[];
    int fail = -1;
    // the thread index rebuilt from the wave's scalar index and the lane count (nothing vector survives CHAIN_CUT_VGPRS; a reload
    // from scratch would cost a memory round trip at the start of every block); opaque to the optimiser, so the per-lane offsets
    // derived from it are recomputed per block instead of being hoisted out of the block loop and spilled
    int tid = wave_s * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(tid));
    diag_block_factor<false, false, true>(smem, Ab, ld, inv16b, col0, 8, fail, nullptr, pubword, seq0, info, tid + 0x10000);
}

// this lane's index, rebuilt from the lane count behind an optimisation barrier: whatever is derived from it is computed where it is
// used instead of being hoisted out of the block loop and kept alive (= spilled to scratch) across the phase cuts
__device__ __forceinline__ int chain_lane_opaque() {
    int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane));
    return lane;
}

__global__ __launch_bounds__(DIAG_THREADS) void potrf_chain_kernel(double* __restrict__ A, int ld, int nblk,
                                                                   double* __restrict__ inv16, int* __restrict__ info,
                                                                   unsigned long long* __restrict__ sig, unsigned long long base,
                                                                   unsigned long long cbase, unsigned budget) {
    extern __shared__ double smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    volatile lds_int_t* ctrl = (volatile lds_int_t*)(unsigned)(unsigned long long)(smem + CHAIN_CTRL);
    // The workgroup keeps its CU to itself: 16 waves × 128 registers fill every SIMD's register file, so no follower or bulk wave
    // can settle beside the pivot-chain wave (a co-resident wave on that SIMD stretches the chain 3980 -> 5450 cycles per panel).
    asm volatile("" ::: "v127");
    if (threadIdx.x == 0) __hip_atomic_fetch_add(as_global(sig + SIGW_UP), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // resident (chain_ready_kernel)
    // No per-lane value lives across an iteration of this loop or across the phase cuts inside it (the kernel ran with 21 spilled
    // registers and 16 scratch accesses per block before: the thread index and the hoisted addresses of the b = 0 copy): every use
    // of the lane index below rebuilds it (chain_lane_opaque).
    for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
        double* Ab = A + (size_t)b * BLK * ((size_t)ld + 1);
        // ---- block b with every panel < b-1 applied: the Gram matrix itself (b = 0, 1: the first follow kernel has started) or
        // tile (b, b) as the critical strips of step b-2's column update stored it (sig[SIGW_CRIT] >= cbase + 8 (b-1))
        if (wave == 0) {
            const bool ok = b < 2 ? poll_ge(sig + SIGW_WDONE, base + 1, info, budget) : poll_ge(sig + SIGW_CRIT, cbase + 8ull * (b - 1), info, budget);
            if (chain_lane_opaque() == 0) {
                *ctrl = ok ? 1 : -1;
                ctrl[1] = 0;                                  // (the follower phase's posted panel count)
            }
        }
        __syncthreads();
        if (*ctrl < 0) return;
        if (wave == 0) CTRACE(b, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // written by kernels that ran while this one was resident
        if (b == 0) {
            const int lane = chain_lane_opaque();
            for (int tl = wave; tl < DIAG_TILES; tl += DIAG_THREADS / 64) {
                int ti = 0;
                while ((ti + 1) * (ti + 2) / 2 <= tl) ++ti;
                const int tj = tl - ti * (ti + 1) / 2;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = lane + 64 * h, c = e >> 3, rp = e & 7;
                    *reinterpret_cast<v2d*>(smem + tl * 256 + c * 16 + 2 * rp) =
                        *reinterpret_cast<const v2d*>(Ab + (size_t)(tj * 16 + c) * ld + ti * 16 + 2 * rp);
                }
            }
        } else {
            CHAIN_CUT_VGPRS();
            chain_follow_phase(A, ld, b, info, sig, base, wave, budget);
        }
        CHAIN_CUT_VGPRS();
        __syncthreads();
        if (wave == 0) CTRACE(b, 3);
        chain_factor_phase(Ab, ld, inv16 + (size_t)b * (8 * 256), b * BLK, sig + SIGW_PANEL, base + 8ull * b, info, wave);
        CHAIN_CUT_VGPRS();
        if (wave == 15) CTRACE(b, 4);
        __syncthreads();                                         // (F(b + 2) writes the tile area again)
    }
}

// ------------------------------------------------------------------------------------------
// Panel solve of step k as a follower of C(k): the geometry and the arithmetic of potrf_trsm_kernel (two waves per 16-row
// strip; every tile receives its contributions in ascending column order: bit-identical results), but the operands — the
// inverse of diagonal tile jb and the row tiles of panel jb — are read (sc1) as the chain publishes them.  A wave that finds
// the whole block published when it starts fetches every operand up front like potrf_trsm_kernel; otherwise it walks the
// panels right-looking, one poll per panel.
//   follow_strip         one strip (one 128-thread workgroup's worth of work);
//   potrf_follow_kernel  step k's strips below block row k+1 (one launch per step, ordered behind the previous step's column
//                        update by its stream).  At its entry it raises sig[SIGW_WDONE] and the bulk update's gate word
//                        (everything before it on its stream is done); before it ends it waits until block row k+1's tiles are
//                        out, so that the column update behind it may read them;
//   potrf_strips_kernel  the eight strips of block row k+1 (CRIT = true), for every step, as ONE resident kernel of eight
//                        workgroups: strip s starts step k as soon as the column update of step k-1 has delivered tile (k+1, k)
//                        (its first eight strips raise sig[SIGW_CRIT]) — not when that whole kernel and the next launch are through —
//                        and publishes its tiles (sc1) and its progress for F(k+1).
// ------------------------------------------------------------------------------------------
template <bool CRIT>
__device__ __forceinline__ void follow_strip(double* __restrict__ A, int ld, int k, const double* __restrict__ inv16base,
                                             double* __restrict__ Brow, unsigned long long* __restrict__ prog,
                                             unsigned long long* __restrict__ sig, unsigned long long base, int* __restrict__ info,
                                             v4d (&xs)[TRSM_NA][64], int& ready, int& drained, int& avail, unsigned budget) {
    constexpr bool crit = CRIT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double* Lkk = A + (size_t)k * BLK * ((size_t)ld + 1);
    const double* inv16k = inv16base + (size_t)k * (8 * 256);
    const unsigned long long pb = base + 8ull * k;               // panel jb is out when sig[SIGW_PANEL] >= pb + jb + 1
    auto bload = [&](const double* p) { return crit ? ld_sc1(p) : *p; };   // (the resident strips read what a concurrent kernel stored)
    // Only wave 0 polls the chain's panel word in memory; it posts the number of panels that are out in LDS (avail) and wave 1 waits
    // on that (hundreds of follower waves polling one line in memory for a whole step slowed the bulk update they run beside).
    auto panels_out = [&](unsigned long long v) { return v > pb ? (v - pb > 8ull ? 8 : (int)(v - pb)) : 0; };
    if (threadIdx.x == 0) {
        *(volatile lds_int_t*)&ready = 0;
        *(volatile lds_int_t*)&drained = 0;
        *(volatile lds_int_t*)&avail = panels_out(ld_word(sig + SIGW_PANEL));
    }
    __syncthreads();
    auto lval = [&](int jb, int m, int s) { return ld_sc1(Lkk + (size_t)(m * 16 + 4 * s + (lane >> 4)) * ld + jb * 16 + (lane & 15)); };
    auto ival = [&](int jb, int s) { return ld_sc1(inv16k + jb * 256 + (4 * s + (lane >> 4)) * 16 + (lane & 15)); };
    bool dead = false;                                           // a wait gave up: finish without waiting (the update is marked failed)
    // CRIT: the resident strips do not wait for the chain's panel word before they fetch the inverse of diagonal tile jb — they poll
    // the DATA: inv16 is filled with an all-ones pattern before the factorisation (factor_enqueue), no computed value has it, and
    // an 8-byte word is stored whole, so a lane that sees four real values has the final ones.  One round trip instead of two on
    // the path from the chain's last panel to the next diagonal block; the row tiles of the panel (they overwrite live matrix
    // entries: no pattern possible) are still taken behind the word.
    auto ival_poll = [&](int jb, double (&iv)[4]) {
        const PollTimer tm(budget);
        for (int it = 0; it < POLL_CAP; ++it) {
            bool miss = false;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                iv[s] = ival(jb, s);
                miss |= __double_as_longlong(iv[s]) == -1ll;
            }
            if (!__any(miss)) return;
            if (dead) return;
            if (tm.check(it, info)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!dead && lane == 0) {
            st_info(info, INT_MIN);
            note_giveup(8, k);
        }
        dead = true;
        // (wave 1 waits on `avail` in LDS for the panels this wave sees: without the release it sat out its whole spin count in every
        // step of a factorisation that had already been given up — the fallback of an N = 8192 update took most of a minute)
        if (wave == 0 && lane == 0) *(volatile lds_int_t*)&avail = 99;
    };
    auto store_tile = [&](int jb, const v4d& t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double* dst = Brow + (lane & 15) + (size_t)(jb * 16 + (lane >> 4) + 4 * i) * ld;
            if (crit) st_sc1(dst, t[i]);
            else *dst = t[i];
        }
    };
    int seen = *(volatile lds_int_t*)&avail;                     // panels known to be out
    auto await = [&](int jb) {                                   // panel jb of block k is out
        if (dead || seen > jb) return;
        if (wave == 0) {
            const PollTimer tm(budget);
            for (int i = 0; i < POLL_CAP; ++i) {
                seen = panels_out(ld_word(sig + SIGW_PANEL));
                if (seen > jb) {
                    asm volatile("" ::: "memory");
                    if (lane == 0) *(volatile lds_int_t*)&avail = seen;
                    return;
                }
                if (tm.check(i, info)) break;
                __builtin_amdgcn_s_sleep(crit ? 1 : 8);
            }
            dead = true;
            if (lane == 0) {
                st_info(info, INT_MIN);
                note_giveup(crit ? 5 : 6, k);
                *(volatile lds_int_t*)&avail = 99;               // (releases wave 1)
            }
        } else {
            for (int i = 0; i < POLL_CAP_LDS; ++i) {              // (an LDS word of this workgroup: wave 0 posts 99 when it gives up)
                seen = *(volatile lds_int_t*)&avail;
                if (seen > jb) {
                    asm volatile("" ::: "memory");
                    if (seen > 8) dead = true;
                    return;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            dead = true;
        }
    };
    const bool late = seen >= 8;                                 // the whole diagonal block is out already
#ifdef BOSS_CHAIN_TRACE
    if (crit && blockIdx.x == 0 && threadIdx.x == 0) g_ctrace[(k & 63) * 16 + 7] = late ? 1 : 0;
    if (crit && blockIdx.x == 0 && threadIdx.x == 0) g_ctrace[(k & 63) * 16 + 8] = __builtin_amdgcn_s_memrealtime();
#endif
    if (wave == 0) {
        v4d acc[TRSM_NA];
#pragma unroll
        for (int jb = 0; jb < TRSM_NA; ++jb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[jb][i] = bload(Brow + (lane & 15) + (size_t)(jb * 16 + (lane >> 4) + 4 * i) * ld);
        if (late) {
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
            for (int jb = 0; jb < TRSM_NA; ++jb) store_tile(jb, acc[jb]);
            if (crit) {
                drain_stores();
                if (lane == 0) raise_word(prog, pb + TRSM_NA);
            }
        } else {
#pragma unroll
            for (int jb = 0; jb < TRSM_NA; ++jb) {
                double iv[4], lv[TRSM_NA][4];
                if (crit) {
                    ival_poll(jb, iv);
                } else {
                    await(jb);
#pragma unroll
                    for (int s = 0; s < 4; ++s) iv[s] = ival(jb, s);
#pragma unroll
                    for (int j = jb + 1; j < TRSM_NA; ++j)
#pragma unroll
                        for (int s = 0; s < 4; ++s) lv[j][s] = lval(j, jb, s);
                }
                v4d nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) nw = mfma_f64(iv[s], acc[jb][s], nw);
                acc[jb] = nw;
                xs[jb][lane] = nw;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) *(volatile lds_int_t*)&ready = jb + 1;
                store_tile(jb, nw);
                if (crit) {
                    drain_stores();
                    if (lane == 0) raise_word(prog, pb + jb + 1);
                    await(jb);                                   // the panel's row tiles
#pragma unroll
                    for (int j = jb + 1; j < TRSM_NA; ++j)
#pragma unroll
                        for (int s = 0; s < 4; ++s) lv[j][s] = lval(j, jb, s);
                }
#pragma unroll
                for (int j = jb + 1; j < TRSM_NA; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[j] = mfma_f64(-lv[j][s], nw[s], acc[j]);
            }
        }
        if (crit) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile lds_int_t*)&drained = 1;   // wave 1 may now raise the strip's word beyond TRSM_NA
        }
#pragma unroll 1
        for (int jb = TRSM_NA; jb < 8; ++jb) await(jb);          // (wave 1 learns of the remaining panels through this wave)
    } else {
        constexpr int NB = 8 - TRSM_NA;
        v4d acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = bload(Brow + (lane & 15) + (size_t)((TRSM_NA + j) * 16 + (lane >> 4) + 4 * i) * ld);
        auto hand_off = [&](int m) {
            while (*(volatile lds_int_t*)&ready <= m) __builtin_amdgcn_s_sleep(1);   // (wave 0 of this workgroup: it never waits for this wave)
            asm volatile("" ::: "memory");
            return xs[m][lane];
        };
        auto publish = [&](int upto) {                           // tiles TRSM_NA .. upto-1 of this strip are stored
            if (!crit) return;
            drain_stores();
            while (*(volatile lds_int_t*)&drained == 0) __builtin_amdgcn_s_sleep(1);
            if (lane == 0) raise_word(prog, pb + upto);
        };
        if (late) {
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
            for (int m = 0; m < TRSM_NA; ++m) {
                const v4d x = hand_off(m);
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
            for (int j = 0; j < NB; ++j) store_tile(TRSM_NA + j, acc[j]);
            publish(8);
        } else {
#pragma unroll
            for (int m = 0; m < TRSM_NA; ++m) {
                await(m);
                double lu[NB][4];
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s) lu[j][s] = lval(TRSM_NA + j, m, s);
                const v4d x = hand_off(m);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[j] = mfma_f64(-lu[j][s], x[s], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                double iv[4], lo[NB][4];
                if (crit) {
                    ival_poll(TRSM_NA + j, iv);
                } else {
                    await(TRSM_NA + j);
#pragma unroll
                    for (int s = 0; s < 4; ++s) iv[s] = ival(TRSM_NA + j, s);
#pragma unroll
                    for (int j2 = j + 1; j2 < NB; ++j2)
#pragma unroll
                        for (int s = 0; s < 4; ++s) lo[j2][s] = lval(TRSM_NA + j2, TRSM_NA + j, s);
                }
                v4d nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) nw = mfma_f64(iv[s], acc[j][s], nw);
                acc[j] = nw;
                store_tile(TRSM_NA + j, nw);
                if (crit) {
                    publish(TRSM_NA + j + 1);
                    if (j + 1 < NB) {
                        await(TRSM_NA + j);                      // the panel's row tiles (the last panel has none)
#pragma unroll
                        for (int j2 = j + 1; j2 < NB; ++j2)
#pragma unroll
                            for (int s = 0; s < 4; ++s) lo[j2][s] = lval(TRSM_NA + j2, TRSM_NA + j, s);
                    }
                }
#pragma unroll
                for (int j2 = j + 1; j2 < NB; ++j2)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[j2] = mfma_f64(-lo[j2][s], nw[s], acc[j2]);
                if (!crit) publish(TRSM_NA + j + 1);
            }
        }
    }
}

__global__ __launch_bounds__(TRSM_THREADS) void potrf_follow_kernel(double* __restrict__ A, int ld, int k,
                                                                    const double* __restrict__ inv16base, int row0, int wait_rows,
                                                                    unsigned long long* __restrict__ sig, unsigned long long base,
                                                                    unsigned long long gateval, int* __restrict__ info, unsigned budget) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        raise_word(sig + SIGW_WDONE, base + k + 1);
        if (gateval) raise_word(sig + SIGW_GATE, gateval);
    }
    __builtin_amdgcn_s_setprio(3);
    CTRACE_MIN(k, 5);
    __shared__ v4d xs[TRSM_NA][64];
    __shared__ int ready, drained, avail;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* Brow = A + (size_t)row0 + (size_t)blockIdx.x * 16 + (size_t)k * BLK * ld;
    follow_strip<false>(A, ld, k, inv16base, Brow, nullptr, sig, base, info, xs, ready, drained, avail, budget);
    if (blockIdx.x == 0 && wave == 0 && wait_rows) {
        // block row k+1 (solved by the resident strips) is out: the column update enqueued behind this kernel reads it
        const int lane = threadIdx.x & 63;
        const PollTimer tm(budget);
        for (int it = 0; it < POLL_CAP; ++it) {
            unsigned long long v = ~0ull;
            if (lane < 8) v = ld_word(sig + SIGW_PROG + SIGW_PROG_STRIDE * lane);
            if (__all(v >= base + 8ull * k + 8)) break;
            const int st = tm.check(it, info);
            if (st == 1 && lane == 0) {
                st_info(info, INT_MIN);
                note_giveup(7, k);
            }
            if (st) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    if (blockIdx.x == 0 && wave == 1) CTRACE(k, 6);
    __syncthreads();
    CTRACE_MAX(k, 9);
}

// sig[SIGW_CRIT] >= cbase + 8 k : the column update of step k-1 has stored (sc1) tiles (k+1, k) and (k+1, k+1) — eight strips
__global__ __launch_bounds__(TRSM_THREADS) void potrf_strips_kernel(double* __restrict__ A, int ld, int nblk,
                                                                    const double* __restrict__ inv16base,
                                                                    unsigned long long* __restrict__ sig, unsigned long long base,
                                                                    unsigned long long cbase, int* __restrict__ info, unsigned budget) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ v4d xs[TRSM_NA][64];
    __shared__ int ready, drained, avail;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(as_global(sig + SIGW_UP), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // resident (chain_ready_kernel)
    for (int k = 0; k + 1 < nblk; ++k) {
        // tile (k+1, k) carries every panel < k
        bool ok;
        if (k == 0) ok = poll_ge(sig + SIGW_WDONE, base + 1, info, budget);
        else ok = poll_ge(sig + SIGW_CRIT, cbase + 8ull * k, info, budget);
        (void)ok;                                                // (gave up: info is marked, carry on without waiting — every loop stays bounded)
        __syncthreads();                                         // xs / ready / drained of the previous step are no longer in use
        double* Brow = A + (size_t)(k + 1) * BLK + (size_t)blockIdx.x * 16 + (size_t)k * BLK * ld;
        follow_strip<true>(A, ld, k, inv16base, Brow, sig + SIGW_PROG + SIGW_PROG_STRIDE * blockIdx.x, sig, base, info, xs, ready, drained, avail, budget);
#ifdef BOSS_CHAIN_TRACE
        if (blockIdx.x == 0 && threadIdx.x == 64) g_ctrace[(k & 63) * 16 + 10] = __builtin_amdgcn_s_memrealtime();
#endif
    }
}

// Every workgroup of the resident kernels (chain 2, strips 8) counts itself into sig[SIGW_UP] when it starts.
// This one-lane kernel sits on the main stream between the Gram kernel and the first panel solve and ends when all of them have:
// the kernels behind it — hundreds of panel-solve workgroups per step that WAIT for the chain, two and more per CU from 64 block
// columns on — would otherwise settle on every CU before a chain workgroup (a whole CU) had been placed, and wait for it for ever
// (measured: most first updates at N = 8192 ended in the waits' time-out).  While the Gram kernel runs the resident workgroups
// normally get placed anyway: the wait then costs one short launch.
__global__ __launch_bounds__(64) void chain_ready_kernel(unsigned long long* __restrict__ sig, unsigned long long want,
                                                         int* __restrict__ info, unsigned budget) {
    if (threadIdx.x >= 64) return;
    (void)poll_ge(sig + SIGW_UP, want, info, budget);
}

// Start-up probe (ctx_init): the chain schedule needs kernels of its four streams to run at the same time.  HIP maps streams to
// a limited number of hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and kernels of two streams that share a queue run one
// after the other — the resident kernels would wait for a kernel that is queued behind them (a fifth stream did exactly that).
// One lane per stream, the main stream included: each counts itself in and waits until all have (≈20 ms at most); a lane that
// never sees the full count means its stream cannot run beside the others, and the chain schedule stays off.
__global__ __launch_bounds__(64) void chain_probe_wait_kernel(unsigned long long* __restrict__ word, unsigned long long want,
                                                              int* __restrict__ ok) {
    if (threadIdx.x != 0) return;
    __hip_atomic_fetch_add(as_global(word), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = 0; i < (1 << 15); ++i) {
        if (ld_word(word) >= want) {
            *ok = 1;
            return;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    *ok = 0;
}

}  // namespace boss

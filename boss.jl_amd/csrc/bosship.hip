// bosship.hip — C ABI (include/bosship.h) over the HIP kernels.  gfx950 only, no CPU fallback.
#include "../../include/bosship.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "gp_kernels.hpp"
#include "potrf.hpp"

using namespace boss;

#ifndef BOSS_PRED_D
#define BOSS_PRED_D 10
#endif
typedef GemmDirect<4, 1, 4, 2, BOSS_PRED_D> PredG32;   // 256-row steps × 32 candidates, ring depth BOSS_PRED_D
typedef GemmDirect<4, 1, 2, 4, 4> PredG64;   // 128 rows × 64 candidates per workgroup

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(e_ == hipErrorOutOfMemory ? BOSS_E_ALLOC : BOSS_E_NO_DEVICE,                     \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                              \
    } while (0)

// ------------------------------------------------------------------------------------------
// per-device context: stream, grow-only workspaces, per-kernel-class event profiling
// ------------------------------------------------------------------------------------------
// pinned host block per device: results of the acquisition epilogue, then two staging areas that let calls with a
// handful of candidates skip a stream synchronisation (upload) and two of three download copies
constexpr size_t PINNED_BYTES = 64 * 1024, PINNED_UP_OFF = 4096, PINNED_UP_BYTES = 32 * 1024, PINNED_DOWN_OFF = 36 * 1024,
                 PINNED_DOWN_BYTES = 28 * 1024;

// every device allocation goes through here; BOSS_POISON_ALLOC=1 (tests) fills new memory with NaN bit patterns so that
// reads of never-written memory show up instead of passing on the zeros a fresh process happens to get
static hipError_t dev_malloc(void** p, size_t bytes) {
    static const bool poison = getenv("BOSS_POISON_ALLOC") && atoi(getenv("BOSS_POISON_ALLOC"));
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && poison) {
        (void)hipMemset(*p, 0xff, bytes);                    // null stream: the library's streams do not wait for it ...
        (void)hipDeviceSynchronize();                        // ... so finish it before anything else touches the block
    }
    return e;
}

struct Workspace {
    void* p = nullptr;
    size_t bytes = 0;
};

struct Ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;          // trailing updates of the look-ahead Cholesky
    std::vector<hipEvent_t> ev_panel, ev_rest;  // per-step dependency events (no timing)
    hipEvent_t ev_fork = nullptr, ev_up = nullptr;            // ev_up: last upload out of the pinned staging area
    bool lookahead = true;
    bool prof_on = false;
    std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof;
    std::vector<hipEvent_t> ev_pool;
    Workspace vscratch, csc, pred, acq, batchA, batchX, batchMisc, craw, lgA, lgB, lgC, few;   // lg*: log-likelihood gradient (LinvT, K⁻¹, partials)   // craw: raw candidates of one-shot calls
    void* pinned = nullptr;   // small host-pinned result area
    std::mutex mtx;
};

static std::mutex g_ctx_mtx;
static std::map<int, Ctx*> g_ctx;

static int get_ctx(int device, Ctx** out) {
    std::lock_guard<std::mutex> lk(g_ctx_mtx);
    auto it = g_ctx.find(device);
    if (it != g_ctx.end()) {
        *out = it->second;
        HIPCHK(hipSetDevice(device));
        return BOSS_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(BOSS_E_NO_DEVICE, "no HIP device visible (bosship has no CPU fallback)");
    if (device < 0 || device >= n) return fail(BOSS_E_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    Ctx* c = new Ctx();
    c->device = device;
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    {
        // Panel chain on the highest-priority stream, bulk trailing updates on the lowest-priority one.
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        (void)hipStreamDestroy(c->own_stream);
        HIPCHK(hipStreamCreateWithPriority(&c->own_stream, hipStreamNonBlocking, greatest));
        c->stream = c->own_stream;
        HIPCHK(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, least));
    }
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
    c->lookahead = !(getenv("BOSS_NO_LOOKAHEAD") && atoi(getenv("BOSS_NO_LOOKAHEAD")));
    HIPCHK(hipHostMalloc(&c->pinned, PINNED_BYTES, hipHostMallocDefault));   // [0, 4 KiB) epilogue results, then staging (see temp_cand)
    // kernels that need more than 64 KiB of dynamic LDS
    HIPCHK(hipFuncSetAttribute((const void*)potrf_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)predict_kernel<PredG32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG32>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)grad_accum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_gemv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_gemv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_gemv_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)few_back_finish_kernel<PredG32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG32>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)few_finish_kernel<PredG32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG32>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)linvt_kernel<PredG32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG32>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)backsolve_kernel<PredG32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG32>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)predict_kernel<PredG64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG64>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)predict_kernel<PredG32, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG32>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)predict_kernel<PredG64, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG64>::BYTES));
    g_ctx[device] = c;
    *out = c;
    return BOSS_OK;
}

static int ws_reserve(Workspace& w, size_t bytes) {
    if (w.bytes >= bytes) return BOSS_OK;
    if (w.p) (void)hipFree(w.p);
    w.p = nullptr;
    w.bytes = 0;
    HIPCHK(dev_malloc(&w.p, bytes));
    w.bytes = bytes;
    return BOSS_OK;
}

struct ProfScope {
    Ctx* c;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char* name;
    ProfScope(Ctx* ctx, const char* n) : c(ctx), name(n) {
        if (!c->prof_on) return;
        e0 = take();
        e1 = take();
        (void)hipEventRecord(e0, c->stream);
    }
    ~ProfScope() {
        if (!c->prof_on) return;
        (void)hipEventRecord(e1, c->stream);
        c->prof[name].push_back({e0, e1});
    }
    hipEvent_t take() {
        if (!c->ev_pool.empty()) {
            hipEvent_t e = c->ev_pool.back();
            c->ev_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
};

// ------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------
struct boss_gp {
    Ctx* ctx = nullptr;
    int kernel = 0, d = 0, N = 0, Np = 0, nblk = 0, ld = 0;
    double *Xraw = nullptr, *Xsc = nullptr, *y = nullptr, *mean = nullptr, *A = nullptr;
    double *LT = nullptr, *DT2 = nullptr, *avec = nullptr;   // gradients: transposed factor, transposed 256×256 inverses, a = L⁻ᵀz (lazily)
    bool have_lt = false;
    double *inv16 = nullptr, *Dinv = nullptr, *Dinv2 = nullptr, *hyp = nullptr, *invlam = nullptr, *scal = nullptr;
    int* info = nullptr;
    unsigned char* discrete_dev = nullptr;     // d flags (device) or null
    std::vector<unsigned char> discrete;
    double amp2 = 0.0;
    bool has_mean = false, fitted = false, pending = false, have_dinv = false;
    double* host_res = nullptr;                // pinned: scal[2], info
    double* host_par = nullptr;                // pinned staging: invlam[d], hyp[2]
    hipEvent_t par_ev = nullptr;               // recorded after the staging copies were enqueued
    unsigned long long epoch = 0;              // bumped by every boss_gp_update: tracked candidate states go stale
    hipEvent_t dinv_ev = nullptr;              // the side stream finished building Dinv / Dinv2
    bool dinv_pending = false;
    // gradient-observation posterior (GradientGaussianProcess): npts points, N = npts (1 + d) observations,
    // Xraw is [d][ldx]; hyp = {α², σ², σ_∂²}
    bool aug = false;
    int npts = 0, ldx = 0;
    // nonstationary posterior (NonstationaryGP, Gibbs kernel): per-point λ [d][Np], α [Np], σ [Np]; Xraw holds the
    // (rounded where discrete) training points
    bool gibbs = false;
    double *lamX = nullptr, *ampX = nullptr, *noiseX = nullptr;
    // explicit L⁻ᵀ for calls with one to four candidates (built on the second such call on a factorisation)
    double *Winv = nullptr, *Linv = nullptr;   // L⁻ᵀ (upper) and L⁻¹ (lower)
    bool have_winv = false;
    int few_calls = 0;
    int append_calls = 0;                      // single-observation appends since the last update (the second one builds the inverses)
};

struct boss_cand {
    Ctx* ctx = nullptr;
    int d = 0, M = 0, Mp = 0;
    double* Craw = nullptr;                    // [d][Mp]
};

struct boss_track {                            // resident predictive state of (posterior, candidate set)
    Ctx* ctx = nullptr;
    boss_gp* gp = nullptr;
    int d = 0, M = 0, Mp = 0, tiles = 0, Ncap = 0, N = 0;
    unsigned long long epoch = 0;
    double *V = nullptr, *Csc = nullptr, *mu = nullptr, *var = nullptr, *mean = nullptr;
    bool has_mean = false;
};

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static void linv_enqueue(boss_gp* g, hipStream_t s, double* U, double* Lw);
// largest system the entry points accept: element offsets into the factor stay below 2^31 (exercised up to
// 36 864 rows = 10.9 GB by the tests)
constexpr int MAX_ROWS = 46080;

// ------------------------------------------------------------------------------------------
// library
// ------------------------------------------------------------------------------------------
extern "C" const char* boss_version(void) { return "bosship 0.1.0 (gfx950, fp64 MFMA)"; }
extern "C" const char* boss_last_error(void) { return g_last_error.c_str(); }

extern "C" int boss_device_count(int* n_out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        if (n_out) *n_out = 0;
        return fail(BOSS_E_NO_DEVICE, "no HIP device visible (bosship has no CPU fallback)");
    }
    if (n_out) *n_out = n;
    return BOSS_OK;
}

extern "C" int boss_set_stream(int device, void* hip_stream) {
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return BOSS_OK;
}

extern "C" int boss_device_sync(int device) {
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return BOSS_OK;
}

// ------------------------------------------------------------------------------------------
// factorisation driver (shared by the single handle and the batched log-likelihood)
// ------------------------------------------------------------------------------------------
static void potrf_enqueue(Ctx* c, double* A, int ld, int Np, int batch, size_t bstride, double* inv16,
                          size_t inv16_bstride, int* info) {
    const int nblk = Np / BLK;
    hipStream_t s = c->stream;
    static const int pair_min_batch = getenv("BOSS_PAIR_MIN_BATCH") ? atoi(getenv("BOSS_PAIR_MIN_BATCH")) : 4;
    if (batch >= pair_min_batch && nblk >= 3) {
        // Batched factorisations have parallelism to spare and are bound by the HBM traffic of the
        // trailing updates (every step reads and writes the whole trailing matrix).  Pair the panels:
        //   diag_k, solve_k, column k+1 <- panel k, diag_{k+1}, solve_{k+1}, trailing(>= k+2) <- panels k,k+1 (K = 256)
        // so each trailing tile moves through HBM half as often.  One stream, no events.
        for (int k = 0; k < nblk; k += 2) {
            for (int kk = k; kk < k + 2 && kk < nblk; ++kk) {
                {
                    ProfScope ps(c, "potrf_diag");
                    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, 1, batch), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, A, ld, bstride,
                                       kk, inv16, inv16_bstride, info);
                }
                const int nrows16 = (Np - (kk + 1) * BLK) / 16 + 1;
                {
                    ProfScope ps(c, "potrf_trsm");
                    hipLaunchKernelGGL(potrf_trsm_kernel, dim3(nrows16, 1, batch), dim3(64), 0, s, A, ld, bstride, kk, inv16,
                                       inv16_bstride, (kk + 1) * BLK);
                }
                const int m = nblk - 1 - kk;
                if (kk == k && m > 0) {
                    ProfScope ps(c, "potrf_syrk");
                    hipLaunchKernelGGL(potrf_colupd_kernel, dim3(4 * m + 1, 1, batch), dim3(256), 0, s, A, ld, bstride, kk, m, 1, 0, 1);
                }
            }
            const int m2 = nblk - (k + 2);                   // block triangle behind the pair
            if (m2 > 0) {
                ProfScope ps(c, "potrf_syrk");
                hipLaunchKernelGGL(potrf_syrk_kernel<2>, dim3((m2 * (m2 + 1) / 2 + m2) * batch, 1, 1), dim3(256), 0, s, A, ld, bstride,
                                   k, k + 2, m2, batch);
            } else if (m2 == 0 && k + 1 < nblk) {
                // the pair ends the matrix: only the δ^T rows behind it are left (their solve rides in solve_{k+1},
                // but panel k's contribution to the δ^T entries of block column k+1 was applied by colupd) — nothing to do
            }
        }
        return;
    }
    // Look-ahead needs no profiling scopes (they would serialise the two streams) and >= 3 blocks.
    const bool la = c->lookahead && !c->prof_on && nblk >= 3;
    if (la) {
        while ((int)c->ev_panel.size() < nblk) {
            hipEvent_t e0, e1;
            (void)hipEventCreateWithFlags(&e0, hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&e1, hipEventDisableTiming);
            c->ev_panel.push_back(e0);
            c->ev_rest.push_back(e1);
        }
        (void)hipEventRecord(c->ev_fork, s);                 // side stream starts after everything enqueued so far
        (void)hipStreamWaitEvent(c->side_stream, c->ev_fork, 0);
    }
    static const int small_m = getenv("BOSS_SMALL_M") ? atoi(getenv("BOSS_SMALL_M")) : 12;
    static const bool no_pairs = getenv("BOSS_NO_PAIRS") && atoi(getenv("BOSS_NO_PAIRS"));
    int kstart = 0, tail_join = -1;
    if (la && !no_pairs && batch == 1 && nblk - 1 > small_m + 2) {
        // ---- paired look-ahead: the bulk update applies TWO panels at a time (K = 256: half the trailing-matrix
        // traffic, better tile efficiency, half the events) and has TWO steps of slack:
        //   even step e:  diag, solve,  column e+1 <- panel e                                   (K = 128)
        //   odd  step o:  diag, solve,  [bulk(o-2) done]  columns o+1, o+2 <- panels o-1, o     (K = 256)
        //                 side stream:  bulk(o): columns >= o+3 <- panels o-1, o                 (K = 256)
        // Every column still receives every earlier panel exactly once; the tail (few block columns left)
        // falls through to the single-stream steps below, entered at an even step.
        int last = -1;
        int k = 0;
        for (; k + 1 < nblk; k += 2) {
            if (nblk - 1 - k <= small_m + 1) break;          // tail: switch at an even step
            for (int kk = k; kk < k + 2; ++kk) {
                hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, 1, 1), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, A, ld, bstride, kk, inv16,
                                   inv16_bstride, info);
                const int nrows16 = (Np - (kk + 1) * BLK) / 16 + 1;
                hipLaunchKernelGGL(potrf_trsm_kernel, dim3(nrows16, 1, 1), dim3(64), 0, s, A, ld, bstride, kk, inv16, inv16_bstride,
                                   (kk + 1) * BLK);
                const int m = nblk - 1 - kk;
                if (kk == k) {
                    hipLaunchKernelGGL(potrf_colupd_kernel, dim3(4 * m + 1, 1, 1), dim3(256), 0, s, A, ld, bstride, kk, m, 1, 0, 1);
                } else {
                    if (last >= 0) (void)hipStreamWaitEvent(s, c->ev_rest[last], 0);
                    const int nc = m >= 2 ? 2 : 1;
                    hipLaunchKernelGGL(potrf_colupd_kernel, dim3((4 * m + 1) + (nc == 2 ? 4 * (m - 1) + 1 : 0), 1, 1), dim3(256), 0, s,
                                       A, ld, bstride, kk, m, nc, 0, 2);
                    const int m3 = m - 2;                     // block triangle beyond the two columns just updated
                    if (m3 > 0) {
                        (void)hipEventRecord(c->ev_panel[kk], s);
                        (void)hipStreamWaitEvent(c->side_stream, c->ev_panel[kk], 0);
                        hipLaunchKernelGGL(potrf_syrk_kernel<2>, dim3(m3 * (m3 + 1) / 2 + m3, 1, 1), dim3(256), 0, c->side_stream, A, ld,
                                           bstride, kk - 1, kk + 3, m3, 0);
                        (void)hipEventRecord(c->ev_rest[kk], c->side_stream);
                        last = kk;
                    }
                }
            }
        }
        // the last bulk update still runs on the side stream; the tail's first diagonal block and panel solve touch
        // only block column k (brought up to date by the column updates above), so the join waits until the tail's
        // first trailing update
        tail_join = last;
        kstart = k;
    }
    int last_rest = tail_join;
    for (int k = kstart; k < nblk; ++k) {
        {
            ProfScope ps(c, "potrf_diag");
            hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, 1, batch), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, A, ld, bstride,
                               k, inv16, inv16_bstride, info);
        }
        const int nrows16 = (Np - (k + 1) * BLK) / 16 + 1;   // rows below + one 16-row group of the RHS block
        {
            ProfScope ps(c, "potrf_trsm");
            hipLaunchKernelGGL(potrf_trsm_kernel, dim3(nrows16, 1, batch), dim3(64), 0, s, A, ld, bstride, k, inv16,
                               inv16_bstride, (k + 1) * BLK);
        }
        const int m = nblk - 1 - k;
        if (m == 0) break;
        if (la && m <= small_m) {
            // few tiles left: one small single-stream launch beats the two-stream choreography
            if (last_rest >= 0) (void)hipStreamWaitEvent(s, c->ev_rest[last_rest], 0);
            last_rest = -1;
            hipLaunchKernelGGL(potrf_colupd_kernel, dim3(2 * m * (m + 1) + m, 1, batch), dim3(256), 0, s, A, ld, bstride, k,
                               m, m, 0, 1);
            continue;
        }
        if (!la) {
            ProfScope ps(c, "potrf_syrk");
            hipLaunchKernelGGL(potrf_syrk_kernel<1>, dim3(m * (m + 1) / 2 + m, 1, batch), dim3(256), 0, s, A, ld, bstride, k,
                               k + 1, m, 0);
            continue;
        }
        // ---- look-ahead: panel chain on `s`, bulk of the trailing update on the side stream ----------
        // the next panel's block column was last written by the side stream's update of step k-1
        if (last_rest >= 0) (void)hipStreamWaitEvent(s, c->ev_rest[last_rest], 0);
        hipLaunchKernelGGL(potrf_colupd_kernel, dim3(4 * m + 1, 1, batch), dim3(256), 0, s, A, ld, bstride, k, m, 1, 0, 1);
        last_rest = -1;
        static const int exp_norest = getenv("BOSS_EXP_NOREST") ? atoi(getenv("BOSS_EXP_NOREST")) : 0;   // timing experiments only
        if (m >= 2 && exp_norest != 1) {
            // the bulk update is released only AFTER the column update has been dispatched, so the two
            // do not fight for CUs; it then overlaps the next diagonal block + panel solve
            const int m2 = m - 1;                             // block triangle beyond the next panel
            (void)hipEventRecord(c->ev_panel[k], s);
            (void)hipStreamWaitEvent(c->side_stream, c->ev_panel[k], 0);
            if (exp_norest != 2)
                hipLaunchKernelGGL(potrf_syrk_kernel<1>, dim3(m2 * (m2 + 1) / 2 + m2, 1, batch), dim3(256), 0, c->side_stream, A,
                                   ld, bstride, k, k + 2, m2, 0);
            (void)hipEventRecord(c->ev_rest[k], c->side_stream);
            last_rest = k;
        }
    }
    if (la && last_rest >= 0) (void)hipStreamWaitEvent(s, c->ev_rest[last_rest], 0);   // join
}

static void gram_enqueue(Ctx* c, const double* Xsc, size_t xs_bstride, int d, int N, int Np, int kern,
                         const double* hyp, double* A, int ld, size_t bstride, int batch) {
    ProfScope ps(c, "gram");
    const int nt = Np / 64;
    hipLaunchKernelGGL(gram_kernel, dim3(nt * (nt + 1) / 2, 1, batch), dim3(256), 0, c->stream, Xsc, xs_bstride, d, N,
                       Np, kern, hyp, A, ld, bstride, 0);
}

// ------------------------------------------------------------------------------------------
// posterior handle
// ------------------------------------------------------------------------------------------
static void gp_release(boss_gp* g) {
    if (!g) return;
    if (g->ctx) (void)hipSetDevice(g->ctx->device);
    void* ptrs[] = {g->Xraw, g->Xsc, g->y, g->mean, g->A, g->inv16, g->Dinv, g->Dinv2, g->LT, g->DT2, g->avec, g->hyp, g->invlam, g->scal, g->info, g->discrete_dev};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (g->host_res) (void)hipHostFree(g->host_res);
    if (g->Winv) (void)hipFree(g->Winv);
    if (g->Linv) (void)hipFree(g->Linv);
    if (g->lamX) (void)hipFree(g->lamX);
    if (g->ampX) (void)hipFree(g->ampX);
    if (g->noiseX) (void)hipFree(g->noiseX);
    if (g->host_par) (void)hipHostFree(g->host_par);
    if (g->par_ev) (void)hipEventDestroy(g->par_ev);
    if (g->dinv_ev) (void)hipEventDestroy(g->dinv_ev);
    delete g;
}

static void pack_points(std::vector<double>& dst, const double* X, int d, int n, int ldp,
                        const unsigned char* discrete) {
    // X is d×n column-major (point-contiguous); dst is [d][ldp] (dimension-major), padding = 0.
    dst.assign((size_t)d * ldp, 0.0);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < d; ++k) {
            double v = X[(size_t)j * d + k];
            if (discrete && discrete[k]) v = std::nearbyint(v);   // Julia round(): half-to-even (kernels.jl:56-59)
            dst[(size_t)k * ldp + j] = v;
        }
}

// npts points of dimension d carrying N observations (N = npts for the plain model, npts (1 + d) with
// gradient observations); y holds the N observed values in the posterior's own ordering.
static int gp_create_common(int device, int kernel, int d, int npts, int N, const double* X, const double* y,
                            const unsigned char* discrete, bool aug, boss_gp_t** out) {
    if (!out) return fail(BOSS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (kernel < 0 || kernel > 2) return fail(BOSS_E_INVALID, "unknown kernel id");
    if (d < 1 || N < 1 || !X || !y) return fail(BOSS_E_INVALID, "need d >= 1, N >= 1 and non-NULL X, y");
    if (N > MAX_ROWS) return fail(BOSS_E_INVALID, "more than 46080 rows (observations, or n (1 + d) with gradients) are not supported");
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    boss_gp* g = new boss_gp();
    g->ctx = c;
    g->kernel = kernel;
    g->d = d;
    g->N = N;
    g->Np = round_up(N, PRED_RB);                 // whole 256-row prediction steps (padding = identity)
    g->nblk = g->Np / BLK;
    g->ld = g->Np + RHS_ROWS;
    g->aug = aug;
    g->npts = npts;
    g->ldx = aug ? round_up(npts, 64) : g->Np;
    const size_t Np = g->Np, ldx = g->ldx;
#define GALLOC(ptr, bytes)                                        \
    do {                                                          \
        hipError_t e_ = dev_malloc((void**)&(ptr), (bytes));       \
        if (e_ != hipSuccess) {                                   \
            gp_release(g);                                        \
            return fail(BOSS_E_ALLOC, "device allocation failed"); \
        }                                                         \
    } while (0)
    GALLOC(g->Xraw, sizeof(double) * d * ldx);
    GALLOC(g->Xsc, sizeof(double) * d * ldx);
    GALLOC(g->y, sizeof(double) * Np);
    GALLOC(g->mean, sizeof(double) * Np);
    GALLOC(g->A, sizeof(double) * (size_t)g->ld * Np);
    GALLOC(g->inv16, sizeof(double) * g->nblk * 8 * 256);
    GALLOC(g->Dinv, sizeof(double) * g->nblk * BLK * BLK);
    GALLOC(g->Dinv2, sizeof(double) * (g->Np / PRED_RB) * PRED_RB * PRED_RB);
    GALLOC(g->hyp, sizeof(double) * 4);
    GALLOC(g->invlam, sizeof(double) * d);
    GALLOC(g->scal, sizeof(double) * 2);
    GALLOC(g->info, sizeof(int));
    if (hipHostMalloc((void**)&g->host_res, 64, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&g->host_par, sizeof(double) * (d + 4), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&g->par_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->dinv_ev, hipEventDisableTiming) != hipSuccess) {
        gp_release(g);
        return fail(BOSS_E_ALLOC, "pinned allocation failed");
    }
    if (discrete) {
        g->discrete.assign(discrete, discrete + d);
        bool any = false;
        for (int k = 0; k < d; ++k) any |= discrete[k] != 0;
        if (any) {
            GALLOC(g->discrete_dev, d);
            (void)hipMemcpy(g->discrete_dev, discrete, d, hipMemcpyHostToDevice);
        } else {
            g->discrete.clear();
        }
    }
#undef GALLOC
    std::vector<double> buf;
    pack_points(buf, X, d, npts, (int)ldx, g->discrete.empty() ? nullptr : g->discrete.data());
    HIPCHK(hipMemcpyAsync(g->Xraw, buf.data(), sizeof(double) * d * ldx, hipMemcpyHostToDevice, c->stream));
    std::vector<double> yb(Np, 0.0);
    std::copy(y, y + N, yb.begin());
    HIPCHK(hipMemcpyAsync(g->y, yb.data(), sizeof(double) * Np, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(g->mean, 0, sizeof(double) * Np, c->stream));
    HIPCHK(hipMemsetAsync(g->A, 0, sizeof(double) * (size_t)g->ld * Np, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // host staging buffers go out of scope
    *out = g;
    return BOSS_OK;
}

extern "C" int boss_gp_create(int device, int kernel, int d, int N, const double* X, const double* y,
                              const unsigned char* discrete, boss_gp_t** out) {
    return gp_create_common(device, kernel, d, N, N, X, y, discrete, false, out);
}

// GradientGaussianProcess (src/models/gradient_gp.jl): dY is d×n column-major, dY[l + d*j] = ∂y/∂x_l at x_j.
extern "C" int boss_ggp_create(int device, int kernel, int d, int n, const double* X, const double* y, const double* dY,
                               boss_gp_t** out) {
    if (!out) return fail(BOSS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (d < 1 || n < 1 || !X || !y || !dY) return fail(BOSS_E_INVALID, "need d >= 1, n >= 1 and non-NULL X, y, dY");
    if (d > AUG_MAX_D) return fail(BOSS_E_INVALID, "gradient observations: x_dim above 16 is not supported");
    if ((long long)n * (1 + d) > MAX_ROWS) return fail(BOSS_E_INVALID, "augmented system too large (n (1 + d) > 46080)");
    // `_build_obs_vector` (gradient_gp.jl:288-302): [y_1..n, ∂y/∂x_1 (1..n), …, ∂y/∂x_d (1..n)]
    std::vector<double> yt((size_t)n * (1 + d));
    for (int j = 0; j < n; ++j) {
        yt[j] = y[j];
        for (int l = 0; l < d; ++l) yt[(size_t)n * (1 + l) + j] = dY[(size_t)j * d + l];
    }
    return gp_create_common(device, kernel, d, n, n * (1 + d), X, yt.data(), nullptr, true, out);
}

#define NOT_FOR_AUG(g)                                                                                            \
    do {                                                                                                          \
        if ((g)->aug) return fail(BOSS_E_INVALID, "not available for gradient-observation posteriors (boss_ggp_*)"); \
        if ((g)->gibbs) return fail(BOSS_E_INVALID, "not available for nonstationary posteriors (boss_ngp_*)");      \
    } while (0)

// NonstationaryGP (src/models/nonstationary_gp/nonstationary_gp.jl): the data of one output slice; the
// per-point hyper-parameters arrive with boss_ngp_update.
extern "C" int boss_ngp_create(int device, int d, int N, const double* X, const double* y, const unsigned char* discrete,
                               boss_gp_t** out) {
    int rc = gp_create_common(device, KERN_SQEXP, d, N, N, X, y, discrete, false, out);
    if (rc) return rc;
    boss_gp* g = *out;
    g->gibbs = true;
    g->kernel = KERN_GIBBS;
    const size_t Np = g->Np;
    if (dev_malloc((void**)&g->lamX, sizeof(double) * d * Np) != hipSuccess ||
        dev_malloc((void**)&g->ampX, sizeof(double) * Np) != hipSuccess ||
        dev_malloc((void**)&g->noiseX, sizeof(double) * Np) != hipSuccess) {
        gp_release(g);
        *out = nullptr;
        return fail(BOSS_E_ALLOC, "device allocation failed");
    }
    return BOSS_OK;
}

extern "C" int boss_gp_set_y(boss_gp_t* g, const double* y) {
    if (!g || !y) return fail(BOSS_E_INVALID, "NULL argument");
    if (g->aug) return fail(BOSS_E_INVALID, "not available for gradient-observation posteriors (boss_ggp_*)");
    HIPCHK(hipSetDevice(g->ctx->device));
    HIPCHK(hipMemcpyAsync(g->y, y, sizeof(double) * g->N, hipMemcpyHostToDevice, g->ctx->stream));
    HIPCHK(hipStreamSynchronize(g->ctx->stream));
    g->fitted = false;
    return BOSS_OK;
}

static int validate_hyper(int d, const double* lam, double amp, double sig) {
    // gaussian_process.jl:227-229: negative values signal an error (zero is lifted to 1e-8)
    if (!lam) return fail(BOSS_E_INVALID, "lengthscale is NULL");
    for (int k = 0; k < d; ++k)
        if (!(lam[k] >= 0.0)) return fail(BOSS_E_INVALID, "lengthscales must be >= 0");
    if (!(amp >= 0.0)) return fail(BOSS_E_INVALID, "amplitude must be >= 0");
    if (!(sig >= 0.0)) return fail(BOSS_E_INVALID, "noise_std must be >= 0");
    return BOSS_OK;
}

static int gp_finish(boss_gp* g, double* logpdf_out) {
    Ctx* c = g->ctx;
    HIPCHK(hipStreamSynchronize(c->stream));
    g->pending = false;
    const double logdet = g->host_res[0], zz = g->host_res[1];
    int info;
    std::memcpy(&info, &g->host_res[2], sizeof(int));
    if (info != 0 || !std::isfinite(logdet) || !std::isfinite(zz)) {
        g->fitted = false;
        if (logpdf_out) *logpdf_out = -std::numeric_limits<double>::infinity();
        char msg[160];
        std::snprintf(msg, sizeof msg, "matrix is not positive definite (pivot %d failed) — PosDefException", info);
        return fail(BOSS_E_NOT_PD, msg);
    }
    g->fitted = true;
    if (logpdf_out) *logpdf_out = -0.5 * (g->N * 1.8378770664093453 + logdet + zz);
    return BOSS_OK;
}

// Dense inverses of the diagonal blocks (prediction operands): 128×128 (Dinv) and 256×256 (Dinv2,
// lower-left quadrant = −C⁻¹ (B A⁻¹)).  Built right after a factorisation on the side stream, so it
// overlaps the log-det reduction, the result copies and the host's return to the caller; the first
// prediction waits on dinv_ev.  (While profiling it runs lazily on the main stream instead.)
static void dinv_launch(boss_gp* g, hipStream_t s) {
    hipLaunchKernelGGL(potrf_dinv_kernel, dim3(8, g->nblk, 1), dim3(64), 0, s, g->A, g->ld, (size_t)0, g->inv16, (size_t)0,
                       g->Dinv, (size_t)0);
    const int npair = g->Np / PRED_RB;
    const size_t s2 = (size_t)PRED_RB * PRED_RB, s1 = (size_t)2 * BLK * BLK;
    double* T1 = g->Dinv2 + (size_t)BLK * PRED_RB;                         // upper-right quadrant as scratch
    hipLaunchKernelGGL(small_gemm128_kernel, dim3(16, npair), dim3(256), 0, s, g->A + BLK, g->ld,
                       (size_t)PRED_RB * ((size_t)g->ld + 1), g->Dinv, BLK, s1, T1, PRED_RB, s2, 1.0);
    hipLaunchKernelGGL(small_gemm128_kernel, dim3(16, npair), dim3(256), 0, s, g->Dinv + (size_t)BLK * BLK, BLK, s1,
                       (const double*)T1, PRED_RB, s2, g->Dinv2 + BLK, PRED_RB, s2, -1.0);
    hipLaunchKernelGGL(dinv_pair_assemble_kernel, dim3(PRED_RB, npair), dim3(256), 0, s, (const double*)g->Dinv, g->Dinv2);
}

static void dinv_eager(boss_gp* g) {
    Ctx* c = g->ctx;
    if (c->prof_on) return;                                  // lazily, inside its own profiling scope
    (void)hipEventRecord(c->ev_fork, c->stream);
    (void)hipStreamWaitEvent(c->side_stream, c->ev_fork, 0);
    dinv_launch(g, c->side_stream);
    (void)hipEventRecord(g->dinv_ev, c->side_stream);
    g->dinv_pending = true;
    g->have_dinv = true;
}

// the side stream may still be reading A / inv16 for the previous factorisation's inverses
static void dinv_join(boss_gp* g) {
    if (g->dinv_pending) {
        (void)hipStreamWaitEvent(g->ctx->stream, g->dinv_ev, 0);
        g->dinv_pending = false;
    }
}

// Gram + blocked Cholesky + solves + logdet of the handle's resident data under its resident
// hyper-parameters; results land in host_res when the stream drains.
static int factor_enqueue(boss_gp* g) {
    Ctx* c = g->ctx;
    hipStream_t s = c->stream;
    dinv_join(g);
    HIPCHK(hipMemsetAsync(g->info, 0, sizeof(int), s));
    {
        ProfScope ps(c, "prep");
        if (!g->aug && !g->gibbs)
            hipLaunchKernelGGL(scale_points_kernel, dim3((g->Np + 255) / 256, 1, 1), dim3(256), 0, s, g->Xraw, g->Xsc,
                               (size_t)0, g->invlam, g->d, g->Np);
        hipLaunchKernelGGL(rhs_rows_kernel, dim3((g->Np + 255) / 256, 1, 1), dim3(256), 0, s, g->A, g->ld, (size_t)0,
                           g->N, g->Np, g->y, g->mean, (size_t)0, 0);
    }
    if (g->aug) {
        ProfScope ps(c, "gram");
        const long long t64 = g->Np / 64;
        hipLaunchKernelGGL(aug_gram_kernel, dim3((unsigned)(t64 * (t64 + 1) / 2)), dim3(256), 0, s, (const double*)g->Xraw, g->ldx,
                           g->d, g->npts, g->N, g->Np, g->kernel, (const double*)g->hyp, (const double*)g->invlam, g->A, g->ld);
    } else if (g->gibbs) {
        ProfScope ps(c, "gram");
        const int t64 = g->Np / 64;
        hipLaunchKernelGGL(gibbs_gram_kernel, dim3(t64 * (t64 + 1) / 2), dim3(256), 0, s, (const double*)g->Xraw,
                           (const double*)g->lamX, (const double*)g->ampX, (const double*)g->noiseX, g->d, g->N, g->Np, g->A, g->ld);
    } else {
        gram_enqueue(c, g->Xsc, 0, g->d, g->N, g->Np, g->kernel, g->hyp, g->A, g->ld, 0, 1);
    }
    potrf_enqueue(c, g->A, g->ld, g->Np, 1, 0, g->inv16, 0, g->info);
    dinv_eager(g);
    {
        ProfScope ps(c, "logdet");
        hipLaunchKernelGGL(potrf_logdet_kernel, dim3(1, 1, 1), dim3(256), 0, s, g->A, g->ld, (size_t)0, g->N, g->Np,
                           g->scal);
    }
    HIPCHK(hipMemcpyAsync(g->host_res, g->scal, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&g->host_res[2], g->info, sizeof(int), hipMemcpyDeviceToHost, s));
    return BOSS_OK;
}

extern "C" int boss_gp_update(boss_gp_t* g, const double* lengthscale, double amplitude, double noise_std,
                              const double* mean_X, int flags, double* logpdf_out) {
    if (!g) return fail(BOSS_E_INVALID, "gp is NULL");
    NOT_FOR_AUG(g);
    int rc = validate_hyper(g->d, lengthscale, amplitude, noise_std);
    if (rc) return rc;
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);                 // streams, events and scratch of the device context are shared
    hipStream_t s = c->stream;
    g->fitted = false;
    g->have_dinv = false;
    g->have_winv = false;
    g->few_calls = 0;
    g->have_lt = false;
    ++g->epoch;
    g->append_calls = 0;
    // +1e-8 on every parameter (gaussian_process.jl:239-241)
    HIPCHK(hipEventSynchronize(g->par_ev));   // previous update's staging copies have been consumed
    double* invlam = g->host_par;
    double* hyp = g->host_par + g->d;
    for (int k = 0; k < g->d; ++k) invlam[k] = 1.0 / (lengthscale[k] + MIN_PARAM_VALUE);
    const double amp = amplitude + MIN_PARAM_VALUE, sig = noise_std + MIN_PARAM_VALUE;
    hyp[0] = amp * amp;
    hyp[1] = sig * sig;
    g->amp2 = hyp[0];
    HIPCHK(hipMemcpyAsync(g->invlam, invlam, sizeof(double) * g->d, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g->hyp, hyp, 2 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(g->par_ev, s));
    if (mean_X) {
        HIPCHK(hipMemcpyAsync(g->mean, mean_X, sizeof(double) * g->N, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));      // caller's (pageable) mean buffer may be reused on return
        g->has_mean = true;
    } else if (g->has_mean) {
        HIPCHK(hipMemsetAsync(g->mean, 0, sizeof(double) * g->Np, s));
        g->has_mean = false;
    }
    rc = factor_enqueue(g);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    g->pending = true;
    if (flags & BOSS_FIT_NO_SYNC) return BOSS_OK;
    return gp_finish(g, logpdf_out);
}

// model_posterior_slice / data_loglike of GradientGaussianProcess (gradient_gp.jl:307-329, :367-397):
// augmented Gram, Cholesky, α-solve and log marginal likelihood under (λ, α, σ, σ_∂).
extern "C" int boss_ggp_update(boss_gp_t* g, const double* lengthscale, double amplitude, double noise_std,
                               double grad_noise_std, int flags, double* logpdf_out) {
    if (!g) return fail(BOSS_E_INVALID, "gp is NULL");
    if (!g->aug) return fail(BOSS_E_INVALID, "handle was not created by boss_ggp_create");
    int rc = validate_hyper(g->d, lengthscale, amplitude, noise_std);
    if (rc) return rc;
    if (!(grad_noise_std >= 0.0)) return fail(BOSS_E_INVALID, "grad_noise_std must be >= 0");
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    hipStream_t s = c->stream;
    g->fitted = false;
    g->have_dinv = false;
    g->have_winv = false;
    g->few_calls = 0;
    ++g->epoch;
    g->append_calls = 0;
    HIPCHK(hipEventSynchronize(g->par_ev));
    double* invlam = g->host_par;
    double* hyp = g->host_par + g->d;
    for (int k = 0; k < g->d; ++k) invlam[k] = 1.0 / (lengthscale[k] + MIN_PARAM_VALUE);   // gradient_gp.jl:128-131
    const double amp = amplitude + MIN_PARAM_VALUE, sig = noise_std + MIN_PARAM_VALUE, sgd = grad_noise_std + MIN_PARAM_VALUE;
    hyp[0] = amp * amp;
    hyp[1] = sig * sig;                                      // :200-204
    hyp[2] = sgd * sgd;
    g->amp2 = hyp[0];
    HIPCHK(hipMemcpyAsync(g->invlam, invlam, sizeof(double) * g->d, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g->hyp, hyp, 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(g->par_ev, s));
    rc = factor_enqueue(g);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    g->pending = true;
    if (flags & BOSS_FIT_NO_SYNC) return BOSS_OK;
    return gp_finish(g, logpdf_out);
}

// finite_nongp + logpdf / posterior (nonstationary_gp.jl:153-196, :237-245): the caller evaluates its latent
// models at the training points — lam_X d×N (column j = λ(x_j)), amp_X N, noise_X N.
extern "C" int boss_ngp_update(boss_gp_t* g, const double* lam_X, const double* amp_X, const double* noise_X,
                               const double* mean_X, int flags, double* logpdf_out) {
    if (!g || !lam_X || !amp_X || !noise_X) return fail(BOSS_E_INVALID, "NULL argument");
    if (!g->gibbs) return fail(BOSS_E_INVALID, "handle was not created by boss_ngp_create");
    const int d = g->d, N = g->N, Np = g->Np;
    std::vector<double> lam((size_t)d * Np, 1.0), amp(Np, 0.0), noi(Np, 0.0);
    for (int j = 0; j < N; ++j) {
        for (int k = 0; k < d; ++k) {
            const double v = lam_X[(size_t)j * d + k];
            if (!(v > 0.0) || !std::isfinite(v)) return fail(BOSS_E_INVALID, "lengthscales must be finite and > 0");
            lam[(size_t)k * Np + j] = v;
        }
        if (!(amp_X[j] >= 0.0) || !std::isfinite(amp_X[j])) return fail(BOSS_E_INVALID, "amplitudes must be finite and >= 0");
        if (!(noise_X[j] >= 0.0) || !std::isfinite(noise_X[j])) return fail(BOSS_E_INVALID, "noise stds must be finite and >= 0");
        amp[j] = amp_X[j];
        noi[j] = noise_X[j];
    }
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    hipStream_t s = c->stream;
    g->fitted = false;
    g->have_dinv = false;
    g->have_winv = false;
    g->few_calls = 0;
    ++g->epoch;
    g->append_calls = 0;
    HIPCHK(hipMemcpyAsync(g->lamX, lam.data(), sizeof(double) * d * Np, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g->ampX, amp.data(), sizeof(double) * Np, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g->noiseX, noi.data(), sizeof(double) * Np, hipMemcpyHostToDevice, s));
    if (mean_X) {
        HIPCHK(hipMemcpyAsync(g->mean, mean_X, sizeof(double) * N, hipMemcpyHostToDevice, s));
        g->has_mean = true;
    } else if (g->has_mean) {
        HIPCHK(hipMemsetAsync(g->mean, 0, sizeof(double) * Np, s));
        g->has_mean = false;
    }
    HIPCHK(hipStreamSynchronize(s));                         // the staging vectors go out of scope
    int rc = factor_enqueue(g);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    g->pending = true;
    if (flags & BOSS_FIT_NO_SYNC) return BOSS_OK;
    return gp_finish(g, logpdf_out);
}

extern "C" int boss_gp_sync(boss_gp_t* g, double* logpdf_out) {
    if (!g) return fail(BOSS_E_INVALID, "gp is NULL");
    HIPCHK(hipSetDevice(g->ctx->device));
    if (!g->pending) {
        if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "no update pending and handle is not fitted");
        if (logpdf_out) *logpdf_out = -0.5 * (g->N * 1.8378770664093453 + g->host_res[0] + g->host_res[1]);
        return BOSS_OK;
    }
    return gp_finish(g, logpdf_out);
}

// ------------------------------------------------------------------------------------------
// block Cholesky append: augment_dataset! + model_posterior with unchanged
// hyper-parameters.  Only the block rows that contain new observations are (re)built: their
// Gram rows are swept through the finished panels 0..kb-1 (solve with L_kk, rank-128 update of
// the rest of the block row and of its δ^T entries), then the diagonal block is factorised like
// any other — the right-looking factorisation restricted to one block row, O(N²) per 128 rows.
// ------------------------------------------------------------------------------------------
static int gp_grow(boss_gp* g, int Nnew) {
    if (Nnew > MAX_ROWS) return fail(BOSS_E_INVALID, "more than 46080 observations are not supported");
    const int Np2 = round_up(Nnew, PRED_RB);
    if (Np2 <= g->Np) return BOSS_OK;
    Ctx* c = g->ctx;
    hipStream_t s = c->stream;
    const int Np = g->Np, d = g->d, nblk2 = Np2 / BLK, ld2 = Np2 + RHS_ROWS;
    const size_t szA = sizeof(double) * (size_t)ld2 * Np2;
    double* nw[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t bytes[8] = {sizeof(double) * d * Np2, sizeof(double) * d * Np2, sizeof(double) * Np2, sizeof(double) * Np2,
                             szA, sizeof(double) * nblk2 * 8 * 256, sizeof(double) * nblk2 * BLK * BLK,
                             sizeof(double) * (size_t)Np2 * PRED_RB};
    for (int i = 0; i < 8; ++i)
        if (dev_malloc((void**)&nw[i], bytes[i]) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipFree(nw[j]);
            return fail(BOSS_E_ALLOC, "device allocation failed while growing the posterior handle");
        }
    for (int i = 0; i < 5; ++i) HIPCHK(hipMemsetAsync(nw[i], 0, bytes[i], s));
    HIPCHK(hipMemcpy2DAsync(nw[0], sizeof(double) * Np2, g->Xraw, sizeof(double) * Np, sizeof(double) * Np, d,
                            hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpy2DAsync(nw[1], sizeof(double) * Np2, g->Xsc, sizeof(double) * Np, sizeof(double) * Np, d,
                            hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(nw[2], g->y, sizeof(double) * Np, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(nw[3], g->mean, sizeof(double) * Np, hipMemcpyDeviceToDevice, s));
    // factor: rows 0..Np-1 of every old column; the δ^T / z row block moves from row Np to row Np2
    HIPCHK(hipMemcpy2DAsync(nw[4], sizeof(double) * ld2, g->A, sizeof(double) * g->ld, sizeof(double) * Np, Np,
                            hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpy2DAsync(nw[4] + Np2, sizeof(double) * ld2, g->A + Np, sizeof(double) * g->ld,
                            sizeof(double) * RHS_ROWS, Np, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(nw[5], g->inv16, sizeof(double) * g->nblk * 8 * 256, hipMemcpyDeviceToDevice, s));
    dinv_join(g);
    HIPCHK(hipStreamSynchronize(s));
    double* old[8] = {g->Xraw, g->Xsc, g->y, g->mean, g->A, g->inv16, g->Dinv, g->Dinv2};
    for (double* p : old) (void)hipFree(p);
    if (g->LT) (void)hipFree(g->LT);
    if (g->DT2) (void)hipFree(g->DT2);
    if (g->avec) (void)hipFree(g->avec);
    g->LT = g->DT2 = g->avec = nullptr;
    if (g->Winv) (void)hipFree(g->Winv);
    if (g->Linv) (void)hipFree(g->Linv);
    g->Winv = g->Linv = nullptr;
    g->Xraw = nw[0]; g->Xsc = nw[1]; g->y = nw[2]; g->mean = nw[3]; g->A = nw[4]; g->inv16 = nw[5]; g->Dinv = nw[6]; g->Dinv2 = nw[7];
    g->Np = Np2;
    g->nblk = nblk2;
    g->ld = ld2;
    g->have_dinv = false;
    g->have_winv = false;
    g->few_calls = 0;
    g->have_lt = false;
    return BOSS_OK;
}

// Reserve storage for observations that will be appended later (no re-allocation / re-layout when
// they arrive).  The extra rows are identity padding of the factor: harmless, a little extra work per
// factorisation.  The handle is left unfitted: call boss_gp_update afterwards.
extern "C" int boss_gp_reserve(boss_gp_t* g, int N_total) {
    if (!g || N_total < 1) return fail(BOSS_E_INVALID, "bad arguments");
    NOT_FOR_AUG(g);
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    if (g->pending) (void)gp_finish(g, nullptr);
    int rc = gp_grow(g, N_total);
    if (rc) return rc;
    g->fitted = false;
    ++g->epoch;
    g->append_calls = 0;
    return BOSS_OK;
}

static int append_locked(boss_gp* g, int n, const double* X_new, const double* y_new, const double* mean_new, double* logpdf_out);

extern "C" int boss_gp_append(boss_gp_t* g, int n, const double* X_new, const double* y_new, const double* mean_new,
                              double* logpdf_out) {
    if (!g || n < 1 || !X_new || !y_new) return fail(BOSS_E_INVALID, "need a handle, n >= 1 and non-NULL X_new, y_new");
    NOT_FOR_AUG(g);
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    if (n > 1 && n <= 8 && g->have_winv && g->fitted && !g->pending && g->N + n <= g->Np) {
        // a handful of observations on resident inverse factors: n rank-one appends beat one block-row sweep
        int rc = BOSS_OK;
        for (int j = 0; j < n && rc == BOSS_OK; ++j)
            rc = append_locked(g, 1, X_new + (size_t)j * g->d, y_new + j, mean_new ? mean_new + j : nullptr, logpdf_out);
        return rc;
    }
    return append_locked(g, n, X_new, y_new, mean_new, logpdf_out);
}

// caller holds the context lock
static int append_locked(boss_gp* g, int n, const double* X_new, const double* y_new, const double* mean_new, double* logpdf_out) {
    Ctx* c = g->ctx;
    if (g->pending) {
        int rc0 = gp_finish(g, nullptr);
        if (rc0) return rc0;
    }
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "append needs a fitted handle (its hyper-parameters are reused)");
    hipStream_t s = c->stream;
    const int d = g->d, N0 = g->N, N1 = N0 + n;
    const int Np_before = g->Np;
    // one observation into existing storage, not the first such append on these hyper-parameters: rank-one append on the
    // resident inverse factors (built here if they are not resident yet)
    static const int winv_after = getenv("BOSS_WINV_AFTER") ? atoi(getenv("BOSS_WINV_AFTER")) : 2;
    bool fast = false;
    if (n == 1 && N1 <= g->Np && winv_after > 0 && sizeof(double) * (size_t)g->Np <= 144 * 1024) {
        if (!g->have_winv && ++g->append_calls >= 2 && g->few_calls >= 0) {
            const size_t bytes = sizeof(double) * (size_t)g->ld * g->Np;
            bool ok = (g->Winv != nullptr || dev_malloc((void**)&g->Winv, bytes) == hipSuccess) &&
                      (g->Linv != nullptr || dev_malloc((void**)&g->Linv, bytes) == hipSuccess);
            if (ok) {
                dinv_join(g);
                if (!g->have_dinv) {
                    dinv_launch(g, s);
                    g->have_dinv = true;
                }
                linv_enqueue(g, s, g->Winv, g->Linv);
                g->have_winv = true;
            } else {
                (void)hipGetLastError();
                g->few_calls = -(1 << 30);
            }
        }
        fast = g->have_winv;
    }
    int rc = gp_grow(g, N1);
    if (rc) return rc;
    g->fitted = false;
    g->have_lt = false;
    if (!fast) {
        g->have_dinv = false;
        g->have_winv = false;
        g->few_calls = 0;
    }
    {
        std::vector<double> xb;
        pack_points(xb, X_new, d, n, n, g->discrete.empty() ? nullptr : g->discrete.data());
        HIPCHK(hipMemcpy2DAsync(g->Xraw + N0, sizeof(double) * g->Np, xb.data(), sizeof(double) * n, sizeof(double) * n, d,
                                hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(g->y + N0, y_new, sizeof(double) * n, hipMemcpyHostToDevice, s));
        if (mean_new) {
            HIPCHK(hipMemcpyAsync(g->mean + N0, mean_new, sizeof(double) * n, hipMemcpyHostToDevice, s));
            g->has_mean = true;
        }
        HIPCHK(hipStreamSynchronize(s));       // staging buffers go out of scope
    }
    g->N = N1;
    if (fast) {
        const int Np = g->Np, ld = g->ld, nwg = Np / WINV_ROWS;
        dinv_join(g);
        if (!g->have_dinv) {                                 // the patched blocks must exist
            dinv_launch(g, s);
            g->have_dinv = true;
        }
        rc = ws_reserve(c->few, sizeof(double) * ((size_t)Np * 32 + (size_t)nwg * 8 + 2 * (size_t)Np + 8));
        if (rc) return rc;
        rc = ws_reserve(c->csc, sizeof(double) * (size_t)d * 64);
        if (rc) return rc;
        double* R = (double*)c->few.p;                       // k = K(X, x_new) in column 0 of a 32-wide tile
        double* part = R + (size_t)Np * 32;
        double* lvec = part + (size_t)nwg * 8;
        double* wvec = lvec + Np;
        double* dz = wvec + Np;
        double* Csc = (double*)c->csc.p;
        HIPCHK(hipMemsetAsync(g->info, 0, sizeof(int), s));
        hipLaunchKernelGGL(scale_points_kernel, dim3((Np + 255) / 256, 1, 1), dim3(256), 0, s, g->Xraw, g->Xsc, (size_t)0, g->invlam, d, Np);
        // the new (scaled) point as candidate 0 of a 64-wide candidate block
        HIPCHK(hipMemcpy2DAsync(Csc, sizeof(double) * 64, g->Xsc + N0, sizeof(double) * Np, sizeof(double), d, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(kstar_rows_kernel, dim3(Np / 256, 1), dim3(256), sizeof(double) * d * 32, s, (const double*)g->Xsc, Np, N0,
                           (const double*)Csc, d, 64, g->kernel, g->amp2, R, 1);
        hipLaunchKernelGGL(winv_gemv_kernel<1>, dim3(nwg), dim3(256), sizeof(double) * (size_t)Np, s, (const double*)g->Winv, ld, Np,
                           (const double*)g->A, ld, (const double*)R, 1, part, lvec);
        hipLaunchKernelGGL(append_scalars_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nwg, (const double*)g->hyp,
                           (const double*)g->y, (const double*)g->mean, N0, g->scal, dz, g->info);
        hipLaunchKernelGGL(linv_col_gemv_kernel, dim3((N0 + 7) / 8), dim3(256), 0, s, (const double*)g->Linv, ld, N0,
                           (const double*)lvec, wvec);
        hipLaunchKernelGGL(append_write_kernel, dim3(N0 / 256 + 1), dim3(256), 0, s, g->A, ld, Np, N0, (const double*)lvec,
                           (const double*)wvec, (const double*)dz, g->Linv, g->Winv, g->Dinv, g->Dinv2, g->inv16);
        HIPCHK(hipMemcpyAsync(g->host_res, g->scal, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(&g->host_res[2], g->info, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipGetLastError());
        g->pending = true;
        rc = gp_finish(g, logpdf_out);
        if (rc) {                                            // not positive definite: nothing resident describes the data any more
            g->have_dinv = false;
            g->have_winv = false;
        }
        return rc;
    }
    // block rows to (re)build: those holding new observations and, when the storage has just grown, the pure padding
    // block rows behind them as well (identity blocks of the factor and their inverses: the new arrays are uninitialised)
    const int kb0 = N0 / BLK, kb1 = (g->Np != Np_before) ? g->nblk - 1 : (N1 - 1) / BLK;
    if (kb1 - kb0 + 1 > 4 || kb1 - kb0 + 1 >= g->nblk) {
        // most of the matrix is new: a plain re-factorisation is cheaper than block-row sweeps
        rc = factor_enqueue(g);
        if (rc) return rc;
    } else {
        const int Np = g->Np, ld = g->ld;
        dinv_join(g);
        HIPCHK(hipMemsetAsync(g->info, 0, sizeof(int), s));
        hipLaunchKernelGGL(scale_points_kernel, dim3((Np + 255) / 256, 1, 1), dim3(256), 0, s, g->Xraw, g->Xsc, (size_t)0,
                           g->invlam, d, Np);
        for (int kb = kb0; kb <= kb1; ++kb) {
            hipLaunchKernelGGL(rhs_rows_kernel, dim3(1, 1, 1), dim3(BLK), 0, s, g->A, ld, (size_t)0, N1, Np, g->y, g->mean,
                               (size_t)0, kb * BLK);
            hipLaunchKernelGGL(gram_kernel, dim3(4 * kb + 3, 1, 1), dim3(256), 0, s, g->Xsc, (size_t)0, d, N1, Np, g->kernel,
                               g->hyp, g->A, ld, (size_t)0, kb * (2 * kb + 1));
            for (int k = 0; k < kb; ++k) {
                hipLaunchKernelGGL(potrf_trsm_kernel, dim3(8, 1, 1), dim3(64), 0, s, g->A, ld, (size_t)0, k, g->inv16, (size_t)0,
                                   kb * BLK);
                hipLaunchKernelGGL(potrf_rowupd_kernel, dim3(4 * (kb - k) + 1), dim3(256), 0, s, g->A, ld, k, kb, Np);
            }
            hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, 1, 1), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, g->A, ld, (size_t)0, kb,
                               g->inv16, (size_t)0, g->info);
            hipLaunchKernelGGL(potrf_trsm_kernel, dim3(1, 1, 1), dim3(64), 0, s, g->A, ld, (size_t)0, kb, g->inv16, (size_t)0,
                               Np);
        }
        dinv_eager(g);
        hipLaunchKernelGGL(potrf_logdet_kernel, dim3(1, 1, 1), dim3(256), 0, s, g->A, ld, (size_t)0, N1, Np, g->scal);
        HIPCHK(hipMemcpyAsync(g->host_res, g->scal, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(&g->host_res[2], g->info, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipGetLastError());
    g->pending = true;
    return gp_finish(g, logpdf_out);
}

extern "C" int boss_gp_fit(int device, int kernel, int d, int N, const double* X, const double* y,
                           const double* mean_X, const double* lengthscale, double amplitude, double noise_std,
                           const unsigned char* discrete, boss_gp_t** out, double* logpdf_out) {
    if (!out) return fail(BOSS_E_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validate_hyper(d, lengthscale, amplitude, noise_std);
    if (rc) return rc;
    boss_gp_t* g = nullptr;
    rc = boss_gp_create(device, kernel, d, N, X, y, discrete, &g);
    if (rc) return rc;
    rc = boss_gp_update(g, lengthscale, amplitude, noise_std, mean_X, BOSS_FIT_DEFAULT, logpdf_out);
    if (rc) {
        std::string keep = g_last_error;
        boss_gp_free(g);
        g_last_error = keep;
        return rc;
    }
    *out = g;
    return BOSS_OK;
}

extern "C" void boss_gp_free(boss_gp_t* g) {
    if (!g) return;
    if (g->ctx) {
        (void)hipSetDevice(g->ctx->device);
        (void)hipStreamSynchronize(g->ctx->stream);
        if (g->dinv_pending) (void)hipEventSynchronize(g->dinv_ev);
    }
    gp_release(g);
}

extern "C" int boss_gp_get_factor(const boss_gp_t* g, double* L_out, double* z_out) {
    if (!g) return fail(BOSS_E_INVALID, "gp is NULL");
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    HIPCHK(hipSetDevice(g->ctx->device));
    HIPCHK(hipStreamSynchronize(g->ctx->stream));
    const int N = g->N;
    if (L_out) {
        HIPCHK(hipMemcpy2D(L_out, sizeof(double) * N, g->A, sizeof(double) * g->ld, sizeof(double) * N, N,
                           hipMemcpyDeviceToHost));
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < j; ++i) L_out[(size_t)j * N + i] = 0.0;
    }
    if (z_out) {
        HIPCHK(hipMemcpy2D(z_out, sizeof(double), g->A + g->Np, sizeof(double) * g->ld, sizeof(double), N,
                           hipMemcpyDeviceToHost));
    }
    return BOSS_OK;
}

// ------------------------------------------------------------------------------------------
// batched log-likelihood
// ------------------------------------------------------------------------------------------
extern "C" int boss_gp_loglike_batch(int device, int kernel, int d, int N, const double* X, const double* y,
                                     const double* mean_X, int mean_stride, const unsigned char* discrete, int S,
                                     const double* lengthscales, const double* amplitudes, const double* noise_stds,
                                     double* ll_out, int* status_out) {
    if (kernel < 0 || kernel > 2) return fail(BOSS_E_INVALID, "unknown kernel id");
    if (d < 1 || N < 1 || S < 0 || !X || !y || !ll_out) return fail(BOSS_E_INVALID, "bad arguments");
    if (S == 0) return BOSS_OK;
    if (!lengthscales || !amplitudes || !noise_stds) return fail(BOSS_E_INVALID, "NULL hyper-parameter array");
    if (mean_X && mean_stride != 0 && mean_stride != N) return fail(BOSS_E_INVALID, "mean_stride must be 0 or N");
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mtx);
    hipStream_t s = c->stream;
    const int Np = round_up(N, BLK), nblk = Np / BLK, ld = Np + RHS_ROWS;
    const size_t bstride = (size_t)ld * Np;
    // chunk the batch so the matrices stay below ~12 GiB
    size_t per = bstride * sizeof(double);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)S, ((size_t)12 << 30) / per));
    const size_t xs_bstride = (size_t)d * Np;
    rc = ws_reserve(c->batchA, per * chunk);
    if (rc) return rc;
    rc = ws_reserve(c->batchX, sizeof(double) * (xs_bstride * (chunk + 1) + (size_t)Np * (chunk + 1)));
    if (rc) return rc;
    const size_t inv16_b = (size_t)nblk * 8 * 256;
    rc = ws_reserve(c->batchMisc, sizeof(double) * ((size_t)chunk * (inv16_b + 2 + 2 + d)) + sizeof(int) * chunk + 64);
    if (rc) return rc;
    double* A = (double*)c->batchA.p;
    double* Xraw = (double*)c->batchX.p;
    double* Xsc = Xraw + xs_bstride;
    double* ydev = Xsc + xs_bstride * chunk;
    double* meandev = ydev + Np;                       // chunk × Np (or Np when shared)
    double* inv16 = (double*)c->batchMisc.p;
    double* hyp = inv16 + inv16_b * chunk;
    double* scal = hyp + 2 * (size_t)chunk;
    double* invlam = scal + 2 * (size_t)chunk;
    int* info = (int*)(invlam + (size_t)d * chunk);

    std::vector<double> buf;
    pack_points(buf, X, d, N, Np, discrete);
    HIPCHK(hipMemcpy(Xraw, buf.data(), sizeof(double) * xs_bstride, hipMemcpyHostToDevice));
    std::vector<double> yb(Np, 0.0);
    std::copy(y, y + N, yb.begin());
    HIPCHK(hipMemcpy(ydev, yb.data(), sizeof(double) * Np, hipMemcpyHostToDevice));

    std::vector<double> h_invlam((size_t)d * chunk), h_hyp(2 * (size_t)chunk), h_scal(2 * (size_t)chunk), h_mean;
    std::vector<int> h_info(chunk), valid(chunk);
    for (int s0 = 0; s0 < S; s0 += chunk) {
        const int nb = std::min(chunk, S - s0);
        for (int b = 0; b < nb; ++b) {
            const double* lam = lengthscales + (size_t)(s0 + b) * d;
            bool ok = amplitudes[s0 + b] >= 0.0 && noise_stds[s0 + b] >= 0.0;
            for (int k = 0; k < d; ++k) ok = ok && lam[k] >= 0.0;
            valid[b] = ok;
            const double amp = (ok ? amplitudes[s0 + b] : 1.0) + MIN_PARAM_VALUE;
            const double sig = (ok ? noise_stds[s0 + b] : 1.0) + MIN_PARAM_VALUE;
            for (int k = 0; k < d; ++k) h_invlam[(size_t)b * d + k] = 1.0 / ((ok ? lam[k] : 1.0) + MIN_PARAM_VALUE);
            h_hyp[2 * b] = amp * amp;
            h_hyp[2 * b + 1] = sig * sig;
        }
        HIPCHK(hipMemcpy(invlam, h_invlam.data(), sizeof(double) * d * nb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(hyp, h_hyp.data(), sizeof(double) * 2 * nb, hipMemcpyHostToDevice));
        size_t mean_b = 0;
        const double* mean_arg = nullptr;
        if (mean_X) {
            if (mean_stride == 0) {
                h_mean.assign(Np, 0.0);
                std::copy(mean_X, mean_X + N, h_mean.begin());
                HIPCHK(hipMemcpy(meandev, h_mean.data(), sizeof(double) * Np, hipMemcpyHostToDevice));
            } else {
                h_mean.assign((size_t)Np * nb, 0.0);
                for (int b = 0; b < nb; ++b)
                    std::copy(mean_X + (size_t)(s0 + b) * N, mean_X + (size_t)(s0 + b + 1) * N, h_mean.begin() + (size_t)b * Np);
                HIPCHK(hipMemcpy(meandev, h_mean.data(), sizeof(double) * Np * nb, hipMemcpyHostToDevice));
                mean_b = Np;
            }
            mean_arg = meandev;
        }
        HIPCHK(hipMemsetAsync(info, 0, sizeof(int) * nb, s));
        {
            ProfScope ps(c, "prep");
            hipLaunchKernelGGL(scale_points_kernel, dim3((Np + 255) / 256, 1, nb), dim3(256), 0, s, Xraw, Xsc, xs_bstride,
                               invlam, d, Np);
            hipLaunchKernelGGL(rhs_rows_kernel, dim3((Np + 255) / 256, 1, nb), dim3(256), 0, s, A, ld, bstride, N, Np,
                               ydev, mean_arg, mean_b, 0);
        }
        gram_enqueue(c, Xsc, xs_bstride, d, N, Np, kernel, hyp, A, ld, bstride, nb);
        potrf_enqueue(c, A, ld, Np, nb, bstride, inv16, inv16_b, info);
        {
            ProfScope ps(c, "logdet");
            hipLaunchKernelGGL(potrf_logdet_kernel, dim3(1, 1, nb), dim3(256), 0, s, A, ld, bstride, N, Np, scal);
        }
        HIPCHK(hipMemcpyAsync(h_scal.data(), scal, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h_info.data(), info, sizeof(int) * nb, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipGetLastError());
        for (int b = 0; b < nb; ++b) {
            const double logdet = h_scal[2 * b], zz = h_scal[2 * b + 1];
            int st = BOSS_OK;
            double ll;
            if (!valid[b]) {
                st = BOSS_E_INVALID;
                ll = -std::numeric_limits<double>::infinity();
            } else if (h_info[b] != 0 || !std::isfinite(logdet) || !std::isfinite(zz)) {
                st = BOSS_E_NOT_PD;
                ll = -std::numeric_limits<double>::infinity();   // safe_data_loglike: exception → -Inf
            } else {
                ll = -0.5 * (N * 1.8378770664093453 + logdet + zz);
            }
            ll_out[s0 + b] = ll;
            if (status_out) status_out[s0 + b] = st;
        }
    }
    return BOSS_OK;
}

// ------------------------------------------------------------------------------------------
// candidates + prediction
// ------------------------------------------------------------------------------------------
extern "C" int boss_cand_create(int device, int d, int M, const double* Xs, boss_cand_t** out) {
    if (!out) return fail(BOSS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (d < 1 || M < 1 || !Xs) return fail(BOSS_E_INVALID, "need d >= 1, M >= 1 and non-NULL Xs");
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    boss_cand* cd = new boss_cand();
    cd->ctx = c;
    cd->d = d;
    cd->M = M;
    cd->Mp = round_up(M, 64);
    if (dev_malloc((void**)&cd->Craw, sizeof(double) * d * cd->Mp) != hipSuccess) {
        delete cd;
        return fail(BOSS_E_ALLOC, "device allocation failed");
    }
    std::vector<double> buf;
    pack_points(buf, Xs, d, M, cd->Mp, nullptr);
    HIPCHK(hipMemcpy(cd->Craw, buf.data(), sizeof(double) * d * cd->Mp, hipMemcpyHostToDevice));
    *out = cd;
    return BOSS_OK;
}

// One-shot entry points (predict / predict_cov / predict_grad / acq_ei_grad) upload their candidates into a
// grow-only per-device workspace instead of allocating: hipMalloc/hipFree synchronise the device and cost
// a few hundred microseconds per call.  Caller holds the context lock.
static int temp_cand(Ctx* c, int d, int M, const double* Xs, boss_cand* cd) {
    cd->ctx = c;
    cd->d = d;
    cd->M = M;
    cd->Mp = round_up(M, 64);
    int rc = ws_reserve(c->craw, sizeof(double) * (size_t)d * cd->Mp);
    if (rc) return rc;
    cd->Craw = (double*)c->craw.p;
    std::vector<double> buf;
    pack_points(buf, Xs, d, M, cd->Mp, nullptr);
    const size_t bytes = sizeof(double) * d * cd->Mp;
    if (bytes <= PINNED_UP_BYTES) {
        // few candidates: stage through pinned memory, no synchronisation (every entry point that calls this ends with
        // a stream synchronisation before it returns, so the area is free again at the next call)
        void* stage = (char*)c->pinned + PINNED_UP_OFF;
        HIPCHK(hipEventSynchronize(c->ev_up));             // the previous upload from this area (normally long complete)
        std::memcpy(stage, buf.data(), bytes);
        HIPCHK(hipMemcpyAsync(cd->Craw, stage, bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipEventRecord(c->ev_up, c->stream));
        return BOSS_OK;
    }
    HIPCHK(hipMemcpyAsync(cd->Craw, buf.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));               // staging buffer goes out of scope
    return BOSS_OK;
}

extern "C" void boss_cand_free(boss_cand_t* cd) {
    if (!cd) return;
    if (cd->ctx) {
        (void)hipSetDevice(cd->ctx->device);
        (void)hipStreamSynchronize(cd->ctx->stream);
    }
    if (cd->Craw) (void)hipFree(cd->Craw);
    delete cd;
}

// scaled (and, for DiscreteKernel dims, rounded) candidates for one GP
__global__ void scale_cand_kernel(const double* __restrict__ Craw, double* __restrict__ Csc, const double* __restrict__ invlam,
                                  const unsigned char* __restrict__ discrete, int d, int Mp) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Mp) return;
    for (int k = 0; k < d; ++k) {
        double v = Craw[(size_t)k * Mp + j];
        if (discrete && discrete[k]) v = rint(v);
        Csc[(size_t)k * Mp + j] = v * invlam[k];
    }
}

// candidate tiles (of 32) up to which the resident-inverse GEMMs replace the step-by-step substitution (measured at N=4096:
// 0.19 vs 0.59 ms at 7 tiles, 0.46 vs 0.85 ms at 32, 0.75 vs 1.16 ms at 64, 1.36 vs 1.85 ms at 128 = the whole few-candidates range)
static int invgemm_max_tiles() {
    static const int v = getenv("BOSS_INVGEMM_MAX_TILES") ? atoi(getenv("BOSS_INVGEMM_MAX_TILES")) : 128;
    return v;
}


// U = L⁻ᵀ (upper, leading dimension g->ld) by recursive doubling from the 256×256 diagonal inverses; Lw (same shape)
// is the lower work matrix.  Dinv2 must be current on stream s.
static void linv_enqueue(boss_gp* g, hipStream_t s, double* U, double* Lw) {
    const int Np = g->Np, ld = g->ld;
    hipLaunchKernelGGL(linv_seed_kernel, dim3(PRED_RB, Np / PRED_RB), dim3(PRED_RB), 0, s, (const double*)g->Dinv2, Lw, ld, U, ld);
    for (int sz = PRED_RB; sz < Np; sz *= 2) {
        const int pairs = (Np + 2 * sz - 1) / (2 * sz), tiles = (sz / BLK) * (sz / BLK);
        hipLaunchKernelGGL((linv_level_kernel<SyrkG, 1>), dim3(tiles, pairs), dim3(256), 0, s, (const double*)g->A, ld, Lw, ld, U, ld, Np, sz);
        hipLaunchKernelGGL((linv_level_kernel<SyrkG, 2>), dim3(tiles, pairs), dim3(256), 0, s, (const double*)g->A, ld, Lw, ld, U, ld, Np, sz);
    }
}

// enqueue μ/σ² (unclipped) of one posterior at resident candidates into device arrays mu, var (length ≥ M)
static int predict_enqueue(boss_gp* g, const boss_cand* cd, const double* mean_s_dev, double* mu, double* var,
                           bool for_grad = false, const double* clam_dev = nullptr, const double* camp_dev = nullptr,
                           bool need_v = false) {
    // need_v: the caller reads V = L⁻¹K* from the slab scratch afterwards (covariances); for_grad implies it
    Ctx* c = g->ctx;
    hipStream_t s = c->stream;
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    if (g->gibbs && !clam_dev)
        return fail(BOSS_E_INVALID, "nonstationary posteriors predict through boss_ngp_predict (λ(x*), α(x*) are needed)");
    if (cd->ctx != c) return fail(BOSS_E_INVALID, "candidates and posterior live on different devices");
    if (cd->d != g->d) return fail(BOSS_E_INVALID, "candidate dimension differs from the model's x_dim");
    dinv_join(g);
    if (!g->have_dinv) {
        ProfScope ps(c, "dinv");
        dinv_launch(g, s);
        g->have_dinv = true;
    }
    const int Mp = cd->Mp;
    // 64 candidates per workgroup once that still fills the machine (BOSS_FORCE_BN64=1: tests)
    static const bool force64 = getenv("BOSS_FORCE_BN64") && atoi(getenv("BOSS_FORCE_BN64"));
    // the gradient pass (adjoint substitution) runs on 32-candidate slabs
    const int BN = for_grad ? 32 : ((cd->M >= 64 * 256 || force64) ? 64 : 32);
    const int tiles = (cd->M + BN - 1) / BN;
    int rc = ws_reserve(c->csc, sizeof(double) * (size_t)g->d * Mp);
    if (rc) return rc;
    rc = ws_reserve(c->vscratch, sizeof(double) * (size_t)tiles * BN * g->Np * (for_grad ? 2 : 1));   // gradients: V and W slabs
    if (rc) return rc;
    double* Csc = (double*)c->csc.p;
    if (!g->gibbs)
        hipLaunchKernelGGL(scale_cand_kernel, dim3((Mp + 255) / 256), dim3(256), 0, s, cd->Craw, Csc, g->invlam,
                           g->discrete_dev, g->d, Mp);
    const int dbg = getenv("BOSS_DBG") ? atoi(getenv("BOSS_DBG")) : 0;
    static const bool no_few = getenv("BOSS_NO_FEW") && atoi(getenv("BOSS_NO_FEW"));
    static const int few_max_tiles = getenv("BOSS_FEW_MAX_TILES") ? atoi(getenv("BOSS_FEW_MAX_TILES")) : 128;
    const int ftiles = (cd->M + 31) / 32;
    const size_t aug_lds = sizeof(double) * ((size_t)g->d * (64 + 256) + g->d);
    if (ftiles <= few_max_tiles && g->Np >= 4 * PRED_RB && g->d <= 64 && !no_few && (BN == 32 || for_grad)) {
        // few candidates (the fused kernel would occupy `ftiles` of 256 CUs for its whole latency):
        // right-looking substitution, every 256-row step spread over the chip
        typedef PredG32 G;
        typedef GemmDirect<4, 1, 2, 2, 8> GU;                // 128×32 update tiles
        ProfScope ps(c, "predict");
        const int nb = g->Np / PRED_RB;
        const int nwg = g->Np / WINV_ROWS;
        rc = ws_reserve(c->few, sizeof(double) * ((size_t)ftiles * ((size_t)g->Np * 32 + 64) + (size_t)nwg * 8));
        if (rc) return rc;
        double* R = (double*)c->few.p;                       // residuals [tile][Np][32], start as K*
        double* ssmz = R + (size_t)ftiles * g->Np * 32;
        double* V = (double*)c->vscratch.p;
        // repeated calls with few candidates on one factorisation: from the second call on both inverse factors are
        // resident — one to four candidates take a single pass over L⁻ᵀ (winv_gemv_kernel), more take two GEMMs
        // without sequential steps (inv_fwd_kernel / inv_bwd_kernel)
        static const int winv_after = getenv("BOSS_WINV_AFTER") ? atoi(getenv("BOSS_WINV_AFTER")) : 2;
        const size_t winv_lds = sizeof(double) * (size_t)g->Np * WINV_MAX_M;
        if (winv_after > 0 && !g->have_winv && ++g->few_calls >= winv_after) {
            const size_t bytes = sizeof(double) * (size_t)g->ld * g->Np;
            bool ok = (g->Winv != nullptr || dev_malloc((void**)&g->Winv, bytes) == hipSuccess) &&
                      (g->Linv != nullptr || dev_malloc((void**)&g->Linv, bytes) == hipSuccess);
            if (ok) {
                linv_enqueue(g, s, g->Winv, g->Linv);
                g->have_winv = true;
            } else {
                (void)hipGetLastError();                 // no memory for the inverses: stay on the substitution path
                g->few_calls = -(1 << 30);
            }
        }
        const bool use_winv = g->have_winv && !for_grad && !need_v && cd->M <= WINV_MAX_M && winv_lds <= 144 * 1024;
        const bool use_invgemm = g->have_winv && !use_winv && ftiles <= invgemm_max_tiles();   // beyond: the step path is faster
        if (!use_winv) (void)hipMemsetAsync(ssmz, 0, sizeof(double) * 64 * ftiles, s);
        if (g->aug)
            hipLaunchKernelGGL(aug_kstar_kernel, dim3(g->Np / 256, ftiles), dim3(256), aug_lds, s, (const double*)g->Xraw, g->ldx,
                               g->d, g->npts, g->N, g->Np, (const double*)cd->Craw, Mp, g->kernel, g->amp2,
                               (const double*)g->invlam, R, 32);
        else if (g->gibbs)
            hipLaunchKernelGGL(gibbs_kstar_kernel<32>, dim3(g->Np / 256, ftiles), dim3(256), sizeof(double) * (2 * g->d + 1) * 32, s,
                               (const double*)g->Xraw, (const double*)g->lamX, (const double*)g->ampX, g->d, g->N, g->Np,
                               (const double*)cd->Craw, clam_dev, camp_dev, Mp, R);
        else
            hipLaunchKernelGGL(kstar_rows_kernel, dim3(g->Np / 256, ftiles), dim3(256), sizeof(double) * g->d * 32, s,
                               (const double*)g->Xsc, g->Np, g->N, (const double*)Csc, g->d, Mp, g->kernel, g->amp2, R,
                               use_winv ? cd->M : 32);
        if (use_invgemm) {
            const int nrb = g->Np / BLK;
            rc = ws_reserve(c->lgC, sizeof(double) * (size_t)ftiles * nrb * 64);
            if (rc) return rc;
            double* ssp = (double*)c->lgC.p;
            hipLaunchKernelGGL(inv_fwd_kernel<GU>, dim3(ftiles, nrb), dim3(GU::NTHREADS), 0, s, (const double*)g->Linv, g->ld, g->Np,
                               (const double*)g->A, g->ld, (const double*)R, V, ssp);
            hipLaunchKernelGGL(inv_fwd_finish_kernel, dim3(ftiles), dim3(256), 0, s, (const double*)ssp, nrb, mean_s_dev, cd->M,
                               g->amp2, g->aug ? 1 : g->gibbs ? 2 : 0, mu, var);
            if (g->gibbs) hipLaunchKernelGGL(gibbs_var_kernel, dim3((cd->M + 255) / 256), dim3(256), 0, s, var, camp_dev, cd->M);
            HIPCHK(hipGetLastError());
            return BOSS_OK;
        }
        if (use_winv) {
            double* part = ssmz + 64 * (size_t)ftiles;
            const int mc = cd->M == 1 ? 1 : cd->M == 2 ? 2 : 4;
            const size_t lds = sizeof(double) * (size_t)g->Np * mc;
            auto kfn = mc == 1 ? winv_gemv_kernel<1> : mc == 2 ? winv_gemv_kernel<2> : winv_gemv_kernel<4>;
            hipLaunchKernelGGL(kfn, dim3(nwg), dim3(256), lds, s, (const double*)g->Winv, g->ld, g->Np,
                               (const double*)g->A, g->ld, (const double*)R, cd->M, part, (double*)nullptr);
            hipLaunchKernelGGL(winv_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nwg, cd->M, mean_s_dev, g->amp2,
                               g->aug ? 1 : g->gibbs ? 2 : 0, mu, var);
            if (g->gibbs) hipLaunchKernelGGL(gibbs_var_kernel, dim3(1), dim3(256), 0, s, var, camp_dev, cd->M);
            HIPCHK(hipGetLastError());
            return BOSS_OK;
        }
        for (int ib = 0; ib < nb; ++ib) {
            hipLaunchKernelGGL(few_finish_kernel<G>, dim3(ftiles), dim3(G::NTHREADS), PredictLds<G>::BYTES, s, (const double*)g->A,
                               g->ld, g->Np, ib, (const double*)R, (const double*)g->Dinv2, V, ssmz, ib == nb - 1 ? 1 : 0,
                               mean_s_dev, cd->M, g->amp2, mu, var, g->aug ? 1 : g->gibbs ? 2 : 0);
            const int nupd = (nb - 1 - ib) * (PRED_RB / BLK);
            if (nupd > 0)
                hipLaunchKernelGGL(few_update_kernel<GU>, dim3(ftiles, nupd), dim3(GU::NTHREADS), 0, s, (const double*)g->A, g->ld,
                                   g->Np, ib, (const double*)V, R);
        }
        if (g->gibbs) hipLaunchKernelGGL(gibbs_var_kernel, dim3((cd->M + 255) / 256), dim3(256), 0, s, var, camp_dev, cd->M);
        HIPCHK(hipGetLastError());
        return BOSS_OK;
    }
    if (g->aug || g->gibbs) {
        ProfScope ps(c, "predict");
        double* V = (double*)c->vscratch.p;
        if (g->aug)
            hipLaunchKernelGGL(aug_kstar_kernel, dim3(g->Np / 256, tiles), dim3(256), aug_lds, s, (const double*)g->Xraw, g->ldx, g->d,
                               g->npts, g->N, g->Np, (const double*)cd->Craw, Mp, g->kernel, g->amp2, (const double*)g->invlam, V, BN);
        else if (BN == 32)
            hipLaunchKernelGGL(gibbs_kstar_kernel<32>, dim3(g->Np / 256, tiles), dim3(256), sizeof(double) * (2 * g->d + 1) * 32, s,
                               (const double*)g->Xraw, (const double*)g->lamX, (const double*)g->ampX, g->d, g->N, g->Np,
                               (const double*)cd->Craw, clam_dev, camp_dev, Mp, V);
        else
            hipLaunchKernelGGL(gibbs_kstar_kernel<64>, dim3(g->Np / 256, tiles), dim3(256), sizeof(double) * (2 * g->d + 1) * 64, s,
                               (const double*)g->Xraw, (const double*)g->lamX, (const double*)g->ampX, g->d, g->N, g->Np,
                               (const double*)cd->Craw, clam_dev, camp_dev, Mp, V);
        if (BN == 32) {
            typedef PredG32 G;
            hipLaunchKernelGGL((predict_kernel<G, true>), dim3(tiles), dim3(G::NTHREADS), PredictLds<G>::BYTES, s, g->A, g->ld, g->Np,
                               g->N, g->Dinv2, g->Xsc, Csc, g->d, Mp, g->kernel, g->amp2, V, mean_s_dev, cd->M, mu, var, dbg);
        } else {
            typedef PredG64 G;
            hipLaunchKernelGGL((predict_kernel<G, true>), dim3(tiles), dim3(G::NTHREADS), PredictLds<G>::BYTES, s, g->A, g->ld, g->Np,
                               g->N, g->Dinv, g->Xsc, Csc, g->d, Mp, g->kernel, g->amp2, V, mean_s_dev, cd->M, mu, var, dbg);
        }
        if (g->gibbs) hipLaunchKernelGGL(gibbs_var_kernel, dim3((cd->M + 255) / 256), dim3(256), 0, s, var, camp_dev, cd->M);
        HIPCHK(hipGetLastError());
        return BOSS_OK;
    }
    {
        ProfScope ps(c, "predict");
        if (BN == 32) {
            typedef PredG32 G;
            hipLaunchKernelGGL(predict_kernel<G>, dim3(tiles), dim3(G::NTHREADS), PredictLds<G>::BYTES, s, g->A, g->ld, g->Np,
                               g->N, g->Dinv2, g->Xsc, Csc, g->d, Mp, g->kernel, g->amp2, (double*)c->vscratch.p, mean_s_dev, cd->M, mu, var, dbg);
        } else {
            typedef PredG64 G;
            hipLaunchKernelGGL(predict_kernel<G>, dim3(tiles), dim3(G::NTHREADS), PredictLds<G>::BYTES, s, g->A, g->ld, g->Np,
                               g->N, g->Dinv, g->Xsc, Csc, g->d, Mp, g->kernel, g->amp2, (double*)c->vscratch.p, mean_s_dev, cd->M, mu, var, dbg);
        }
    }
    HIPCHK(hipGetLastError());
    return BOSS_OK;
}

extern "C" int boss_gp_predict(boss_gp_t* g, int M, const double* Xs, const double* mean_Xs, double* mu, double* var,
                               long* bad_index) {
    if (!g || !Xs || !mu || !var) return fail(BOSS_E_INVALID, "NULL argument");
    if (M < 1) return fail(BOSS_E_INVALID, "M must be >= 1");
    if (bad_index) *bad_index = -1;
    if (g->aug && mean_Xs) return fail(BOSS_E_INVALID, "gradient-observation posteriors take no prior mean (gradient_gp.jl:334-337)");
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);                 // the per-device scratch areas are shared
    boss_cand cand_tmp;
    boss_cand* cd = &cand_tmp;
    int rc = temp_cand(c, g->d, M, Xs, cd);
    if (rc) return rc;
    rc = ws_reserve(c->pred, sizeof(double) * (3 * (size_t)M + 2));   // mu | var | bad | mean
    if (rc) return rc;
    double* dev = (double*)c->pred.p;
    double *dmu = dev, *dvar = dev + M, *dmean = dev + 2 * (size_t)M + 1;
    unsigned long long* dbad = (unsigned long long*)(dev + 2 * (size_t)M);
    hipStream_t s = c->stream;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
    };
    if (mean_Xs) (void)hipMemcpyAsync(dmean, mean_Xs, sizeof(double) * M, hipMemcpyHostToDevice, s);
    (void)hipMemsetAsync(dbad, 0xff, sizeof(unsigned long long), s);
    rc = predict_enqueue(g, cd, mean_Xs ? dmean : nullptr, dmu, dvar);
    if (rc) {
        cleanup();
        return rc;
    }
    hipLaunchKernelGGL(clip_var_kernel, dim3((M + 255) / 256), dim3(256), 0, s, dvar, M, dbad);
    unsigned long long bad = 0;
    const size_t down = sizeof(double) * (2 * (size_t)M + 1);
    hipError_t e;
    if (down <= PINNED_DOWN_BYTES) {                         // few candidates: one copy into pinned memory
        double* stage = (double*)((char*)c->pinned + PINNED_DOWN_OFF);
        (void)hipMemcpyAsync(stage, dev, down, hipMemcpyDeviceToHost, s);
        e = hipStreamSynchronize(s);
        std::memcpy(mu, stage, sizeof(double) * M);
        std::memcpy(var, stage + M, sizeof(double) * M);
        std::memcpy(&bad, stage + 2 * (size_t)M, sizeof bad);
    } else {
        (void)hipMemcpyAsync(mu, dmu, sizeof(double) * M, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(var, dvar, sizeof(double) * M, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(&bad, dbad, sizeof bad, hipMemcpyDeviceToHost, s);
        e = hipStreamSynchronize(s);
    }
    if (e != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e));
    if (bad != ~0ULL) {
        if (bad_index) *bad_index = (long)bad;
        char msg[160];
        std::snprintf(msg, sizeof msg, "The posterior GP predicted variance %g but only values above -1e-08 are tolerated. (DomainError)", var[bad]);
        return fail(BOSS_E_NEG_VAR, msg);
    }
    return BOSS_OK;
}

// mean_and_var of a NonstationaryGP posterior (GaussianProcessPosterior over the Gibbs kernel,
// nonstationary_gp.jl:153-157 -> gaussian_process.jl:143-194): lam_Xs d×M and amp_Xs M are the caller's latent
// models at the candidates (evaluated at the ROUNDED candidate where dims are discrete, as DiscreteKernel does).
extern "C" int boss_ngp_predict(boss_gp_t* g, int M, const double* Xs, const double* lam_Xs, const double* amp_Xs,
                                const double* mean_Xs, double* mu, double* var, long* bad_index) {
    if (!g || !Xs || !lam_Xs || !amp_Xs || !mu || !var) return fail(BOSS_E_INVALID, "NULL argument");
    if (!g->gibbs) return fail(BOSS_E_INVALID, "handle was not created by boss_ngp_create");
    if (M < 1) return fail(BOSS_E_INVALID, "M must be >= 1");
    if (bad_index) *bad_index = -1;
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    const int d = g->d, Mp = round_up(M, 64);
    std::vector<double> buf, lam((size_t)d * Mp, 1.0), amp(Mp, 0.0);
    pack_points(buf, Xs, d, M, Mp, g->discrete.empty() ? nullptr : g->discrete.data());
    for (int j = 0; j < M; ++j) {
        for (int k = 0; k < d; ++k) {
            const double v = lam_Xs[(size_t)j * d + k];
            if (!(v > 0.0) || !std::isfinite(v)) return fail(BOSS_E_INVALID, "lengthscales must be finite and > 0");
            lam[(size_t)k * Mp + j] = v;
        }
        if (!(amp_Xs[j] >= 0.0) || !std::isfinite(amp_Xs[j])) return fail(BOSS_E_INVALID, "amplitudes must be finite and >= 0");
        amp[j] = amp_Xs[j];
    }
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    hipStream_t s = c->stream;
    int rc = ws_reserve(c->craw, sizeof(double) * ((size_t)2 * d * Mp + Mp));
    if (rc) return rc;
    boss_cand cd;
    cd.ctx = c;
    cd.d = d;
    cd.M = M;
    cd.Mp = Mp;
    cd.Craw = (double*)c->craw.p;
    double* clam = cd.Craw + (size_t)d * Mp;
    double* camp = clam + (size_t)d * Mp;
    HIPCHK(hipMemcpyAsync(cd.Craw, buf.data(), sizeof(double) * d * Mp, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(clam, lam.data(), sizeof(double) * d * Mp, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(camp, amp.data(), sizeof(double) * Mp, hipMemcpyHostToDevice, s));
    rc = ws_reserve(c->pred, sizeof(double) * (3 * (size_t)M + 2));   // mu | var | mean | bad
    if (rc) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    double* dev = (double*)c->pred.p;
    double *dmu = dev, *dvar = dev + M, *dmean = dev + 2 * (size_t)M;
    unsigned long long* dbad = (unsigned long long*)(dev + 3 * (size_t)M);
    if (mean_Xs) (void)hipMemcpyAsync(dmean, mean_Xs, sizeof(double) * M, hipMemcpyHostToDevice, s);
    (void)hipMemsetAsync(dbad, 0xff, sizeof(unsigned long long), s);
    rc = predict_enqueue(g, &cd, mean_Xs ? dmean : nullptr, dmu, dvar, false, clam, camp);
    if (rc) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    hipLaunchKernelGGL(clip_var_kernel, dim3((M + 255) / 256), dim3(256), 0, s, dvar, M, dbad);
    unsigned long long bad = 0;
    (void)hipMemcpyAsync(mu, dmu, sizeof(double) * M, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(var, dvar, sizeof(double) * M, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(&bad, dbad, sizeof bad, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e));
    if (bad != ~0ULL) {
        if (bad_index) *bad_index = (long)bad;
        char msg[160];
        std::snprintf(msg, sizeof msg, "The posterior GP predicted variance %g but only values above -1e-08 are tolerated. (DomainError)", var[bad]);
        return fail(BOSS_E_NEG_VAR, msg);
    }
    return BOSS_OK;
}

static int ei_params(Ctx* c, hipStream_t s, int P, const double* fit_coefs, const double* y_max, int has_best, double best,
                     EiPar* par, double* dcoef, double* dymax);

// SURVEY §8f3.  Enqueue μ, σ² (unclipped) and ∇μ, ∇σ² of one posterior at resident candidates:
// forward substitution (prediction kernel, 32-wide V slabs), adjoint substitution in place, gradient
// accumulation.  The transposed factor, the transposed 256×256 inverses and a = L⁻ᵀz are built once
// per factorisation.  mean_s_dev / mean_grad_dev may be null.
static int grad_enqueue(boss_gp* g, const boss_cand* cd, const double* mean_s_dev, const double* mean_grad_dev, double* mu,
                        double* var, double* dmu, double* dvar) {
    Ctx* c = g->ctx;
    hipStream_t s = c->stream;
    const int d = g->d, Np = g->Np, M = cd->M;
    const size_t glds = sizeof(double) * ((size_t)d * GRAD_CHUNK + GRAD_CHUNK + 8 * 2 * (GRAD_MAX_D + 1) * 32);
    if (glds > 150 * 1024) return fail(BOSS_E_INVALID, "x_dim too large for the gradient kernel's LDS staging");
    if (!g->LT) {
        if (dev_malloc((void**)&g->LT, sizeof(double) * (size_t)g->ld * Np) != hipSuccess ||
            dev_malloc((void**)&g->DT2, sizeof(double) * (size_t)Np * PRED_RB) != hipSuccess ||
            dev_malloc((void**)&g->avec, sizeof(double) * (size_t)Np * 2) != hipSuccess) {
            if (g->LT) (void)hipFree(g->LT);
            if (g->DT2) (void)hipFree(g->DT2);
            g->LT = g->DT2 = g->avec = nullptr;
            (void)hipGetLastError();
            return fail(BOSS_E_ALLOC, "device allocation failed (transposed factor)");
        }
        g->have_lt = false;
    }
    int rc = predict_enqueue(g, cd, mean_s_dev, mu, var, true);    // V slabs (32 wide) + scaled candidates
    if (rc) return rc;
    if (!g->have_lt) {                                      // once per factorisation
        hipLaunchKernelGGL(transpose_kernel, dim3(Np / 64, Np / 64, 1), dim3(256), 0, s, (const double*)g->A, g->ld, (size_t)0,
                           g->LT, g->ld, (size_t)0, Np);            // same (non power-of-two) leading dimension as the factor
        hipLaunchKernelGGL(transpose_kernel, dim3(PRED_RB / 64, PRED_RB / 64, Np / PRED_RB), dim3(256), 0, s,
                           (const double*)g->Dinv2, PRED_RB, (size_t)PRED_RB * PRED_RB, g->DT2, PRED_RB,
                           (size_t)PRED_RB * PRED_RB, PRED_RB);
        // a = L⁻ᵀ z: 256-row steps from the last to the first (GEMV partials live in the second half of avec's buffer)
        const int nb = Np / PRED_RB;
        double* partial = g->avec + Np;                      // [<= nb-1][256] fits: (nb-1)*256 < Np
        for (int ib = nb - 1; ib >= 0; --ib) {
            const int nch = nb - 1 - ib;
            if (nch > 0)
                hipLaunchKernelGGL(bt_gemv_partial_kernel, dim3(nch), dim3(256), 0, s, (const double*)g->LT, g->ld, ib,
                                   (const double*)g->avec, partial);
            hipLaunchKernelGGL(bt_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)g->A, g->ld, Np, g->N, ib, nch,
                               (const double*)partial, (const double*)g->DT2, g->avec);
        }
        g->have_lt = true;
    }
    typedef PredG32 G;
    const int tiles = (M + 31) / 32;
    double* slabs = (double*)c->vscratch.p;
    static const bool no_few = getenv("BOSS_NO_FEW") && atoi(getenv("BOSS_NO_FEW"));
    static const int few_max_tiles = getenv("BOSS_FEW_MAX_TILES") ? atoi(getenv("BOSS_FEW_MAX_TILES")) : 128;
    if (tiles <= few_max_tiles && tiles <= invgemm_max_tiles() && Np >= 4 * PRED_RB && !no_few && g->have_winv) {
        // both inverse factors are resident (repeated calls on this factorisation): W = L⁻ᵀV as one GEMM
        typedef GemmDirect<4, 1, 2, 2, 8> GU;
        double* Wsl = slabs + (size_t)tiles * 32 * Np;
        hipLaunchKernelGGL(inv_bwd_kernel<GU>, dim3(tiles, Np / BLK), dim3(GU::NTHREADS), 0, s, (const double*)g->Winv, g->ld, Np,
                           (const double*)slabs, Wsl);
        slabs = Wsl;
    } else if (tiles <= few_max_tiles && Np >= 4 * PRED_RB && !no_few) {
        // few candidates: the adjoint substitution step by step across the chip (see few_back_* kernels)
        typedef GemmDirect<4, 1, 2, 2, 8> GU;
        for (int ib = Np / PRED_RB - 1; ib >= 0; --ib) {
            hipLaunchKernelGGL(few_back_finish_kernel<G>, dim3(tiles), dim3(G::NTHREADS), PredictLds<G>::BYTES, s,
                               (const double*)g->DT2, Np, ib, slabs);
            if (ib > 0)
                hipLaunchKernelGGL(few_back_update_kernel<GU>, dim3(tiles, ib * (PRED_RB / BLK)), dim3(GU::NTHREADS), 0, s,
                                   (const double*)g->LT, g->ld, Np, ib, slabs);
        }
    } else {
        hipLaunchKernelGGL(backsolve_kernel<G>, dim3(tiles), dim3(G::NTHREADS), PredictLds<G>::BYTES, s, (const double*)g->LT,
                           g->ld, Np, (const double*)g->DT2, slabs);
    }
    // few tiles: split the rows of every tile over several workgroups (one workgroup per tile would walk all N rows alone)
    int rsplit = 1;
    if (d <= GRAD_MAX_D && tiles < 128) {
        rsplit = std::min(32, std::max(1, 512 / tiles));
        rsplit = std::min(rsplit, (g->N + GRAD_CHUNK - 1) / GRAD_CHUNK);
    }
    double* part = nullptr;
    if (rsplit > 1) {
        rc = ws_reserve(c->few, sizeof(double) * (size_t)tiles * std::max((size_t)Np * 32 + 64, (size_t)rsplit * 2 * (GRAD_MAX_D + 1) * 32));
        if (rc) return rc;
        part = (double*)c->few.p;                          // the forward pass's residuals are dead by now
    }
    hipLaunchKernelGGL(grad_accum_kernel, dim3(tiles, rsplit), dim3(256), glds, s, (const double*)slabs, (const double*)g->avec, Np,
                       g->N, (const double*)g->Xsc, (const double*)c->csc.p, d, cd->Mp, M, g->kernel, g->amp2,
                       (const double*)g->invlam, (const unsigned char*)g->discrete_dev, mean_grad_dev, dmu, dvar, part);
    if (rsplit > 1)
        hipLaunchKernelGGL(grad_finalize_kernel, dim3(tiles), dim3(32), 0, s, (const double*)part, rsplit, (const double*)c->csc.p, d,
                           cd->Mp, M, (const double*)g->invlam, (const unsigned char*)g->discrete_dev, mean_grad_dev, dmu, dvar);
    HIPCHK(hipGetLastError());
    return BOSS_OK;
}

extern "C" int boss_gp_predict_grad(boss_gp_t* g, int M, const double* Xs, const double* mean_Xs, const double* mean_grad,
                                    double* mu, double* var, double* dmu, double* dvar, long* bad_index) {
    if (!g || !Xs || !mu || !var || !dmu || !dvar) return fail(BOSS_E_INVALID, "NULL argument");
    NOT_FOR_AUG(g);
    if (M < 1) return fail(BOSS_E_INVALID, "M must be >= 1");
    if (bad_index) *bad_index = -1;
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    boss_cand cand_tmp;
    boss_cand* cd = &cand_tmp;
    int rc = temp_cand(c, g->d, M, Xs, cd);
    if (rc) return rc;
    const int d = g->d;
    const size_t dm = (size_t)d * M;
    rc = ws_reserve(c->pred, sizeof(double) * (3 * (size_t)M + 3 * dm + 2));   // mu | var | mean | mean_grad | dmu | dvar | bad
    if (rc) return rc;
    double* dev = (double*)c->pred.p;
    double *dmu_ = dev, *dvar_ = dev + M, *dmean = dev + 2 * (size_t)M, *dmg = dev + 3 * (size_t)M;
    double *dgm = dmg + dm, *dgv = dgm + dm;
    unsigned long long* dbad = (unsigned long long*)(dgv + dm);
    hipStream_t s = c->stream;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
    };
    if (mean_Xs) (void)hipMemcpyAsync(dmean, mean_Xs, sizeof(double) * M, hipMemcpyHostToDevice, s);
    if (mean_grad) (void)hipMemcpyAsync(dmg, mean_grad, sizeof(double) * dm, hipMemcpyHostToDevice, s);
    (void)hipMemsetAsync(dbad, 0xff, sizeof(unsigned long long), s);
    rc = grad_enqueue(g, cd, mean_Xs ? dmean : nullptr, mean_grad ? dmg : nullptr, dmu_, dvar_, dgm, dgv);
    if (rc) {
        cleanup();
        return rc;
    }
    hipLaunchKernelGGL(clip_var_kernel, dim3((M + 255) / 256), dim3(256), 0, s, dvar_, M, dbad);
    unsigned long long bad = 0;
    (void)hipMemcpyAsync(mu, dmu_, sizeof(double) * M, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(var, dvar_, sizeof(double) * M, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(dmu, dgm, sizeof(double) * dm, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(dvar, dgv, sizeof(double) * dm, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(&bad, dbad, sizeof bad, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    hipError_t e2 = hipGetLastError();
    cleanup();
    if (e != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e2));
    if (bad != ~0ULL) {
        if (bad_index) *bad_index = (long)bad;
        char msg[160];
        std::snprintf(msg, sizeof msg, "The posterior GP predicted variance %g but only values above -1e-08 are tolerated. (DomainError)", var[bad]);
        return fail(BOSS_E_NEG_VAR, msg);
    }
    return BOSS_OK;
}

// Acquisition value AND gradient w.r.t. the candidates for one hyper-parameter sample (MAP): the EI x feasibility
// chain rule runs on the device right behind the moment gradients of the P outputs.
extern "C" int boss_acq_ei_grad(int P, boss_gp_t* const* gps, int M, const double* Xs, const double* mean_Xs,
                                const double* mean_grad, const double* fit_coefs, const double* y_max, int has_best,
                                double best, const unsigned char* valid_mask, double* acq_out, double* dacq_out) {
    if (P < 1 || !gps || M < 1 || !Xs || !fit_coefs || !acq_out || !dacq_out) return fail(BOSS_E_INVALID, "bad arguments");
    for (int p = 0; p < P; ++p) {
        if (!gps[p]) return fail(BOSS_E_INVALID, "NULL posterior handle");
        NOT_FOR_AUG(gps[p]);
        if (gps[p]->ctx != gps[0]->ctx || gps[p]->d != gps[0]->d)
            return fail(BOSS_E_INVALID, "all handles must live on one device and share x_dim");
        if (!gps[p]->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    }
    Ctx* c = gps[0]->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    const int d = gps[0]->d;
    boss_cand cand_tmp;
    boss_cand* cd = &cand_tmp;
    int rc = temp_cand(c, d, M, Xs, cd);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const size_t dm = (size_t)d * M;
    // device scratch: mu[P][M] | var[P][M] | mean[P][M] | mean_grad[P][dM] | dmu[P][dM] | dvar[P][dM] | acq[M] | dacq[dM] | coefs[P] | ymax[P] | mask
    const size_t nd = (size_t)3 * P * M + (size_t)3 * P * dm + M + dm + 2 * P;
    rc = ws_reserve(c->pred, sizeof(double) * nd + M);
    if (rc) return rc;
    double* dev = (double*)c->pred.p;
    double* dmu = dev;
    double* dvar = dmu + (size_t)P * M;
    double* dmean = dvar + (size_t)P * M;
    double* dmg = dmean + (size_t)P * M;
    double* dgm = dmg + (size_t)P * dm;
    double* dgv = dgm + (size_t)P * dm;
    double* dacq = dgv + (size_t)P * dm;
    double* ddacq = dacq + M;
    double* dcoef = ddacq + dm;
    double* dymax = dcoef + P;
    unsigned char* dmask = (unsigned char*)(dev + nd);
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
    };
    EiPar par;
    rc = ei_params(c, s, P, fit_coefs, y_max, has_best, best, &par, dcoef, dymax);
    if (rc) {
        cleanup();
        return rc;
    }
    if (mean_Xs) (void)hipMemcpyAsync(dmean, mean_Xs, sizeof(double) * P * M, hipMemcpyHostToDevice, s);
    if (mean_grad) (void)hipMemcpyAsync(dmg, mean_grad, sizeof(double) * P * dm, hipMemcpyHostToDevice, s);
    if (valid_mask) (void)hipMemcpyAsync(dmask, valid_mask, M, hipMemcpyHostToDevice, s);
    if (par.mode != 0) {
        for (int p = 0; p < P; ++p) {
            rc = grad_enqueue(gps[p], cd, mean_Xs ? dmean + (size_t)p * M : nullptr, mean_grad ? dmg + (size_t)p * dm : nullptr,
                              dmu + (size_t)p * M, dvar + (size_t)p * M, dgm + (size_t)p * dm, dgv + (size_t)p * dm);
            if (rc) {
                cleanup();
                return rc;
            }
        }
    }
    hipLaunchKernelGGL(ei_grad_kernel, dim3((M + 127) / 128), dim3(128), 0, s, (const double*)dmu, (const double*)dvar,
                       (const double*)dgm, (const double*)dgv, M, d, par, (const double*)dcoef, (const double*)dymax,
                       valid_mask ? (const unsigned char*)dmask : nullptr, dacq, ddacq);
    (void)hipMemcpyAsync(acq_out, dacq, sizeof(double) * M, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(dacq_out, ddacq, sizeof(double) * dm, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    hipError_t e2 = hipGetLastError();
    cleanup();
    if (e != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e2));
    return BOSS_OK;
}

extern "C" int boss_gp_predict_cov(boss_gp_t* g, int M, const double* Xs, const double* mean_Xs, double* mu,
                                   double* cov, long* bad_index) {
    if (!g || !Xs || !mu || !cov) return fail(BOSS_E_INVALID, "NULL argument");
    NOT_FOR_AUG(g);
    if (M < 1) return fail(BOSS_E_INVALID, "M must be >= 1");
    if (bad_index) *bad_index = -1;
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);                 // the per-device scratch areas are shared
    boss_cand_t* cd = nullptr;
    int rc = boss_cand_create(c->device, g->d, M, Xs, &cd);
    if (rc) return rc;
    double* dev = nullptr;   // mu | var | mean | bad | cov
    if (dev_malloc((void**)&dev, sizeof(double) * (3 * (size_t)M + 2 + (size_t)M * M)) != hipSuccess) {
        boss_cand_free(cd);
        return fail(BOSS_E_ALLOC, "device allocation failed");
    }
    double *dmu = dev, *dvar = dev + M, *dmean = dev + 2 * (size_t)M;
    unsigned long long* dbad = (unsigned long long*)(dev + 3 * (size_t)M);
    double* dcov = dev + 3 * (size_t)M + 2;
    hipStream_t s = c->stream;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
        (void)hipFree(dev);
        boss_cand_free(cd);
    };
    if (mean_Xs) (void)hipMemcpyAsync(dmean, mean_Xs, sizeof(double) * M, hipMemcpyHostToDevice, s);
    (void)hipMemsetAsync(dbad, 0xff, sizeof(unsigned long long), s);
    rc = predict_enqueue(g, cd, mean_Xs ? dmean : nullptr, dmu, dvar, false, nullptr, nullptr, true);   // leaves V in the slab scratch, Csc scaled
    if (rc) {
        cleanup();
        return rc;
    }
    static const bool force64 = getenv("BOSS_FORCE_BN64") && atoi(getenv("BOSS_FORCE_BN64"));
    const int BN = (M >= 64 * 256 || force64) ? 64 : 32;     // must mirror predict_enqueue's choice (V slab layout)
    const int gb = (M + 15) / 16;
    hipLaunchKernelGGL(predict_cov_kernel, dim3(gb, gb), dim3(256), 0, s, (const double*)c->vscratch.p, g->Np, BN,
                       (const double*)c->csc.p, g->d, cd->Mp, M, g->kernel, g->amp2, dcov);
    hipLaunchKernelGGL(clip_cov_diag_kernel, dim3((M + 255) / 256), dim3(256), 0, s, dcov, M, dbad);
    unsigned long long bad = 0;
    (void)hipMemcpyAsync(mu, dmu, sizeof(double) * M, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(cov, dcov, sizeof(double) * (size_t)M * M, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(&bad, dbad, sizeof bad, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e));
    if (bad != ~0ULL) {
        if (bad_index) *bad_index = (long)bad;
        char msg[160];
        std::snprintf(msg, sizeof msg, "The posterior GP predicted variance %g but only values above -1e-08 are tolerated. (DomainError)", cov[bad * (size_t)M + bad]);
        return fail(BOSS_E_NEG_VAR, msg);
    }
    return BOSS_OK;
}

// ------------------------------------------------------------------------------------------
// acquisition
// ------------------------------------------------------------------------------------------
// EI parameters: by value in the kernel arguments for P <= EI_MAXP, else in device arrays.
static int ei_params(Ctx* c, hipStream_t s, int P, const double* fit_coefs, const double* y_max, int has_best, double best,
                     EiPar* par, double* dcoef, double* dymax) {
    par->P = P;
    par->mode = (has_best ? 1 : 0) | (y_max ? 2 : 0);
    par->best = best;
    if (P <= EI_MAXP) {
        for (int p = 0; p < P; ++p) {
            par->coefs[p] = fit_coefs[p];
            par->ymax[p] = y_max ? y_max[p] : std::numeric_limits<double>::infinity();
        }
    } else {
        HIPCHK(hipMemcpyAsync(dcoef, fit_coefs, sizeof(double) * P, hipMemcpyHostToDevice, s));
        if (y_max) HIPCHK(hipMemcpyAsync(dymax, y_max, sizeof(double) * P, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    return BOSS_OK;
}

extern "C" int boss_acq_ei(int P, int S, boss_gp_t* const* gps, const boss_cand_t* cand, const double* mean_Xs,
                           const double* fit_coefs, const double* y_max, int has_best, double best,
                           const unsigned char* valid_mask, double* acq_out, long* argmax_out, double* max_out) {
    if (P < 1 || S < 1 || !gps || !cand || !fit_coefs) return fail(BOSS_E_INVALID, "bad arguments");
    Ctx* c = cand->ctx;
    HIPCHK(hipSetDevice(c->device));
    for (int i = 0; i < P * S; ++i) {
        if (!gps[i]) return fail(BOSS_E_INVALID, "NULL posterior handle");
        if (gps[i]->ctx != c) return fail(BOSS_E_INVALID, "all handles and candidates must live on one device");
        if (!gps[i]->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    }
    std::lock_guard<std::mutex> lk(c->mtx);                 // the per-device scratch areas are shared
    hipStream_t s = c->stream;
    const int M = cand->M;
    // device scratch: mu[P][M] | var[P][M] | acq[M] | mean[P][M] | coefs[P] | ymax[P] | mask
    const size_t nd = (size_t)3 * P * M + M + 2 * P;
    {
        int rc = ws_reserve(c->acq, sizeof(double) * nd + M);      // grow-only: no hipMalloc/hipFree (= device sync) per call
        if (rc) return rc;
    }
    double* dev = (double*)c->acq.p;
    double* dmu = dev;
    double* dvar = dmu + (size_t)P * M;
    double* dacq = dvar + (size_t)P * M;
    double* dmean = dacq + M;
    double* dcoef = dmean + (size_t)P * M;
    double* dymax = dcoef + P;
    unsigned char* dmask = (unsigned char*)(dev + nd);
    double* hres = (double*)c->pinned;                      // written by the epilogue kernel (mapped host memory)
    EiPar par;
    int rc = ei_params(c, s, P, fit_coefs, y_max, has_best, best, &par, dcoef, dymax);
    if (rc) return rc;
    if (valid_mask) (void)hipMemcpyAsync(dmask, valid_mask, M, hipMemcpyHostToDevice, s);
    if (S > 1) (void)hipMemsetAsync(dacq, 0, sizeof(double) * M, s);
    std::vector<double> hmean;
    for (int sm = 0; sm < S; ++sm) {
        if (par.mode != 0) {
            if (mean_Xs) {
                // caller layout p + P*(j + M*s)  →  device [p][j]
                hmean.resize((size_t)P * M);
                for (int j = 0; j < M; ++j)
                    for (int p = 0; p < P; ++p) hmean[(size_t)p * M + j] = mean_Xs[p + (size_t)P * (j + (size_t)M * sm)];
                (void)hipMemcpyAsync(dmean, hmean.data(), sizeof(double) * P * M, hipMemcpyHostToDevice, s);
                (void)hipStreamSynchronize(s);
            }
            for (int p = 0; p < P; ++p) {
                rc = predict_enqueue(gps[p + (size_t)P * sm], cand, mean_Xs ? dmean + (size_t)p * M : nullptr,
                                     dmu + (size_t)p * M, dvar + (size_t)p * M);
                if (rc) {
                    (void)hipStreamSynchronize(s);
                    return rc;
                }
            }
        }
        ProfScope ps(c, sm + 1 < S ? "ei" : "argmax");
        if (sm + 1 < S)
            hipLaunchKernelGGL(ei_accumulate_kernel, dim3((M + 255) / 256), dim3(256), 0, s, dmu, dvar, M, M, par, dcoef, dymax,
                               dacq);
        else                                                 // last sample: EI + BI average + mask + arg-max in one launch
            hipLaunchKernelGGL(acq_epilogue_kernel, dim3(1), dim3(ACQ_EPI_THREADS), 0, s, dmu, dvar, M, M, par, dcoef, dymax,
                               dacq, S > 1 ? 1 : 0, 1.0 / S, valid_mask ? dmask : nullptr, hres);
    }
    if (acq_out) (void)hipMemcpyAsync(acq_out, dacq, sizeof(double) * M, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    hipError_t e2 = hipGetLastError();
    if (e != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e2));
    if (argmax_out) std::memcpy(argmax_out, &hres[1], sizeof(long));
    if (max_out) *max_out = hres[0];
    return BOSS_OK;
}

// EI x feasibility from posterior moments that are already on the host (outputs fitted on other
// ranks and all-gathered, SURVEY 8e "outputs"): the same K8/K9 epilogue kernels as boss_acq_ei.
extern "C" int boss_acq_ei_moments(int device, int P, int S, int M, const double* mu, const double* var,
                                   const double* fit_coefs, const double* y_max, int has_best, double best,
                                   const unsigned char* valid_mask, double* acq_out, long* argmax_out,
                                   double* max_out) {
    if (P < 1 || S < 1 || M < 1 || !mu || !var || !fit_coefs) return fail(BOSS_E_INVALID, "bad arguments");
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    hipStream_t s = c->stream;
    const size_t pm = (size_t)P * M;
    const size_t nd = 2 * pm * S + M + 2 * P;
    rc = ws_reserve(c->acq, sizeof(double) * nd + M);
    if (rc) return rc;
    double* dev = (double*)c->acq.p;
    double* dmu = dev;
    double* dvar = dmu + pm * S;
    double* dacq = dvar + pm * S;
    double* dcoef = dacq + M;
    double* dymax = dcoef + P;
    unsigned char* dmask = (unsigned char*)(dev + nd);
    double* hres = (double*)c->pinned;
    EiPar par;
    rc = ei_params(c, s, P, fit_coefs, y_max, has_best, best, &par, dcoef, dymax);
    if (rc) return rc;
    (void)hipMemcpyAsync(dmu, mu, sizeof(double) * pm * S, hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(dvar, var, sizeof(double) * pm * S, hipMemcpyHostToDevice, s);
    if (valid_mask) (void)hipMemcpyAsync(dmask, valid_mask, M, hipMemcpyHostToDevice, s);
    if (S > 1) (void)hipMemsetAsync(dacq, 0, sizeof(double) * M, s);
    for (int sm = 0; sm + 1 < S; ++sm)
        hipLaunchKernelGGL(ei_accumulate_kernel, dim3((M + 255) / 256), dim3(256), 0, s, dmu + pm * sm, dvar + pm * sm, M, M, par,
                           dcoef, dymax, dacq);
    hipLaunchKernelGGL(acq_epilogue_kernel, dim3(1), dim3(ACQ_EPI_THREADS), 0, s, dmu + pm * (S - 1), dvar + pm * (S - 1), M, M,
                       par, dcoef, dymax, dacq, S > 1 ? 1 : 0, 1.0 / S, valid_mask ? dmask : nullptr, hres);
    if (acq_out) (void)hipMemcpyAsync(acq_out, dacq, sizeof(double) * M, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e));
    if (argmax_out) std::memcpy(argmax_out, &hres[1], sizeof(long));
    if (max_out) *max_out = hres[0];
    return BOSS_OK;
}

// ------------------------------------------------------------------------------------------
// gradient of the log marginal likelihood w.r.t. (lengthscale[d], amplitude, noise_std) at the
// hyper-parameters of the last boss_gp_update
// ------------------------------------------------------------------------------------------
extern "C" int boss_gp_loglike_grad(boss_gp_t* g, double* logpdf_out, double* grad_out) {
    if (!g || !grad_out) return fail(BOSS_E_INVALID, "NULL argument");
    NOT_FOR_AUG(g);
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    if (g->pending) {
        int rc0 = gp_finish(g, nullptr);
        if (rc0) return rc0;
    }
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    const int d = g->d, N = g->N, Np = g->Np, ld = g->ld;
    if (d > LLG_MAX_D) return fail(BOSS_E_INVALID, "x_dim too large for the likelihood-gradient kernel");
    hipStream_t s = c->stream;
    const int nt = Np / 64, ntiles = nt * (nt + 1) / 2, nv = d + 2, nch = 8;
    int rc = ws_reserve(c->lgA, sizeof(double) * (size_t)ld * Np);
    if (rc) return rc;
    rc = ws_reserve(c->lgB, sizeof(double) * (size_t)ld * Np);
    if (rc) return rc;
    rc = ws_reserve(c->lgC, sizeof(double) * ((size_t)nch * Np + (size_t)ntiles * nv + nv));
    if (rc) return rc;
    double* LinvT = (double*)c->lgA.p;
    double* Kinv = (double*)c->lgB.p;
    double* apart = (double*)c->lgC.p;
    double* parts = apart + (size_t)nch * Np;
    double* sums = parts + (size_t)ntiles * nv;
    dinv_join(g);
    if (!g->have_dinv) {
        dinv_launch(g, s);
        g->have_dinv = true;
    }
    typedef PredG32 G;
    static const bool linvt_solve = getenv("BOSS_LINVT_SOLVE") && atoi(getenv("BOSS_LINVT_SOLVE"));   // A/B: the substitution
    if (linvt_solve) {
        hipLaunchKernelGGL(linvt_kernel<G>, dim3(Np / 32), dim3(G::NTHREADS), PredictLds<G>::BYTES, s, (const double*)g->A, ld, Np,
                           (const double*)g->Dinv2, LinvT, ld);
    } else {
        // L⁻ᵀ by recursive doubling (see linv_level_kernel); the lower work matrix lives in the buffer K⁻¹ overwrites afterwards
        linv_enqueue(g, s, LinvT, Kinv);
    }
    hipLaunchKernelGGL(kinv_syrk_kernel<SyrkG>, dim3(g->nblk * (g->nblk + 1) / 2), dim3(256), 0, s, (const double*)LinvT, ld, Np, Kinv,
                       ld);
    hipLaunchKernelGGL(avec_partial_kernel, dim3(Np / 256, nch), dim3(256), 0, s, (const double*)LinvT, ld, Np, N,
                       (const double*)g->A, ld, apart);
    hipLaunchKernelGGL(llgrad_tile_kernel, dim3(ntiles), dim3(256), 0, s, (const double*)g->Xsc, d, N, Np, g->kernel, g->amp2,
                       (const double*)Kinv, ld, (const double*)apart, nch, parts);
    hipLaunchKernelGGL(llgrad_reduce_kernel, dim3(nv), dim3(256), 0, s, (const double*)parts, ntiles, nv, sums);
    std::vector<double> h(nv);
    HIPCHK(hipMemcpyAsync(h.data(), sums, sizeof(double) * nv, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    // host_par still holds the staged parameters of the last update: 1/(λ+1e-8), (α+1e-8)², (σ+1e-8)²
    const double* invlam = g->host_par;
    const double amp = std::sqrt(g->host_par[d]), sig2 = g->host_par[d + 1], sig = std::sqrt(sig2);
    const double zz = g->host_res[1], trK = h[d], aa = h[d + 1];
    for (int m = 0; m < d; ++m) grad_out[m] = -h[m] * invlam[m];                       // −S_m / λ_m
    grad_out[d] = (zz - N - sig2 * (aa - trK)) / amp;
    grad_out[d + 1] = sig * (aa - trK);
    if (logpdf_out) *logpdf_out = -0.5 * (N * 1.8378770664093453 + g->host_res[0] + g->host_res[1]);
    return BOSS_OK;
}

// ------------------------------------------------------------------------------------------
// tracked candidates: resident V slabs, updated in O(N·M) per appended observation
// ------------------------------------------------------------------------------------------
static void track_release(boss_track* t) {
    if (!t) return;
    if (t->ctx) (void)hipSetDevice(t->ctx->device);
    void* ptrs[] = {t->V, t->Csc, t->mu, t->var, t->mean};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete t;
}

// (re)build the whole state with the prediction kernel; caller holds the context lock
static int track_rebuild(boss_track* t, const boss_cand* cd) {
    boss_gp* g = t->gp;
    Ctx* c = t->ctx;
    hipStream_t s = c->stream;
    const int Ncap = g->Np + PRED_RB;
    if (Ncap > t->Ncap) {
        if (t->V) (void)hipFree(t->V);
        t->V = nullptr;
        if (dev_malloc((void**)&t->V, sizeof(double) * (size_t)t->tiles * Ncap * 32) != hipSuccess) {
            (void)hipGetLastError();
            return fail(BOSS_E_ALLOC, "device allocation failed (tracked V slabs)");
        }
        t->Ncap = Ncap;
    }
    int rc = predict_enqueue(g, cd, t->has_mean ? t->mean : nullptr, t->mu, t->var, true);   // 32-wide slabs in the scratch
    if (rc) return rc;
    HIPCHK(hipMemcpy2DAsync(t->V, sizeof(double) * (size_t)t->Ncap * 32, c->vscratch.p, sizeof(double) * (size_t)g->Np * 32,
                            sizeof(double) * (size_t)g->Np * 32, t->tiles, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(t->Csc, c->csc.p, sizeof(double) * (size_t)t->d * t->Mp, hipMemcpyDeviceToDevice, s));
    t->N = g->N;
    t->epoch = g->epoch;
    return BOSS_OK;
}

extern "C" int boss_track_create(boss_gp_t* g, const boss_cand_t* cand, const double* mean_Xs, boss_track_t** out) {
    if (!out) return fail(BOSS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!g || !cand) return fail(BOSS_E_INVALID, "NULL argument");
    NOT_FOR_AUG(g);
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "handle has no valid factorisation");
    if (cand->ctx != g->ctx || cand->d != g->d) return fail(BOSS_E_INVALID, "candidates and posterior must share device and x_dim");
    Ctx* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    boss_track* t = new boss_track();
    t->ctx = c;
    t->gp = g;
    t->d = g->d;
    t->M = cand->M;
    t->Mp = cand->Mp;
    t->tiles = (cand->M + 31) / 32;
    if (dev_malloc((void**)&t->Csc, sizeof(double) * (size_t)t->d * t->Mp) != hipSuccess ||
        dev_malloc((void**)&t->mu, sizeof(double) * t->M) != hipSuccess ||
        dev_malloc((void**)&t->var, sizeof(double) * t->M) != hipSuccess ||
        dev_malloc((void**)&t->mean, sizeof(double) * t->M) != hipSuccess) {
        (void)hipGetLastError();
        track_release(t);
        return fail(BOSS_E_ALLOC, "device allocation failed");
    }
    if (mean_Xs) {
        HIPCHK(hipMemcpyAsync(t->mean, mean_Xs, sizeof(double) * t->M, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        t->has_mean = true;
    }
    int rc = track_rebuild(t, cand);
    if (rc) {
        (void)hipStreamSynchronize(c->stream);
        track_release(t);
        return rc;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = t;
    return BOSS_OK;
}

extern "C" void boss_track_free(boss_track_t* t) {
    if (!t) return;
    if (t->ctx) {
        (void)hipSetDevice(t->ctx->device);
        (void)hipStreamSynchronize(t->ctx->stream);
    }
    track_release(t);
}

// bring the state up to the posterior's current N (enqueue only); caller holds the context lock
static int track_sync_locked(boss_track* t) {
    boss_gp* g = t->gp;
    if (!g->fitted) return fail(BOSS_E_NOT_FITTED, "the tracked posterior has no valid factorisation");
    if (t->epoch != g->epoch)
        return fail(BOSS_E_INVALID, "the tracked posterior was re-fitted with new hyper-parameters: create a new track");
    if (g->N < t->N) return fail(BOSS_E_INVALID, "the tracked posterior shrank");
    if (g->N > t->Ncap) return fail(BOSS_E_INVALID, "tracked state out of capacity: create a new track");
    hipStream_t s = t->ctx->stream;
    for (int N0 = t->N; N0 < g->N; N0 += TRACK_ROWS) {
        const int n = std::min(TRACK_ROWS, g->N - N0);
        hipLaunchKernelGGL(track_append_kernel, dim3(t->tiles), dim3(256), 0, s, (const double*)g->A, g->ld, g->Np, N0, n, t->V,
                           t->Ncap, (const double*)g->Xsc, g->Np, (const double*)t->Csc, t->d, t->Mp, t->M, g->kernel, g->amp2,
                           t->mu, t->var);
    }
    t->N = g->N;
    HIPCHK(hipGetLastError());
    return BOSS_OK;
}

extern "C" int boss_track_sync(boss_track_t* t) {
    if (!t) return fail(BOSS_E_INVALID, "track is NULL");
    Ctx* c = t->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = track_sync_locked(t);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return BOSS_OK;
}

extern "C" int boss_track_moments(boss_track_t* t, int first, int count, double* mu, double* var) {
    if (!t || !mu || !var || first < 0 || count < 1 || first + count > t->M) return fail(BOSS_E_INVALID, "bad arguments");
    Ctx* c = t->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    int rc = track_sync_locked(t);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(mu, t->mu + first, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(var, t->var + first, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BOSS_OK;
}

extern "C" int boss_acq_ei_tracks(int P, int S, boss_track_t* const* tracks, const double* fit_coefs, const double* y_max,
                                  int has_best, double best, const unsigned char* valid_mask, double* acq_out,
                                  long* argmax_out, double* max_out) {
    if (P < 1 || S < 1 || !tracks || !fit_coefs) return fail(BOSS_E_INVALID, "bad arguments");
    for (int i = 0; i < P * S; ++i) {
        if (!tracks[i]) return fail(BOSS_E_INVALID, "NULL track");
        if (tracks[i]->ctx != tracks[0]->ctx || tracks[i]->M != tracks[0]->M)
            return fail(BOSS_E_INVALID, "all tracks must share the device and the candidate set");
    }
    Ctx* c = tracks[0]->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->mtx);
    hipStream_t s = c->stream;
    const int M = tracks[0]->M;
    const size_t nd = (size_t)2 * P * M + M + 2 * P;
    int rc = ws_reserve(c->acq, sizeof(double) * nd + M);
    if (rc) return rc;
    double* dev = (double*)c->acq.p;
    double* dmu = dev;
    double* dvar = dmu + (size_t)P * M;
    double* dacq = dvar + (size_t)P * M;
    double* dcoef = dacq + M;
    double* dymax = dcoef + P;
    unsigned char* dmask = (unsigned char*)(dev + nd);
    double* hres = (double*)c->pinned;
    EiPar par;
    rc = ei_params(c, s, P, fit_coefs, y_max, has_best, best, &par, dcoef, dymax);
    if (rc) return rc;
    if (valid_mask) (void)hipMemcpyAsync(dmask, valid_mask, M, hipMemcpyHostToDevice, s);
    if (S > 1) (void)hipMemsetAsync(dacq, 0, sizeof(double) * M, s);
    for (int sm = 0; sm < S; ++sm) {
        for (int p = 0; p < P; ++p) {
            boss_track* t = tracks[p + (size_t)P * sm];
            rc = track_sync_locked(t);
            if (rc) {
                (void)hipStreamSynchronize(s);
                return rc;
            }
            (void)hipMemcpyAsync(dmu + (size_t)p * M, t->mu, sizeof(double) * M, hipMemcpyDeviceToDevice, s);
            (void)hipMemcpyAsync(dvar + (size_t)p * M, t->var, sizeof(double) * M, hipMemcpyDeviceToDevice, s);
        }
        if (sm + 1 < S)
            hipLaunchKernelGGL(ei_accumulate_kernel, dim3((M + 255) / 256), dim3(256), 0, s, dmu, dvar, M, M, par, dcoef, dymax,
                               dacq);
        else
            hipLaunchKernelGGL(acq_epilogue_kernel, dim3(1), dim3(ACQ_EPI_THREADS), 0, s, dmu, dvar, M, M, par, dcoef, dymax,
                               dacq, S > 1 ? 1 : 0, 1.0 / S, valid_mask ? dmask : nullptr, hres);
    }
    if (acq_out) (void)hipMemcpyAsync(acq_out, dacq, sizeof(double) * M, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(BOSS_E_NO_DEVICE, hipGetErrorString(e));
    if (argmax_out) std::memcpy(argmax_out, &hres[1], sizeof(long));
    if (max_out) *max_out = hres[0];
    return BOSS_OK;
}

// ------------------------------------------------------------------------------------------
// measurement helpers
// ------------------------------------------------------------------------------------------
extern "C" int boss_bench_mfma_f64(int device, int iters, double* tflops_out) {
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    const int blocks = prop.multiProcessorCount;   // one 4-wave workgroup per CU = one wave per SIMD
    double* sink;
    HIPCHK(dev_malloc((void**)&sink, 8));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, c->stream, iters / 8 + 1, sink);   // warm-up
    HIPCHK(hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, c->stream, iters, sink);
    HIPCHK(hipEventRecord(e1, c->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 4.0 * (double)iters * 16.0 * 2048.0;
    if (tflops_out) *tflops_out = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    return BOSS_OK;
}

extern "C" int boss_prof_enable(int device, int on) {
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    c->prof_on = on != 0;
    return BOSS_OK;
}

extern "C" int boss_prof_reset(int device) {
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto& kv : c->prof)
        for (auto& pr : kv.second) {
            c->ev_pool.push_back(pr.first);
            c->ev_pool.push_back(pr.second);
        }
    c->prof.clear();
    return BOSS_OK;
}

extern "C" int boss_prof_get(int device, const char* kernel_class, double* ms_total, long* launches) {
    Ctx* c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    if (!kernel_class) return fail(BOSS_E_INVALID, "kernel_class is NULL");
    HIPCHK(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    long n = 0;
    auto it = c->prof.find(kernel_class);
    if (it != c->prof.end()) {
        for (auto& pr : it->second) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) tot += ms;
            ++n;
        }
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = n;
    return BOSS_OK;
}

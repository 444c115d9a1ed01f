// bosship.hip — C ABI (include/bosship.h) over the HIP kernels.  gfx950 only, no CPU fallback.
#include "../../include/bosship.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gp_kernels.hpp"
#include "potrf.hpp"
#include "chain.hpp"
#include "small_calls.hpp"
#include "rider.hpp"

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
constexpr size_t PINNED_UP_OFF = 4096, PINNED_UP_BYTES = 1024 * 1024, PINNED_DOWN_OFF = PINNED_UP_OFF + PINNED_UP_BYTES,
                 PINNED_DOWN_BYTES = 512 * 1024, PINNED_BYTES = PINNED_DOWN_OFF + PINNED_DOWN_BYTES;

// every device allocation goes through here; BOSS_POISON_ALLOC=1 (tests) fills new memory with NaN bit patterns so that
// reads of never-written memory show up instead of passing on the zeros a fresh process happens to get
static bool release_cached_slabs();                        // (defined with the contexts below)
static std::atomic<int> g_fail_allocs{0};                  // tests (boss_debug_fail_next_alloc): that many first attempts report "out of memory"
static hipError_t dev_malloc(void** p, size_t bytes) {
    static const bool poison = getenv("BOSS_POISON_ALLOC") && atoi(getenv("BOSS_POISON_ALLOC"));
    hipError_t e = hipErrorOutOfMemory;
    if (g_fail_allocs.load(std::memory_order_relaxed) > 0 && g_fail_allocs.fetch_sub(1) > 0) *p = nullptr;
    else e = hipMalloc(p, bytes);
    if (e != hipSuccess) (void)hipGetLastError();            // a failed allocation must not surface later as a stale launch error
    if (e != hipSuccess && release_cached_slabs()) {         // the parked storage of a released set (up to 4.5 GB at BASELINE config 5) may be what stands in the way
        e = hipMalloc(p, bytes);
        if (e != hipSuccess) (void)hipGetLastError();
    }
    if (e == hipSuccess && poison) {
        (void)hipMemset(*p, 0xff, bytes);                    // null stream: the library's streams do not wait for it ...
        (void)hipDeviceSynchronize();                        // ... so finish it before anything else touches the block
    }
    return e;
}

// doubles into the 4 KiB pinned result block (its head: {max, arg-max, sequence number} of boss_acq_ei)
constexpr size_t MULTI_RES_OFF = 384;   // (max, arg-max) of a deferred multi-device shard
constexpr size_t FEW_RES_OFF = 400;     // {μ[4], σ²[4], bad, seq} of a one-to-four-candidates prediction

struct Workspace {
    void* p = nullptr;
    size_t bytes = 0;
};

struct Ctx {
    int device = -1;                            // the physical HIP device (hipSetDevice)
    int logical = -1;                           // the index the ABI knows it by (== device unless BOSS_VIRTUAL_DEVICES is set)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;          // trailing updates of the look-ahead Cholesky
    std::vector<hipEvent_t> ev_panel, ev_rest;  // per-step dependency events (no timing)
    hipEvent_t ev_fork = nullptr, ev_up = nullptr;            // ev_up: last upload out of the pinned staging area
    // Cross-stream release without an event: a chain kernel stores the next sequence number here at its entry, and a
    // one-wave gate kernel in front of the bulk update on the side stream polls the word (sleeping between polls) and
    // exits when it is reached.  An event costs the recording stream 3 µs and reaches the waiting stream after 11-13 µs;
    // hipStreamWaitValue64 is fast (2.6 µs) but its polling wave slows whatever shares its SIMD by 2x
    // (tools/streamwait_probe.hip, profiles/r02_chain_timeline_waitvalue.log).  nullptr (BOSS_NO_GATE=1): events.
    unsigned long long* sig_panel = nullptr;
    unsigned long long* sig_panel_host = nullptr;   // BOSS_DEBUG_WATCH: the signal words live in mapped host memory (sig_panel is its device alias; a watcher thread reads it)
    unsigned long long sig_seq = 0;
    // the resident panel chain (chain.hpp): its own stream, the base of its sequence numbers, and whether it may be used at all
    // (it needs kernels of different streams to run at the same time: off wherever launches are known to be serialised)
    hipStream_t chain_stream = nullptr, strip_stream = nullptr;
    unsigned long long chain_seq = 0, crit_seq = 0, up_seq = 0;
    int n_cus = 256;
    bool chain_ok = false;
    bool test_drop_chain = false;              // BOSS_TEST_DROP_CHAIN=1: the chain kernel is never launched (exercises the fallback)
    unsigned long long acq_seq = 0;            // sequence number of the arg-max result block (boss_acq_ei polls it)
    unsigned long long few_seq = 0;            // ... of the few-candidates prediction's result block (boss_gp_predict)
    // one released slab / pinned block of a batch-fitted set (boss_gp_fit_batch), kept for the next fit of that size: a BI fitter calls it
    // once per BO iteration with the same S and N, and hipMalloc / hipFree of 4.5 GB took 0.2 s on some boxes (3 ms on others)
    std::mutex slab_mtx;
    void* slab_cache = nullptr;
    size_t slab_cache_bytes = 0;
    void* hostblk_cache = nullptr;
    size_t hostblk_cache_bytes = 0;
    unsigned long long* few_done = nullptr;    // device counter of finished workgroups (winv_args_kernel) and its value on the host
    unsigned long long few_done_cnt = 0;
    bool lookahead = true;
    bool prof_on = false;
    std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof;
    std::vector<hipEvent_t> ev_pool;
    Workspace vscratch, csc, pred, acq, batchA, batchX, batchMisc, craw, lgA, lgB, lgC, few, setdesc, setmom;
    // batched likelihood gradients: the gradient passes of consecutive sets rotate over LLG_BANKS streams, each with its
    // own bank of workspaces (bank 0 = the main stream and lgA/lgB/lgC)
    static constexpr int LLG_BANKS = 4;
    Workspace lgbank[LLG_BANKS - 1][3];
    hipStream_t llg_stream[LLG_BANKS - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t llg_join[LLG_BANKS - 1] = {nullptr, nullptr, nullptr};
    //   // lg*: log-likelihood gradient (LinvT, K⁻¹, partials)   // craw: raw candidates of one-shot calls
    void* pinned = nullptr;   // small host-pinned result area
    std::mutex mtx;
};

// the resident chain kernels (chain.hpp) run on streams of their own and nothing is ordered behind them: whoever frees or
// re-lays-out memory they touch waits for them first (they end within microseconds of the update's last kernel)
static void chain_quiesce(Ctx* c) {
    if (c->chain_stream) (void)hipStreamSynchronize(c->chain_stream);
    if (c->strip_stream) (void)hipStreamSynchronize(c->strip_stream);
}

static std::mutex g_ctx_mtx;
static std::map<int, Ctx*> g_ctx;

// Every context's parked set storage on the CURRENT device goes back to the runtime (dev_malloc: an allocation failed).  True when
// something was freed.  Never called with g_ctx_mtx or a slab_mtx held.
static bool release_cached_slabs() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mtx);
        for (auto& kv : g_ctx) {
            Ctx* c = kv.second;
            if (c->device != dev) continue;
            std::lock_guard<std::mutex> sl(c->slab_mtx);
            if (c->slab_cache) blocks.push_back(c->slab_cache);
            c->slab_cache = nullptr;
            c->slab_cache_bytes = 0;
        }
    }
    for (void* b : blocks) (void)hipFree(b);
    return !blocks.empty();
}
extern "C" int boss_debug_fail_next_alloc(int n) {         // (tests: the next n device allocations fail their first attempt)
    g_fail_allocs.store(n > 0 ? n : 0);
    return BOSS_OK;
}
extern "C" int boss_debug_slab_cache_bytes(int device, size_t* bytes_out) {   // (tests: how much released set storage the context keeps)
    std::lock_guard<std::mutex> lk(g_ctx_mtx);
    auto it = g_ctx.find(device);
    if (bytes_out) *bytes_out = 0;
    if (it != g_ctx.end() && bytes_out) {
        std::lock_guard<std::mutex> sl(it->second->slab_mtx);
        *bytes_out = it->second->slab_cache_bytes;
    }
    return BOSS_OK;
}

static void ctx_destroy(Ctx* c) {
    if (!c) return;
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->chain_stream) (void)hipStreamDestroy(c->chain_stream);
    if (c->strip_stream) (void)hipStreamDestroy(c->strip_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->sig_panel && !c->sig_panel_host) (void)hipFree(c->sig_panel);   // (watch mode: the detached watcher thread keeps reading the block — it stays)
    if (c->few_done) (void)hipFree(c->few_done);
    if (c->slab_cache) (void)hipFree(c->slab_cache);
    if (c->hostblk_cache) (void)hipHostFree(c->hostblk_cache);
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    for (int i = 0; i < Ctx::LLG_BANKS - 1; ++i) {
        if (c->llg_stream[i]) (void)hipStreamDestroy(c->llg_stream[i]);
        if (c->llg_join[i]) (void)hipEventDestroy(c->llg_join[i]);
    }
    if (c->pinned) (void)hipHostFree(c->pinned);
    delete c;
}

static int ctx_init(Ctx* c) {
    {
        // Panel chain on the highest-priority stream, bulk trailing updates on the lowest-priority one.
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        // BOSS_CU_MASK=1 (experiment, tools/cumask_check.py): the resident chain and strips on reserved CUs, everything else on the
        // complement.  Mask bit b enables CU position b / 8 of XCD b % 8 (tools/cumask_probe.hip); streams with EQUAL masks share a
        // hardware queue (a kernel behind a resident one on it never starts), so all four masks differ.
        static const int cu_mask = getenv("BOSS_CU_MASK") ? atoi(getenv("BOSS_CU_MASK")) : 0;
        if (cu_mask) {
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, c->device));
            const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
            auto make = [&](hipStream_t* st, int lo, int hi) {   // CUs [lo, hi) in mask-bit order
                std::vector<uint32_t> m(words, 0u);
                for (int b = lo; b < hi && b < ncu; ++b) m[b / 32] |= 1u << (b % 32);
                return hipExtStreamCreateWithCUMask(st, (uint32_t)words, m.data());
            };
            HIPCHK(make(&c->chain_stream, 0, 16));           // two CU positions on every XCD (the chain has two workgroups)
            HIPCHK(make(&c->strip_stream, 16, 32));          // two more (eight strip workgroups)
            HIPCHK(make(&c->own_stream, 32, ncu));
            HIPCHK(make(&c->side_stream, cu_mask >= 2 ? 40 : 32, cu_mask >= 2 ? ncu : ncu - 8));
            c->stream = c->own_stream;
            std::fprintf(stderr, "[bosship] BOSS_CU_MASK=%d: chain on CU bits 0-15, strips on 16-31, main stream on 32-%d, side stream %d-%d\n",
                         cu_mask, ncu - 1, cu_mask >= 2 ? 40 : 32, (cu_mask >= 2 ? ncu : ncu - 8) - 1);
        } else {
        HIPCHK(hipStreamCreateWithPriority(&c->own_stream, hipStreamNonBlocking, greatest));
        c->stream = c->own_stream;
        HIPCHK(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, least));
        HIPCHK(hipStreamCreateWithPriority(&c->chain_stream, hipStreamNonBlocking, greatest));
        HIPCHK(hipStreamCreateWithPriority(&c->strip_stream, hipStreamNonBlocking, greatest));
        }
    }
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
    for (int i = 0; i < Ctx::LLG_BANKS - 1; ++i) {
        HIPCHK(hipStreamCreateWithFlags(&c->llg_stream[i], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->llg_join[i], hipEventDisableTiming));
    }
    c->lookahead = !(getenv("BOSS_NO_LOOKAHEAD") && atoi(getenv("BOSS_NO_LOOKAHEAD")));
    {
        // Gate kernels wait for a kernel of ANOTHER queue: where a tool lets only one kernel run on the device at a time
        // (hardware-counter collection and thread trace under rocprofv3 serialise all queues) a gate would hold the device until
        // its timeout.  Those modes announce themselves in the environment; anything else that serialises the device is caught
        // at the first timeout (gp_finish falls back to events and repeats the update).
        auto set_nonzero = [](const char* name) {
            const char* v = getenv(name);
            return v && *v && std::strcmp(v, "0") != 0;
        };
        static const bool gate_off = set_nonzero("BOSS_NO_GATE") || set_nonzero("ROCPROF_COUNTER_COLLECTION") || getenv("ROCPROF_COUNTERS") ||
                                     getenv("ROCPROF_COUNTER_GROUPS") || set_nonzero("ROCPROF_ADVANCED_THREAD_TRACE") ||
                                     getenv("ROCPROF_ATT_PARAM_SERIALIZE_ALL") || getenv("ROCP_METRICS") || set_nonzero("AMD_SERIALIZE_KERNEL");
        if (!gate_off) {
            if (set_nonzero("BOSS_DEBUG_WATCH")) {
                // diagnostics only: the signal words in mapped host memory (slow to poll, but a host thread can watch them without a
                // single HIP call — whatever the runtime or the device is stuck in) and a watcher thread that prints them when they move
                unsigned long long* hp = nullptr;
                HIPCHK(hipHostMalloc((void**)&hp, SIG_WORDS * sizeof(unsigned long long), hipHostMallocMapped));
                std::memset(hp, 0, SIG_WORDS * sizeof(unsigned long long));
                HIPCHK(hipHostGetDevicePointer((void**)&c->sig_panel, hp, 0));
                c->sig_panel_host = hp;
#ifdef BOSS_DEBUG_WATCH_BUILD
                HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(boss::g_dbg), &c->sig_panel, sizeof(void*)));
#endif
                std::thread([hp]() {
                    unsigned long long last[SIG_WORDS] = {0};
                    for (;;) {
                        std::this_thread::sleep_for(std::chrono::milliseconds(500));
                        unsigned long long now[SIG_WORDS];
                        for (int i = 0; i < SIG_WORDS; ++i) now[i] = ((volatile unsigned long long*)hp)[i];
                        if (std::memcmp(now, last, sizeof now) == 0) continue;
                        std::memcpy(last, now, sizeof now);
                        std::fprintf(stderr, "[bosship watch] up %llu wdone %llu panel %llu crit %llu | gate %llu near %llu bulk %llu | strips",
                                     now[SIGW_UP], now[SIGW_WDONE], now[SIGW_PANEL], now[SIGW_CRIT], now[SIGW_GATE], now[SIGW_NEAR], now[SIGW_BULK]);
                        for (int q = 0; q < 8; ++q) std::fprintf(stderr, " %llu", now[SIGW_PROG + SIGW_PROG_STRIDE * q]);
                        std::fprintf(stderr, " | give-ups: first %llu n %llu last %llu; waits ended by mark %llu, by clock %llu", now[SIGW_DBG], now[SIGW_DBG + 1],
                                     now[SIGW_DBG + 2], now[SIGW_DBG + 4], now[SIGW_DBG + 5]);
                        std::fprintf(stderr, "\n");
                    }
                }).detach();
            } else {
                HIPCHK(hipMalloc((void**)&c->sig_panel, SIG_WORDS * sizeof(unsigned long long)));
                HIPCHK(hipMemset(c->sig_panel, 0, SIG_WORDS * sizeof(unsigned long long)));
            }
        }
        // (gates are enqueued behind their producers and survive launches that execute one at a time; the resident chain cannot)
        c->chain_ok = !gate_off && !set_nonzero("BOSS_NO_CHAIN") && !set_nonzero("HIP_LAUNCH_BLOCKING");
        c->test_drop_chain = set_nonzero("BOSS_TEST_DROP_CHAIN");
        if (c->chain_ok) {
            // can kernels of the side, chain and strip streams run beside one of the main stream?  (see chain_probe_wait_kernel)
            int* okw = nullptr;
            HIPCHK(hipMalloc((void**)&okw, 4 * sizeof(int)));
            HIPCHK(hipMemset(okw, 0, 4 * sizeof(int)));
            hipStream_t probed[4] = {c->own_stream, c->side_stream, c->chain_stream, c->strip_stream};
            for (int i = 0; i < 4; ++i)
                hipLaunchKernelGGL(chain_probe_wait_kernel, dim3(1), dim3(64), 0, probed[i], c->sig_panel + SIGW_NEAR + 1, 4ull, okw + i);
            int okh[4] = {0, 0, 0, 0};
            bool fine = hipDeviceSynchronize() == hipSuccess && hipMemcpy(okh, okw, sizeof okh, hipMemcpyDeviceToHost) == hipSuccess;
            (void)hipFree(okw);
            if (!(fine && okh[0] && okh[1] && okh[2] && okh[3])) c->chain_ok = false;
            if (getenv("BOSS_CHAIN_VERBOSE"))
                std::fprintf(stderr, "[bosship] resident chain: %s (main / side / chain / strip streams side by side: %d %d %d %d)\n",
                             c->chain_ok ? "on" : "off", okh[0], okh[1], okh[2], okh[3]);
        }
    }
    HIPCHK(hipHostMalloc(&c->pinned, PINNED_BYTES, hipHostMallocDefault));   // [0, 4 KiB) epilogue results, then staging (see temp_cand)
    std::memset(c->pinned, 0, PINNED_UP_OFF);                              // (the polled sequence words start from a known value)
    // kernels that need more than 64 KiB of dynamic LDS
    HIPCHK(hipFuncSetAttribute((const void*)potrf_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)potrf_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)potrf_strips_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STRIPS_LDS_BYTES));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) cus = 0;
        if (cus > 0) c->n_cus = cus;
    }
    HIPCHK(hipFuncSetAttribute((const void*)small_fit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMALL_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)small_llgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMALL_LLG_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)small_fit_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMALL_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)small_predict_grad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SPG_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)small_predict_grad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SPG_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)potrf_syrk_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)rider_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)predict_kernel<PredG32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG32>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)predict_kernel_set<PredG32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG32>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)grad_accum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_gemv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_gemv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_gemv_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_args_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_args_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_args_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)winv_args_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)few_back_finish_kernel<PredG32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               PredictLds<PredG32>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)few_w_kernel<PredG32>, hipFuncAttributeMaxDynamicSharedMemorySize, PredictLds<PredG32>::BYTES));
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
    return BOSS_OK;
}

// BOSS_VIRTUAL_DEVICES=n: the ABI shows n devices, device i being a context of its own (streams, workspaces, handles) on physical
// device i mod (number of physical devices).  The one-process multi-device entry points (host_multi.inc) can then be run with
// G > 1 on a one-GPU box — RCCL refuses a communicator with one device twice, so their exchanges take the host path there.
static int virtual_devices() {
    static const int v = getenv("BOSS_VIRTUAL_DEVICES") ? std::max(0, atoi(getenv("BOSS_VIRTUAL_DEVICES"))) : 0;
    return v;
}
static int visible_devices(int* nphys_out = nullptr) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    if (nphys_out) *nphys_out = n;
    return virtual_devices() > 0 ? virtual_devices() : n;
}

static int get_ctx(int device, Ctx** out) {
    std::lock_guard<std::mutex> lk(g_ctx_mtx);
    auto it = g_ctx.find(device);
    if (it != g_ctx.end()) {
        *out = it->second;
        HIPCHK(hipSetDevice(it->second->device));
        return BOSS_OK;
    }
    int nphys = 0;
    const int n = visible_devices(&nphys);
    if (n <= 0) return fail(BOSS_E_NO_DEVICE, "no HIP device visible (bosship has no CPU fallback)");
    if (device < 0 || device >= n) return fail(BOSS_E_INVALID, "device index out of range");
    const int phys = device % nphys;
    HIPCHK(hipSetDevice(phys));
    Ctx* c = new Ctx();
    c->device = phys;
    c->logical = device;
    int rc = ctx_init(c);
    if (rc) {                                                // a half-built context does not stay behind
        ctx_destroy(c);
        return rc;
    }
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
// S posteriors of one output slice fitted as a batch (boss_gp_fit_batch): one device slab and one pinned block for all of them;
// the members are ordinary handles whose arrays are views into the slab (boss_gp::set) — the slab goes when the last member does.
struct boss_gpset {
    Ctx* ctx = nullptr;
    std::atomic<int> refs{0};
    void* slab = nullptr;
    void* host_block = nullptr;
    size_t slab_bytes = 0, host_bytes = 0;
};
static bool slab_cache_on() {
    static const bool v = !(getenv("BOSS_SLAB_CACHE") && atoi(getenv("BOSS_SLAB_CACHE")) == 0) &&
                          !(getenv("BOSS_POISON_ALLOC") && atoi(getenv("BOSS_POISON_ALLOC")));   // (poisoned allocations are for finding reads of never-written memory: always fresh)
    return v;
}
// the last member of a set is gone: its storage goes to the context's one-entry cache (the larger block wins) or back to the device
static void gpset_release_storage(boss_gpset* st) {
    Ctx* c = st->ctx;
    void *slab = st->slab, *hb = st->host_block;
    if (c && slab_cache_on()) {
        std::lock_guard<std::mutex> lk(c->slab_mtx);
        if (slab && st->slab_bytes > c->slab_cache_bytes) {
            std::swap(slab, c->slab_cache);
            c->slab_cache_bytes = st->slab_bytes;
        }
        if (hb && st->host_bytes > c->hostblk_cache_bytes) {
            std::swap(hb, c->hostblk_cache);
            c->hostblk_cache_bytes = st->host_bytes;
        }
    }
    if (slab) (void)hipFree(slab);
    if (hb) (void)hipHostFree(hb);
    st->slab = st->host_block = nullptr;
}

struct boss_gp {
    Ctx* ctx = nullptr;
    boss_gpset* set = nullptr;                 // member of a batch-fitted set: Xraw, y (shared with its siblings), Xsc, mean, A, inv16, Dinv, Dinv2, invlam, scal,
                                               // host_res, host_par are views — never freed one by one, detached (gp_own) before anything grows or rewrites the data
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
    bool dinv_used = false;                    // the current block inverses were used by a prediction (see factor_enqueue)
    bool dinv_used_prev = false;               // its value when the pending update was enqueued (restored if that update is repeated)
    bool gated = false;                        // the pending update was enqueued with gate kernels
    bool chained = false;                      // ... and under the resident panel chain (chain.hpp)
    bool fell_back = false;                    // gp_finish repeated the pending update on a simpler schedule
    double* host_res = nullptr;                // pinned: scal[2], info
    double* host_res_dev = nullptr;            // the same memory through its device address (written by small_fit_kernel / potrf_logdet_kernel)
    bool par_in_args = false;                  // this update's hyper-parameters travel in the first kernel's arguments
    unsigned long long res_seq = 0;            // sequence number the update's last kernel writes behind its results (host_res[3])
    bool res_polled = false;                   // the pending update ends with such a kernel: gp_finish polls instead of synchronising
    double* host_par = nullptr;                // pinned staging: invlam[d], hyp[2]
    hipEvent_t par_ev = nullptr;               // recorded after the staging copies were enqueued
    unsigned long long epoch = 0;              // bumped by every boss_gp_update: tracked candidate states go stale
    // factor_gen: bumped whenever the resident factor changes (update, append of any kind, storage growth, re-factorisation);
    // dinv0_gen: the generation at which block 0 of Dinv was built on its own (the small-N prediction kernels need nothing else)
    unsigned long long factor_gen = 0, dinv0_gen = ~0ull;
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
    // W_i = Dinv2_i · L[i, i-1] (few_w_kernel): built by the first few-candidates call on a factorisation
    double* W2 = nullptr;
    int w2_np = 0;
    unsigned long long w2_gen = ~0ull;
    bool w2_used = false;                      // (since the last factorisation: the next one builds W beside the block inverses)
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
// a batch of nb equally shaped posteriors laid out with constant strides (the batched likelihood gradient); nb = 1: one handle
struct SetBatch {
    int nb = 1;
    size_t sA = 0, sInv16 = 0, sDinv = 0, sDinv2 = 0, sW = 0, sX = 0;
};
static void linv_enqueue(boss_gp* g, hipStream_t s, double* U, double* Lw, const SetBatch& B = SetBatch());
// largest system the entry points accept: element offsets into the factor stay below 2^31 (exercised up to
// 36 864 rows = 10.9 GB by the tests)
constexpr int MAX_ROWS = 46080;

// ------------------------------------------------------------------------------------------
// library
// ------------------------------------------------------------------------------------------
extern "C" const char* boss_version(void) { return "bosship 0.1.0 (gfx950, fp64 MFMA)"; }
extern "C" const char* boss_last_error(void) { return g_last_error.c_str(); }

extern "C" int boss_device_count(int* n_out) {
    const int n = visible_devices();
    hipError_t e = n > 0 ? hipSuccess : hipErrorNoDevice;
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
    std::lock_guard<std::mutex> lk(c->mtx);
    HIPCHK(hipStreamSynchronize(c->stream));                // nothing enqueued on the old stream may still be in flight
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

// the device counter of winv_args_kernel (small_calls.hpp): allocated on first use; false = not available (the callers keep their
// multi-launch paths)
static bool few_done_ensure(Ctx* c, hipStream_t s) {
    if (c->few_done) return true;
    if (hipMalloc((void**)&c->few_done, sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();
        c->few_done = nullptr;
        return false;
    }
    (void)hipMemsetAsync(c->few_done, 0, sizeof(unsigned long long), s);
    c->few_done_cnt = 0;
    return true;
}
static bool few_fused() {
    static const bool v = !(getenv("BOSS_FEW_FUSED") && atoi(getenv("BOSS_FEW_FUSED")) == 0);
    return v;
}

#include "host_factor.inc"
#include "host_append.inc"
#include "host_batch.inc"
#include "host_predict.inc"
#include "host_acq.inc"
#include "host_rider.inc"
#include "host_track.inc"
#include "host_multi.inc"

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

// Host-side walk of potrf_colupd_kernel's workgroup -> strip map (the same colupd_decode the kernel calls): R0/C0/crit of every
// workgroup of a launch.  No device work: tests/test_abi_and_host.py checks each launch form of the schedule for duplicates.
extern "C" int boss_debug_colupd_decode(int G, int k, int m, int ncols, int jfirst, int skipdiag, int xblk, int critical,
                                        int* R0, int* C0, int* crit) {
    if (G < 1 || !R0 || !C0 || !crit) return fail(BOSS_E_INVALID, "bad argument");
    for (int t = 0; t < G; ++t) {
        const ColupdWork w = colupd_decode(t, G, k, m, ncols, jfirst, skipdiag, xblk, critical != 0);
        R0[t] = w.R0;
        C0[t] = w.C0;
        crit[t] = w.crit;
    }
    return BOSS_OK;
}

#ifdef BOSS_CHAIN_TRACE
// (tools/chain_trace3.py) read and reset the resident chain's device timeline
extern "C" int boss_debug_ctrace(unsigned long long* chain /*64*16*/, unsigned long long* colupd /*64*4*/, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (chain && hipMemcpyFromSymbol(chain, HIP_SYMBOL(boss::g_ctrace), sizeof(unsigned long long) * 64 * 16) != hipSuccess) return 1;
    if (colupd && hipMemcpyFromSymbol(colupd, HIP_SYMBOL(boss::g_cutrace), sizeof(unsigned long long) * 64 * 4) != hipSuccess) return 1;
    if (colupd && hipMemcpyFromSymbol(colupd + 64 * 4, HIP_SYMBOL(boss::g_ptrace), sizeof(unsigned long long) * 64 * 32) != hipSuccess) return 1;
    if (reset) {
        static unsigned long long a[64 * 16], b[64 * 4];
        for (int i = 0; i < 64; ++i) {
            for (int j = 0; j < 16; ++j) a[i * 16 + j] = (j == 5) ? ~0ull : 0ull;
            b[i * 4 + 0] = ~0ull;
            b[i * 4 + 1] = b[i * 4 + 2] = b[i * 4 + 3] = 0;
        }
        (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_ctrace), a, sizeof a);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_cutrace), b, sizeof b);
    }
    return 0;
}
extern "C" int boss_debug_rtrace(unsigned long long* rider /*64*4*/, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (rider && hipMemcpyFromSymbol(rider, HIP_SYMBOL(boss::g_rtrace), sizeof(unsigned long long) * 64 * 4) != hipSuccess) return 1;
    if (reset) {
        static unsigned long long a[64 * 4];
        for (int i = 0; i < 64; ++i) {
            a[i * 4 + 0] = ~0ull;
            a[i * 4 + 1] = a[i * 4 + 2] = a[i * 4 + 3] = 0;
        }
        (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_rtrace), a, sizeof a);
    }
    return 0;
}
#endif

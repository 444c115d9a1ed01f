"""Device-side timeline of one N=4096 posterior update with WHOLE-KERNEL extents: every workgroup of the chain kernels
(diag / solve / column update) and of the bulk trailing update stamps s_memrealtime at entry and exit; per (kernel, panel k)
the earliest entry and the latest exit are kept (atomicMin / atomicMax).  Prints absolute start / end per step.
Build: python tools/chain_trace2.py --build   (instrumented copy; the product sources are not modified)."""
import os, shutil, subprocess, sys, tempfile, ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def build():
    tmp = tempfile.mkdtemp(prefix="bosship_t2_")
    os.makedirs(os.path.join(tmp, "boss.jl_amd"))
    shutil.copytree(os.path.join(ROOT, "boss.jl_amd", "csrc"), os.path.join(tmp, "boss.jl_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(tmp, "boss.jl_amd", "csrc", "potrf.hpp")
    s = open(p).read()
    s = s.replace("constexpr int DIAG_TILES = 36;",
                  "__device__ unsigned long long g_t0[8 * 64], g_t1[8 * 64], g_ds[8 * 64], g_dm[8 * 64], g_sm[8 * 64], g_n[8 * 64];\n__device__ unsigned int g_cu[2 * 8 * 4 * 16];\n"
                  "#define TR_BEGIN(code, kk) const int tr_i_ = (code) * 64 + ((kk) & 63); const unsigned long long tr_b_ = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0 && blockIdx.z == 0) "
                  "{ atomicMin(&g_t0[tr_i_], tr_b_); atomicMax(&g_sm[tr_i_], tr_b_); unsigned xc_, hw_; "
                  "asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xc_)); asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hw_)); "
                  "atomicAdd(&g_cu[((((code) == 4 ? 1 : 0) * 8 + (xc_ & 7)) * 4 + ((hw_ >> 13) & 3)) * 16 + ((hw_ >> 8) & 15)], 1u); }\n"
                  "#define TR_END() if (threadIdx.x == 0 && blockIdx.z == 0) { const unsigned long long tr_e_ = __builtin_amdgcn_s_memrealtime(); atomicMax(&g_t1[tr_i_], tr_e_); "
                  "atomicAdd(&g_ds[tr_i_], tr_e_ - tr_b_); atomicMax(&g_dm[tr_i_], tr_e_ - tr_b_); atomicAdd(&g_n[tr_i_], 1ull); }\n"
                  "constexpr int DIAG_TILES = 36;", 1)
    s = s.replace("    extern __shared__ double smem[];\n    const int tid = threadIdx.x;", "    TR_BEGIN(1, k);\n    extern __shared__ double smem[];\n    const int tid = threadIdx.x;", 1)
    i = s.index("    // (the tiles below the diagonal went out panel by panel from wave 12")
    j = s.index("\n}\n", i)
    s = s[:j] + "\n    __syncthreads();\n    TR_END();" + s[j:]
    for code, head in ((2, "__global__ __launch_bounds__(TRSM_THREADS) void potrf_trsm_kernel("),
                       ("(gridDim.x == 8) ? 5 : 2", "__global__ __launch_bounds__(64) void potrf_trsm_sync_kernel("),
                       (5, "__global__ __launch_bounds__(64) void potrf_follow_kernel("),
                       (3, "__global__ __launch_bounds__(256, 2) void potrf_colupd_kernel("),
                       ("(t0 == 0) ? 6 : 3", "__global__ __launch_bounds__(256, 2) void potrf_colupd_part_kernel("),
                       (6, "__global__ __launch_bounds__(256) void potrf_diagupd_kernel("),
                       (4, "__global__ __launch_bounds__(256, 2) void potrf_syrk_kernel(")):
        if head not in s:
            continue
        i = s.index(head)
        kpos = s.index("{\n", i) + 2
        s = s[:kpos] + f"    TR_BEGIN({code}, k);\n" + s[kpos:]
        j = s.index("\n}\n", kpos)
        s = s[:j] + ("\n    __syncthreads();" if "trsm" in head else "") + "\n    TR_END();" + s[j:]
    open(p, "w").write(s)
    p = os.path.join(tmp, "boss.jl_amd", "csrc", "bosship.hip")
    s = open(p).read() + '''
extern "C" int boss_debug_cus(unsigned int* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(boss::g_cu), 4 * 2 * 8 * 4 * 16) == hipSuccess ? 0 : 1; }
extern "C" int boss_debug_wgstats(unsigned long long* ds, unsigned long long* dm, unsigned long long* sm, unsigned long long* n) {
    if (hipMemcpyFromSymbol(ds, HIP_SYMBOL(boss::g_ds), 8 * 8 * 64) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(dm, HIP_SYMBOL(boss::g_dm), 8 * 8 * 64) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(sm, HIP_SYMBOL(boss::g_sm), 8 * 8 * 64) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(boss::g_n), 8 * 8 * 64) != hipSuccess) return 1;
    return 0;
}
extern "C" int boss_debug_trace2(unsigned long long* t0, unsigned long long* t1, int reset) {
    if (hipMemcpyFromSymbol(t0, HIP_SYMBOL(boss::g_t0), 8 * 8 * 64) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(t1, HIP_SYMBOL(boss::g_t1), 8 * 8 * 64) != hipSuccess) return 1;
    if (reset) {
        static unsigned long long ones[8 * 64], zeros[8 * 64];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_ds), zeros, sizeof zeros);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_dm), zeros, sizeof zeros);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_sm), zeros, sizeof zeros);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_n), zeros, sizeof zeros);
        for (int i = 0; i < 8 * 64; ++i) { ones[i] = ~0ull; zeros[i] = 0; }
        (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_t0), ones, sizeof ones);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(boss::g_t1), zeros, sizeof zeros);
    }
    return 0;
}
'''
    open(p, "w").write(s)
    out = os.path.join(ROOT, "tools", "libbosship_t2.so")
    import __graft_entry__ as entry
    extra = os.environ.get("BOSS_EXTRA_FLAGS", "").split()
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + entry.HIPCC_FLAGS + extra + ["-o", out, p])
    shutil.rmtree(tmp)
    return out


if __name__ == "__main__":
    if "--build" in sys.argv:
        print("built", build())
        sys.exit(0)
    from boss_jl_amd import api
    lib = api.load_library(os.path.join(ROOT, "tools", os.environ.get("BOSS_T2_LIB", "libbosship_t2.so")))
    lib.boss_debug_trace2.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int]
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (8, N)); y = np.sin(X).sum(0)
    g = api.GP(X, y, "matern52")
    def upd():
        try:
            g.update(np.full(8, .5), 1.0, 0.05)
        except api.BossError as e:          # timing-experiment builds produce garbage factors
            print("update reported:", e)
    for _ in range(3):
        upd()
    t0 = (C.c_ulonglong * 512)(); t1 = (C.c_ulonglong * 512)()
    lib.boss_debug_trace2(t0, t1, 1)
    upd()
    api.device_sync(0)
    lib.boss_debug_trace2(t0, t1, 0)
    U = C.c_ulonglong * 512
    ds, dm, sm, nn = U(), U(), U(), U()
    lib.boss_debug_wgstats.argtypes = [C.POINTER(C.c_ulonglong)] * 4
    lib.boss_debug_wgstats(ds, dm, sm, nn)
    ds, dm, sm, nn = (np.array(v[:], dtype=np.float64).reshape(8, 64) for v in (ds, dm, sm, nn))
    a0 = np.array(t0[:], dtype=np.float64).reshape(8, 64); a1 = np.array(t1[:], dtype=np.float64).reshape(8, 64)
    valid = a1 > 0
    base = a0[valid].min()
    names = {1: "diag", 5: "follower", 6: "update head", 2: "solve", 3: "colupd", 4: "bulk"}
    print(f"N={N}; times in µs from the first chain kernel's entry; whole-kernel extents (earliest workgroup entry .. latest exit)")
    print(" k | " + " | ".join(f"{names[c_]:>16s}" for c_ in (1, 5, 6, 2, 3, 4)))
    nblk = (N + 255) // 256 * 2
    for k in range(nblk):
        row = [f"{k:2d}"]
        for code in (1, 5, 6, 2, 3, 4):
            if valid[code, k]:
                row.append(f"{(a0[code, k]-base)/100:7.1f}..{(a1[code, k]-base)/100:7.1f}")
            else:
                row.append(" " * 16)
        print(" | ".join(row))
    print(f"span {(a1[valid].max()-base)/100:.1f} µs")
    print("per-workgroup statistics (µs): n workgroups, mean / max duration, latest workgroup entry after the kernel's first")
    for k in range(nblk):
        row = [f"{k:2d}"]
        for code in (5, 6, 2, 3, 4):
            if valid[code, k] and nn[code, k] > 0:
                row.append(f"{names[code]:>11s} n={int(nn[code, k]):4d} mean {ds[code, k]/nn[code, k]/100:5.1f} max {dm[code, k]/100:5.1f} last entry +{(sm[code, k]-a0[code, k])/100:5.1f}")
        print(" | ".join(row))
    cu = (C.c_uint * (2 * 8 * 4 * 16))()
    lib.boss_debug_cus.argtypes = [C.POINTER(C.c_uint)]
    lib.boss_debug_cus(cu)
    cu = np.array(cu[:]).reshape(2, 8, 4, 16)
    for kind, nm in ((0, "chain kernels"), (1, "bulk")):
        used = cu[kind] > 0
        print(f"{nm}: workgroups ran on {int(used.sum())} distinct CUs; per XCD: {[int(used[x].sum()) for x in range(8)]}")
    only_chain = (cu[0] > 0) & (cu[1] == 0)
    print("CUs that ran chain workgroups but never a bulk workgroup:", int(only_chain.sum()), "per XCD", [int(only_chain[x].sum()) for x in range(8)])
    print("chain workgroups on those CUs:", int(cu[0][only_chain].sum()), "of", int(cu[0].sum()))

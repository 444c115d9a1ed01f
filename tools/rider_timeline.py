"""Device timeline of boss_gp_update_acq (csrc/rider.hpp) beside the resident chain: builds the library with -DBOSS_CHAIN_TRACE into
tools/libbosship_t3.so (python tools/rider_timeline.py --build, on the CPU box) and prints per block k, in µs from the first stamp:
chain: C start / C end; main stream: panel solve entry / exit, column update entry / exit; rider step k: start, wait passed, solve end,
accumulate end."""
import os, sys, subprocess, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("BOSS_TRACE_LIB", os.path.join(ROOT, "tools", "libbosship_t3.so"))


def build():
    import __graft_entry__ as entry
    src = os.path.join(ROOT, "boss.jl_amd", "csrc", "bosship.hip")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + entry.HIPCC_FLAGS + ["-DBOSS_CHAIN_TRACE", "-o", LIB, src])


def main():
    import numpy as np
    from boss_jl_amd import api
    api.load_library(LIB)
    lib = C.CDLL(LIB)
    N, M = 4096, int(os.environ.get("M", 1024))
    rng = np.random.default_rng(1)
    d = 8
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    Xs = np.random.default_rng(2).uniform(0, 1, (d, M))
    g = api.GP(X, y, "matern52")
    cand = api.Candidates(Xs)
    lam = np.full(d, 0.5)
    for i in range(4):
        g.update_acq(lam, 1.0, 0.05 + 1e-4 * i, cand, best=float(y.max()))
    ch = (C.c_ulonglong * (64 * 16))()
    cu = (C.c_ulonglong * (64 * 4 + 64 * 32))()
    rt = (C.c_ulonglong * (64 * 4))()
    lib.boss_debug_ctrace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.boss_debug_rtrace.argtypes = [C.c_void_p, C.c_int]
    lib.boss_debug_ctrace(ch, cu, 1)
    lib.boss_debug_rtrace(rt, 1)
    t = time.perf_counter()
    r = g.update_acq(lam, 1.0, 0.051, cand, best=float(y.max()))
    dt = time.perf_counter() - t
    lib.boss_debug_ctrace(ch, cu, 0)
    lib.boss_debug_rtrace(rt, 0)
    ch = np.array(ch, dtype=np.uint64).reshape(64, 16)
    cu = np.array(cu, dtype=np.uint64)[:256].reshape(64, 4)
    rt = np.array(rt, dtype=np.uint64).reshape(64, 4)
    nblk = N // 128
    vals = [int(v) for v in ch[:nblk, [0, 3, 4, 5]].ravel() if 0 < int(v) < 2**63]
    t0 = min(vals)
    us = lambda v: "       " if not (0 < int(v) < 2**63) else f"{(int(v) - t0) / 100.0:7.1f}"
    print(f"N={N} M={M}: update + acquisition {dt * 1e3:.3f} ms (host), fused {r['fused']}; µs from the first stamp (traced build: slower than the shipped one)")
    print(" k | C start   C end | S entry  S exit | U entry  U exit | rider: start  waited  solved  accum'd | solved - C end")
    for k in range(nblk):
        lag = "" if not (0 < int(rt[k][2]) < 2**63 and int(ch[k][4])) else f"{(int(rt[k][2]) - int(ch[k][4])) / 100.0:7.1f}"
        print(f"{k:2d} | {us(ch[k][3])} {us(ch[k][4])} | {us(ch[k][5])} {us(ch[k][9])} | {us(cu[k][0])} {us(cu[k][1])} | {us(rt[k][0])} {us(rt[k][1])} {us(rt[k][2])} {us(rt[k][3])} | {lag}")
    print(f"final kernel: start {us(rt[63][0])} end {us(rt[63][2])}")


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()

"""Device timeline of one posterior update under the resident panel chain (csrc/chain.hpp).  Builds the library with
-DBOSS_CHAIN_TRACE into tools/libbosship_t3.so (python tools/chain_trace3.py --build, on the CPU box) and prints, per step k:
chain workgroup: block ready (wdone) / first follower tiles / follower phase done / factorisation start / end;
follow kernel: earliest entry, late flag of strip 0, end of strip 0, latest exit; column update: earliest entry / latest exit.
Times in µs from the first stamp."""
import os, sys, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("BOSS_TRACE_LIB", os.path.join(ROOT, "tools", "libbosship_t3.so"))   # (an experiment build may be traced instead)


def build():
    import __graft_entry__ as entry
    src = os.path.join(ROOT, "boss.jl_amd", "csrc", "bosship.hip")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + entry.HIPCC_FLAGS + ["-DBOSS_CHAIN_TRACE", "-o", LIB, src])


def main():
    import numpy as np
    from boss_jl_amd import api
    api.load_library(LIB)
    lib = api._lib if hasattr(api, "_lib") else C.CDLL(LIB)
    N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
    rng = np.random.default_rng(1)
    d = 8
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    g = api.GP(X, y, "matern52")
    lam = np.full(d, 0.5)
    def upd(noise):
        try:
            g.update(lam, 1.0, noise)
        except Exception as e:                               # an experiment build that skips work leaves a non-positive-definite matrix
            print("update:", str(e)[:80])
    for _ in range(4):
        upd(0.05)
    ch = (C.c_ulonglong * (64 * 16))()
    cu = (C.c_ulonglong * (64 * 4 + 64 * 32))()
    lib.boss_debug_ctrace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.boss_debug_ctrace(ch, cu, 1)
    import time
    t = time.perf_counter()
    upd(0.051)
    dt = time.perf_counter() - t
    lib.boss_debug_ctrace(ch, cu, 0)
    ch = np.array(ch, dtype=np.uint64).reshape(64, 16)
    cu_all = np.array(cu, dtype=np.uint64)
    cu = cu_all[:256].reshape(64, 4)
    pt = cu_all[256:].reshape(64, 4, 8)
    nblk = (N + 255) // 256 * 2
    vals = [int(v) for v in ch[:nblk, [0, 1, 2, 3, 4, 5, 8, 9, 10]].ravel() if 0 < int(v) < 2**63]
    if not vals:
        print('no stamps: the update did not run on the resident chain (fallback?)', api.last_error() if hasattr(api, 'last_error') else '')
        return
    t0 = min(vals)
    us = lambda v: "       " if not (0 < int(v) < 2**63) else f"{(int(v) - t0) / 100.0:7.1f}"
    print(f"N={N}: update {dt * 1e3:.3f} ms (host); µs from the first stamp")
    print(" k | blk ready  F first  F last   C start  C end  | strip0: start late end | S entry  S exit | U entry  U exit | period")
    prev = None
    for k in range(nblk):
        r = ch[k]
        per = "" if prev is None or not r[4] else f"{(int(r[4]) - prev) / 100.0:6.1f}"
        prev = int(r[4]) if r[4] else prev
        print(f"{k:2d} | {us(r[0])} {us(r[1])} {us(r[2])} {us(r[3])} {us(r[4])} |  {us(r[8])}  {int(r[7])}  {us(r[10])} | {us(r[5])} {us(r[9])} | {us(cu[k][0])} {us(cu[k][1])} | {per}")
    print("boundary detail (µs after the previous block's C end): strips end, F sees last panel, operands landed, tiles written, C start")
    for k in range(max(2, nblk - 6), nblk):
        e = int(ch[k - 1][4])
        g = lambda v: f"{(int(v) - e) / 100.0:6.2f}" if int(v) else "      "
        print(f"  block {k:2d}: {g(ch[k - 1][10])} {g(ch[k][2])} {g(ch[k][11])} {g(ch[k][12])} {g(ch[k][3])}")
    panels(pt, t0, [nblk - 4, nblk - 3])




def panels(pt, t0, blocks):
    print("inside the published diagonal block (µs from the block's first stamp): per panel, pivot chain start / end on wave 0; wave 15 drain start / end")
    for b in blocks:
        r = pt[b]
        base = int(r[0][0])
        if base == 0:
            continue
        f = lambda v: "     " if int(v) == 0 else f"{(int(v) - base) / 100.0:5.1f}"
        print(f"block {b:2d}: " + " | ".join(f"{f(r[0][j])} {f(r[1][j])} d{f(r[2][j])} {f(r[3][j])}" for j in range(8)))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()

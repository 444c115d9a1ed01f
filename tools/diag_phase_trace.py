"""Where do the 22-25 µs of one potrf_diag_kernel launch go?  Builds an instrumented copy of the library whose diagonal-block
kernel stamps s_memtime on the pivot-chain wave (wave 0) at: loop entry, end of the 16-column pivot chain, after barrier 1,
end of phase B, after barrier 2 — and on update wave 1 at its arrival at barrier 1 — for every 16-column sub-step of every
panel, then prints the per-phase averages of one N=4096 update.  The product sources are not modified."""
import os, shutil, subprocess, sys, tempfile, ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def build():
    tmp = tempfile.mkdtemp(prefix="bosship_ph_")
    os.makedirs(os.path.join(tmp, "boss.jl_amd"))
    shutil.copytree(os.path.join(ROOT, "boss.jl_amd", "csrc"), os.path.join(tmp, "boss.jl_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(tmp, "boss.jl_amd", "csrc", "potrf.hpp")
    s = open(p).read()
    s = s.replace("constexpr int DIAG_TILES = 36;",
                  "__device__ unsigned long long g_ph[64 * 8 * 8];\n"
                  "#define PH(slot) do { if (lane == 0 && blockIdx.z == 0 && col0 / 128 < 64) { unsigned long long t_; "
                  "asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_) :: \"memory\"); g_ph[(col0 / 128 * 8 + jb) * 8 + (slot)] = t_; } } while (0)\n"
                  "constexpr int DIAG_TILES = 36;", 1)
    s = s.replace("        if (wave == 0) {\n            v4d E;\n            chol16_unscaled(S, E, ipsel, lane, col0 + jb * 16, fail);",
                  "        if (wave == 0) {\n            v4d E;\n            PH(0);\n            chol16_unscaled(S, E, ipsel, lane, col0 + jb * 16, fail);\n            PH(1);", 1)
    s = s.replace("        lds_barrier();                                // barrier 1: S'/E' published, block fully updated through panel jb-1",
                  "        if (wave == 1) PH(5);\n        lds_barrier();                                // barrier 1\n        if (wave == 0) PH(2);", 1)
    s = s.replace("        lds_barrier();                                // barrier 2: panel jb final",
                  "        if (wave == 0) PH(3);\n        lds_barrier();                                // barrier 2\n        if (wave == 0) PH(4);", 1)
    assert s.count("PH(") == 7, s.count("PH(")
    open(p, "w").write(s)
    p = os.path.join(tmp, "boss.jl_amd", "csrc", "bosship.hip")
    s = open(p).read() + '''
extern "C" int boss_debug_phases(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(boss::g_ph), 8 * 64 * 8 * 8) == hipSuccess ? 0 : 1;
}
'''
    open(p, "w").write(s)
    out = os.path.join(ROOT, "tools", "libbosship_ph.so")
    import __graft_entry__ as entry
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + entry.HIPCC_FLAGS + ["-o", out, p]
    subprocess.check_call(cmd)
    shutil.rmtree(tmp)
    return out


if __name__ == "__main__":
    if "--build" in sys.argv:
        print("built", build())
        sys.exit(0)
    from boss_jl_amd import api
    lib = api.load_library(os.path.join(ROOT, "tools", "libbosship_ph.so"))
    lib.boss_debug_phases.argtypes = [C.POINTER(C.c_ulonglong)]
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (8, 4096)); y = np.sin(X).sum(0)
    g = api.GP(X, y, "matern52")
    for _ in range(3):
        g.update(np.full(8, .5), 1.0, 0.05)
    buf = (C.c_ulonglong * (64 * 8 * 8))()
    assert lib.boss_debug_phases(buf) == 0
    a = np.array(buf[:], dtype=np.float64).reshape(64, 8, 8)[:32]          # [k][jb][slot], s_memtime ticks (100 MHz on gfx950? printed raw)
    d = lambda x, y_: (a[:, :, x] - a[:, :, y_])
    print("ticks are s_memtime units; per 16-column sub-step, mean over panels 2..29")
    sl = slice(2, 30)
    for jb in range(8):
        chain = d(1, 0)[sl, jb].mean()
        w1 = d(2, 1)[sl, jb].mean()
        phb = d(3, 2)[sl, jb].mean() if jb < 7 else float("nan")
        w2 = d(4, 3)[sl, jb].mean() if jb < 7 else float("nan")
        upd_arrive = (a[sl, jb, 5] - a[sl, jb, 0]).mean()
        print(f"jb={jb}: pivot chain {chain:8.0f} | wait at barrier 1 {w1:7.0f} | phase B {phb:7.0f} | wait at barrier 2 {w2:7.0f} | update wave 1 reaches barrier 1 at {upd_arrive:8.0f}")
    tot = (a[sl, 7, 2] - a[sl, 0, 0]).mean()
    print(f"loop total (first stamp to last barrier 1): {tot:.0f} ticks")

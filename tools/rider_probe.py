"""Step k of the rider (csrc/rider.hpp) alone on the idle device: duration of one launch (S and E workgroups, or S alone) by HIP
events, and the rate of the E product.  Settings come from the environment (BOSS_RIDER_CHUNKS, BOSS_RIDER_LDS, BOSS_RIDER_EFIRST).
python tools/rider_probe.py [M ...]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boss_jl_amd import api
if os.environ.get("BOSS_LIB_PATH"):
    api.load_library(os.environ["BOSS_LIB_PATH"])
lib = api.load_library()
N, D = 4096, 8
rng = np.random.default_rng(1)
X = rng.uniform(0, 1, (D, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(D) + 0.05 * rng.standard_normal(N)
g = api.GP(X, y, "matern52")
lam = np.full(D, 0.5)
lib.boss_debug_rider_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for M in [int(v) for v in sys.argv[1:]] or [1024, 2048, 4096]:
    cand = api.Candidates(rng.uniform(0, 1, (D, M)))
    r = g.update_acq(lam, 1.0, 0.05, cand, best=float(y.max()))
    assert r["fused"]
    for k in (4, 8, 16, 24, 30):
        out = C.c_double(0.0)
        res = []
        for s_only in (0, 1):
            rc = lib.boss_debug_rider_replay(g._h, cand._h, k, 50, s_only, C.byref(out))
            assert rc == 0, rc
            res.append(out.value * 1e3)
        fl = 2.0 * M * 128 * (k - 1) * 128
        print(f"M {M:5d} k {k:2d}: step {res[0]:6.1f} us (S alone {res[1]:5.1f} us)  E product {fl / res[0] / 1e6:5.1f} TFLOP/s of the step, MFMA floor {fl / 78.6e6:5.1f} us", flush=True)
    cand.close()

"""A/B timing of the posterior update under environment switches: python tools/update_ab.py  (prints one line per N).
Run once per setting, e.g.  BOSS_SMALL_M=8 python tools/update_ab.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
lib = sys.argv[1] if len(sys.argv) > 1 else None
if lib:
    api.load_library(lib)
rng = np.random.default_rng(0)
for N in (2048, 4096, 8192):
    X = rng.uniform(0, 1, (8, N)); y = np.sin(X).sum(0)
    g = api.GP(X, y, "matern52"); lam = np.full(8, .5)
    for _ in range(2): g.update(lam, 1.0, 0.05)
    ts = []
    for r in range(5):
        t = time.perf_counter()
        for i in range(10): g.update(lam, 1.0, 0.05 + 1e-4 * i)
        ts.append((time.perf_counter() - t) / 10)
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("BOSS_"))
    print(f"[{lib or 'default'} {tag}] N={N}: update min {min(ts)*1e3:.3f} ms  median {sorted(ts)[2]*1e3:.3f} ms", flush=True)
    g.close()

"""One posterior update (and a few repeats) at the given size, timed: python tools/one_update.py N [reps] [--warm]
--warm: first run a small batched likelihood (launches the full-tile trailing-update kernel once outside any chain update)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boss_jl_amd import api
api.load_library()
N = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
rng = np.random.default_rng(N); d = 8
if "--warm" in sys.argv:
    Xw = rng.uniform(0, 1, (d, 600)); yw = rng.standard_normal(600)
    ll, st = api.loglike_batch(Xw, yw, "matern52", np.full((d, 12), 0.5), np.ones(12), np.full(12, 0.1))
    print("warm:", ll[:2], flush=True)
X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
g = api.GP(X, y, "matern52"); lam = np.full(d, 0.5)
for i in range(reps):
    t = time.perf_counter(); lp = g.update(lam, 1.0, 0.05 + 1e-4 * i); dt = time.perf_counter() - t
    print(f"N={N} update {i}: {dt * 1e3:.3f} ms logpdf {lp:.6f}", flush=True)

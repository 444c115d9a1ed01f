import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from boss_jl_amd import api
api.load_library()
rng = np.random.default_rng(1); d, N = 8, 4096
X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0)/np.sqrt(d) + 0.05*rng.standard_normal(N)
lam = np.full(d, 0.5); g = api.GP(X, y, "matern52"); best = float(y.max())
for M in (64, 1024, 2048, 4096):
    cm = api.Candidates(rng.uniform(0, 1, (d, M))); ts = []; res = None
    for i in range(10):
        g.update(lam, 1.0, 0.05 + 1e-4 * (i % 2))
        t = time.perf_counter(); r = api.acq_ei([[g]], cm, [1.0], None, best, want_acq=False); ts.append(time.perf_counter() - t)
        if i % 2 == 0: res = r[1:] if res is None else res; assert r[1:] == res
    print(f"M={M}: first call median {np.median(ts[2:])*1e3:.4f} ms  argmax {res}", flush=True)

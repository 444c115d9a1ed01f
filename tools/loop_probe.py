import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from boss_jl_amd import api
api.load_library()
N = int(sys.argv[1]); d = 8
rng = np.random.default_rng(N)
X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0)/np.sqrt(d) + 0.05*rng.standard_normal(N)
lam = np.full(d, 0.5)
g = api.GP(X, y, "matern52")
ts = []
for i in range(24):
    t = time.perf_counter(); g.update(lam, 1.0, 0.05 + 1e-4*(i & 3)); ts.append((time.perf_counter()-t)*1e3)
print("N", N, " ".join(f"{t:.2f}" for t in ts))

"""First-call few-candidates path at N=4096: M candidates on a fresh factorisation, timed per call (and under rocprofv3 for its kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
api.load_library()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N, d = 4096, 8
rng = np.random.default_rng(1)
X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0)/np.sqrt(d) + 0.05*rng.standard_normal(N)
Xs = rng.uniform(0, 1, (d, M)); lam = np.full(d, 0.5)
g = api.GP(X, y, "matern52"); cand = api.Candidates(Xs); best = float(y.max())
ts = []
for i in range(8):
    g.update(lam, 1.0, 0.05 + 1e-4 * i)
    t = time.perf_counter(); api.acq_ei([[g]], cand, [1.0], None, best, want_acq=False); ts.append((time.perf_counter() - t) * 1e3)
print("M", M, "first-call ms:", " ".join(f"{t:.3f}" for t in ts))

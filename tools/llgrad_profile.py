"""Kernel mix of update + log-likelihood gradient at N=4096 and at small N (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
rng = np.random.default_rng(0)
d = 8
for N, K in ((4096, 6), (100, 50), (20, 50)):
    X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    g = api.GP(X, y, "matern52"); lam = np.full(d, 0.5)
    for _ in range(2):
        g.update(lam, 1.0, 0.05); g.loglike_grad()
    t = time.perf_counter()
    for i in range(K):
        g.update(lam, 1.0, 0.05 + 1e-4 * i); g.loglike_grad()
    print(f"N={N}: update + loglike_grad {(time.perf_counter() - t) / K * 1e3:.3f} ms", flush=True)
    t = time.perf_counter()
    for i in range(K):
        g.update(lam, 1.0, 0.05 + 1e-4 * i)
    print(f"N={N}: update alone {(time.perf_counter() - t) / K * 1e3:.3f} ms", flush=True)
    g.close()

"""Batched likelihood rates (boss_gp_loglike_batch): N=4096 at S=8 / S=32, and BASELINE config 5 (512 x N=1024)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
rng = np.random.default_rng(0)
d = 8
for N, S in ((4096, 8), (4096, 32), (1024, 512)):
    X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    lam = rng.uniform(0.3, 0.8, (d, S)); amp = rng.uniform(0.8, 1.2, S); sig = rng.uniform(0.03, 0.08, S)
    for _ in range(2): api.loglike_batch(X, y, "matern52", lam, amp, sig)
    K = 5
    t = time.perf_counter()
    for _ in range(K): api.loglike_batch(X, y, "matern52", lam, amp, sig)
    dt = (time.perf_counter() - t) / K
    print(f"N={N} S={S}: {dt*1e3:.2f} ms per batch -> {S/dt:.0f} factorisations/s", flush=True)

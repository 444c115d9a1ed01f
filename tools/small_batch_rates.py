"""Batched likelihood (+ gradient) calls at the reference's own sizes (N = 20 / 100, S = 20 / 512)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
rng = np.random.default_rng(0)
for d, N, S in ((2, 20, 20), (2, 20, 512), (4, 100, 20), (4, 100, 512)):
    X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    lam = rng.uniform(0.3, 0.8, (d, S)); amp = rng.uniform(0.8, 1.2, S); sig = rng.uniform(0.03, 0.08, S)
    for want in (False, True):
        for _ in range(3): api.loglike_batch(X, y, "matern52", lam, amp, sig, want_grad=want)
        K = 30
        t = time.perf_counter()
        for _ in range(K): api.loglike_batch(X, y, "matern52", lam, amp, sig, want_grad=want)
        dt = (time.perf_counter() - t) / K
        print(f"d={d} N={N} S={S} grad={want}: {dt*1e3:.3f} ms per call", flush=True)

"""boss_gp_loglike_batch at BASELINE config 5 (512 sets, N = 1024), a few calls — for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boss_jl_amd import api
api.load_library()
rs = np.random.default_rng(4); N, S, d = 1024, 512, 8
X = rs.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0)/np.sqrt(d) + 0.05*rs.standard_normal(N)
lam = np.exp(rs.normal(-0.7, 0.3, (d, S))); amp = np.exp(rs.normal(0, 0.3, S)); sig = np.exp(rs.normal(-3, 0.3, S))
for _ in range(5):
    t = time.perf_counter(); api.loglike_batch(X, y, "matern52", lam, amp, sig); print(f"{(time.perf_counter()-t)*1e3:.3f} ms", flush=True)

"""BASELINE config 5 end to end, a few rounds — for rocprofv3 --kernel-trace --stats: 512 posteriors of one data set (N = 1024) fitted
by ONE boss_gp_fit_batch call, then Expected Improvement averaged over all of them at 8192 candidates (one boss_acq_ei call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boss_jl_amd import api
api.load_library()
rs = np.random.default_rng(4); N, S, d, M = 1024, 512, 8, 8192
X = rs.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0)/np.sqrt(d) + 0.05*rs.standard_normal(N)
lam = np.exp(rs.normal(-0.7, 0.3, (d, S))); amp = np.exp(rs.normal(0, 0.3, S)); sig = np.exp(rs.normal(-3, 0.3, S))
cand = api.Candidates(rs.uniform(0, 1, (d, M))); best = float(y.max())
for i in range(4):
    t = time.perf_counter(); gps, ll, st = api.fit_batch(X, y, "matern52", lam, amp, sig); tf = time.perf_counter() - t
    t = time.perf_counter(); r = api.acq_ei([[g] for g in gps], cand, [1.0], None, best, want_acq=False); ta = time.perf_counter() - t
    print(f"round {i}: fit {tf*1e3:.2f} ms, averaged acquisition {ta*1e3:.2f} ms, argmax {r[1:]}", flush=True)
    for g in gps: g.close()

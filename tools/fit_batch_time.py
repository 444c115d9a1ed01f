"""boss_gp_fit_batch at BASELINE config 5 (512 sets, N = 1024): ms per call, first and repeated (members freed in between)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boss_jl_amd import api
api.load_library()
rs = np.random.default_rng(4); N, S, d = 1024, 512, 8
X = rs.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0)/np.sqrt(d) + 0.05*rs.standard_normal(N)
lam = np.exp(rs.normal(-0.7, 0.3, (d, S))); amp = np.exp(rs.normal(0, 0.3, S)); sig = np.exp(rs.normal(-3, 0.3, S))
ref = None
for i in range(6):
    t = time.perf_counter(); gps, ll, st = api.fit_batch(X, y, "matern52", lam, amp, sig); dt = time.perf_counter() - t
    if ref is None: ref = ll.copy()
    assert np.array_equal(ll, ref) and (np.asarray(st) == 0).all()
    t = time.perf_counter()
    for g in gps: g.close()
    dc = time.perf_counter() - t
    print(f"fit_batch call {i}: {dt*1e3:.3f} ms (closing the 512 members: {dc*1e3:.3f} ms)", flush=True)

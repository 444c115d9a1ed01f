"""Batched factorisations (boss_gp_loglike_batch, boss_gp_fit_batch) at BASELINE config 5 and at N = 4096: ms per call.
python tools/batch_ab.py   (BOSS_BATCH_CHUNK_MB / BOSS_BATCH_STREAMS select the chunking)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boss_jl_amd import api
api.load_library()
rs = np.random.default_rng(4)
def run(N, S, d=8, reps=3, fit=False):
    X = rs.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0)/np.sqrt(d) + 0.05*rs.standard_normal(N)
    lam = np.exp(rs.normal(-0.7, 0.3, (d, S))); amp = np.exp(rs.normal(0, 0.3, S)); sig = np.exp(rs.normal(-3, 0.3, S))
    ll, st = api.loglike_batch(X, y, "matern52", lam, amp, sig)
    t = time.perf_counter()
    for _ in range(reps): ll2, st = api.loglike_batch(X, y, "matern52", lam, amp, sig)
    dt = (time.perf_counter() - t) / reps
    fl = S * (N**3/3 + 2*N**2 + N**2*(3*d+20)/2)
    msg = f"N={N} S={S}: loglike_batch {dt*1e3:.3f} ms ({fl/dt/1e12:.1f} TF, {fl/dt/78.6e12:.3f} of peak) all_pd={bool((st==0).all())} same={bool(np.array_equal(ll, ll2))}"
    if fit:
        gps, l3, st3 = api.fit_batch(X, y, "matern52", lam, amp, sig)
        for g in gps: g.close()
        t = time.perf_counter(); gps, l3, st3 = api.fit_batch(X, y, "matern52", lam, amp, sig); dtf = time.perf_counter() - t
        for g in gps: g.close()
        msg += f" | fit_batch {dtf*1e3:.3f} ms equal_ll={bool(np.array_equal(l3, ll))}"
    print(msg, flush=True)
    return ll
run(1024, 512, fit=True); run(1024, 64); run(2048, 64); run(600, 512); run(4096, 8); run(4096, 32)

"""Latency of the small calls on an N = 4096 posterior: one-candidate predictions and rank-one appends (ms per call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boss_jl_amd import api
from oracle import gp_oracle as O
api.load_library()
rng = np.random.default_rng(1); d, N = 8, 4096
X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0)/np.sqrt(d) + 0.05*rng.standard_normal(N)
Xs = rng.uniform(0, 1, (d, 400)); lam = np.full(d, 0.5)
g = api.GP(X, y, "matern52"); g.reserve(N + 64); g.update(lam, 1.0, 0.05)
g.predict(Xs[:, :1]); g.predict(Xs[:, 1:2])
for M in (1, 2, 4):
    t = time.perf_counter()
    for i in range(200): mu, var = g.predict(Xs[:, i:i + M])
    print(f"predict M={M}: {(time.perf_counter() - t) / 200 * 1e3:.4f} ms per call", flush=True)
post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
mu, var = g.predict(Xs[:, :4]); mo, vo = O.gp_mean_and_var(post, Xs[:, :4])
print("parity few:", np.abs(mu - mo).max(), np.abs(var - vo).max(), flush=True)
xa = rng.uniform(0, 1, (d, 20)); ya = rng.standard_normal(20) * 0.1
g.append(xa[:, 0], ya[0]); g.append(xa[:, 1], ya[1])
ts = []
for i in range(2, 18):
    t = time.perf_counter(); lp = g.append(xa[:, i], ya[i]); ts.append(time.perf_counter() - t)
print(f"rank-one append: median {np.median(ts) * 1e3:.4f} ms, min {min(ts) * 1e3:.4f}", flush=True)
Xa = np.concatenate([X, xa[:, :18]], 1); yy = np.concatenate([y, ya[:18]])
post2 = O.gp_fit(Xa, yy, "matern52", lam, 1.0, 0.05)
mu, var = g.predict(Xs[:, :50]); mo, vo = O.gp_mean_and_var(post2, Xs[:, :50])
print("parity after appends: logpdf", abs(lp - post2.logpdf) / (1 + abs(post2.logpdf)), "mu", np.abs(mu - mo).max(), "var", np.abs(var - vo).max(), flush=True)

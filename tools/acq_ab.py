"""A/B of the acquisition pass (8192 candidates, N=4096) under environment switches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
from oracle import gp_oracle as O
if len(sys.argv) > 1: api.load_library(sys.argv[1])
rng = np.random.default_rng(0)
d, N, M = 8, 4096, 8192
X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
Xs = rng.uniform(0, 1, (d, M))
g = api.GP(X, y, "matern52"); g.update(np.full(d, .5), 1.0, 0.05)
cand = api.Candidates(Xs); best = float(y.max())
for _ in range(3): r = api.acq_ei([[g]], cand, [1.0], None, best, want_acq=False)
ts = []
for rep in range(5):
    t = time.perf_counter()
    for _ in range(10): r = api.acq_ei([[g]], cand, [1.0], None, best, want_acq=False)
    ts.append((time.perf_counter() - t) / 10)
mu, var = g.predict(Xs[:, :64])
post = O.gp_fit(X, y, "matern52", np.full(d, .5), 1.0, 0.05)
mu_o, var_o = O.gp_mean_and_var(post, Xs[:, :64])
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("BOSS_"))
print(f"[{sys.argv[1] if len(sys.argv) > 1 else 'default'} {tag}] acq pass min {min(ts)*1e3:.3f} ms; argmax {r[1]}; |dmu| {np.abs(mu-mu_o).max():.1e} |dvar| {np.abs(var-var_o).max():.1e}", flush=True)

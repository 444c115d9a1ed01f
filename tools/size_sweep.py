"""Update / acquisition / gradient / append timings over N (d=8, M=8192): the BASELINE size is N=4096,
real BO runs pass through all the smaller ones."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
rng = np.random.default_rng(0)
d, M = 8, 8192
Xs = rng.uniform(0, 1, (d, M))
cand = api.Candidates(Xs)
for N in (64, 256, 512, 1024, 2048, 4096, 8192):
    X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    g = api.GP(X, y, "matern52"); lam = np.full(d, 0.5)
    for _ in range(2): g.update(lam, 1.0, 0.05)
    K = 10
    t = time.perf_counter()
    for i in range(K): g.update(lam, 1.0, 0.05 + 1e-4 * i)
    tu = (time.perf_counter() - t) / K
    api.acq_ei([[g]], cand, [1.0], None, 1.0, want_acq=False)
    t = time.perf_counter()
    for _ in range(5): api.acq_ei([[g]], cand, [1.0], None, 1.0, want_acq=False)
    ta = (time.perf_counter() - t) / 5
    g.predict_grad(Xs)                                   # warm-up at the timed size (workspaces grow once)
    t = time.perf_counter()
    for _ in range(3): g.predict_grad(Xs)
    tg = (time.perf_counter() - t) / 3
    g.reserve(N + 16); g.update(lam, 1.0, 0.05)
    t = time.perf_counter()
    for i in range(8): g.append(rng.uniform(0, 1, d), 0.0)
    tp = (time.perf_counter() - t) / 8
    fl = N ** 3 / 3
    print(f"N={N:5d}: update {tu*1e3:7.3f} ms ({fl/tu/1e12:5.2f} TF)  acq(8192) {ta*1e3:7.3f} ms ({M*N*N/ta/1e12:5.1f} TF)  "
          f"grad(8192) {tg*1e3:7.3f} ms  append {tp*1e3:6.3f} ms", flush=True)
    g.close()

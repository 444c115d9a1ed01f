"""Quick GPU sanity + timing script (not a pytest file): python tests/gpu_quick.py [stage...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boss_jl_amd import api  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def problem(d, N, M, seed=1, noise=0.05):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + noise * rng.standard_normal(N)
    Xs = np.random.default_rng(seed + 1).uniform(0, 1, (d, M))
    return X, y, Xs


def parity(d, N, M, kernel="matern52", noise=0.05):
    X, y, Xs = problem(d, N, M, noise=noise)
    lam = np.full(d, 0.5)
    t = time.time()
    post = O.gp_fit(X, y, kernel, lam, 1.0, noise)
    mu_o, var_o = O.gp_mean_and_var(post, Xs, clip=False)
    t_cpu = time.time() - t
    g = api.GP(X, y, kernel)
    lp = g.update(lam, 1.0, noise)
    L, z = g.factor()
    mu, var = g.predict(Xs)
    eL = np.abs(L - post.L).max()
    print(f"d={d} N={N} M={M} {kernel}: logpdf gpu={lp:.10f} cpu={post.logpdf:.10f} "
          f"|dL|={eL:.2e} |dmu|={np.abs(mu - mu_o).max():.2e} |dvar|={np.abs(var - var_o).max():.2e} cpu_s={t_cpu:.2f}",
          flush=True)
    g.close()


def timing(d=8, N=4096, M=8192):
    X, y, Xs = problem(d, N, M)
    lam = np.full(d, 0.5)
    g = api.GP(X, y, "matern52")
    g.update(lam, 1.0, 0.05)
    for _ in range(2):
        g.update(lam, 1.0, 0.05)
    t = time.time()
    K = 10
    for _ in range(K):
        g.update(lam, 1.0, 0.05)
    dt = (time.time() - t) / K
    print(f"update N={N}: {dt * 1e3:.3f} ms  ({1 / dt:.1f} updates/s)", flush=True)
    api.prof_enable(0, True)
    api.prof_reset(0)
    g.update(lam, 1.0, 0.05)
    for k in ("prep", "gram", "potrf_diag", "potrf_trsm", "potrf_syrk", "logdet"):
        ms, n = api.prof_get(0, k)
        print(f"   {k:12s} {ms:8.3f} ms over {n} launches", flush=True)
    api.prof_enable(0, False)
    cand = api.Candidates(Xs)
    b = float(y.max())
    api.acq_ei([[g]], cand, [1.0], None, b, want_acq=False)
    t = time.time()
    for _ in range(3):
        acq, am, mx = api.acq_ei([[g]], cand, [1.0], None, b, want_acq=False)
    dt = (time.time() - t) / 3
    print(f"acq M={M}: {dt * 1e3:.3f} ms ({M / dt / 1e6:.3f} M evals/s) argmax={am} max={mx:.6g}", flush=True)
    api.prof_enable(0, True)
    api.prof_reset(0)
    api.acq_ei([[g]], cand, [1.0], None, b, want_acq=False)
    for k in ("dinv", "predict", "ei", "argmax"):
        ms, n = api.prof_get(0, k)
        print(f"   {k:12s} {ms:8.3f} ms over {n} launches", flush=True)
    api.prof_enable(0, False)


if __name__ == "__main__":
    stages = sys.argv[1:] or ["mfma", "parity", "timing"]
    if "mfma" in stages:
        print("mfma f64 TFLOP/s:", api.bench_mfma_f64(0, 20000), flush=True)
    if "parity" in stages:
        parity(1, 3, 4, "matern32", 0.1)
        parity(2, 100, 33)
        parity(8, 300, 70, "sqexp")
        parity(8, 1000, 257)
        parity(8, 1000, 257, noise=1e-3)
    if "parity_big" in stages:
        parity(8, 4096, 512)
    if "timing" in stages:
        timing()

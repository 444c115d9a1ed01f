"""Single-launch fit for N <= 128: parity against the oracle (all three kernels, prior mean, discrete dimension, N around the
16-column panel boundaries), everything that reads the handle afterwards (prediction, covariance, append, likelihood gradient),
and the time of one update at N = 20 (BASELINE config 1 regime)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
from oracle import gp_oracle as O
api.load_library()
worst = 0.0
for kernel in ("matern32", "matern52", "sqexp"):
    for d, N in ((1, 1), (1, 3), (2, 15), (2, 16), (3, 17), (8, 20), (5, 33), (8, 64), (8, 100), (8, 127), (8, 128), (33, 40)):
        rng = np.random.default_rng(100 * N + d)
        X = rng.uniform(0, 4, (d, N)); y = np.sin(X).sum(0) + 0.05 * rng.standard_normal(N)
        Xs = rng.uniform(0, 4, (d, 9))
        lam = rng.uniform(0.8, 2.0, d); disc = [k == 1 for k in range(d)] if d > 1 else None
        mX = 0.3 + 0.1 * X.sum(0); ms = 0.3 + 0.1 * Xs.sum(0)
        post = O.gp_fit(X, y, kernel, lam, 1.3, 0.07, mean=mX, discrete=disc)
        g = api.GP(X, y, kernel, discrete=disc)
        lp = g.update(lam, 1.3, 0.07, mean_X=mX)
        L, z = g.factor()
        mu, var = g.predict(Xs, ms)
        mu_o, var_o = O.gp_mean_and_var(post, Xs, ms)
        e = max(abs(lp - post.logpdf) / (1 + abs(post.logpdf)), np.abs(L - post.L).max(), np.abs(mu - mu_o).max(), np.abs(var - var_o).max())
        worst = max(worst, e)
        assert e < 1e-9, (kernel, d, N, e)
        # second update with other parameters on the same handle, zero mean
        lp2 = g.update(lam * 1.1, 0.9, 0.1)
        p2 = O.gp_fit(X, y, kernel, lam * 1.1, 0.9, 0.1, discrete=disc)
        assert abs(lp2 - p2.logpdf) <= 1e-9 * (1 + abs(p2.logpdf)), (kernel, d, N)
        if d <= 32:
            _, grad = g.loglike_grad()
            _, go = O.gp_data_loglike_grad(X, y, kernel, lam * 1.1, 0.9, 0.1, discrete=disc)
            assert np.allclose(grad, go, rtol=1e-7, atol=1e-8), (kernel, d, N, grad, go)
        xn = rng.uniform(0, 4, d)
        lp3 = g.append(xn, 0.2)
        p3 = O.gp_fit(np.hstack([X, xn[:, None]]), np.append(y, 0.2), kernel, lam * 1.1, 0.9, 0.1, discrete=disc)
        assert abs(lp3 - p3.logpdf) <= 1e-9 * (1 + abs(p3.logpdf)), (kernel, d, N, lp3, p3.logpdf)
        g.close()
print("parity ok, worst error %.2e" % worst)
# not positive definite
g = api.GP(np.array([[1., 1., 2.]]), np.array([1., 2., 3.]), "sqexp")
try:
    g.update([1.0], 1.0, 0.0); print("expected PosDefException")
except api.PosDefException as e:
    print("not PD reported:", str(e)[:60])
print("recovers:", np.isfinite(g.update([1.0], 1.0, 0.1)))
rs = np.random.default_rng(555)
x1 = rs.uniform(0, 20, (1, 20)); y1 = np.exp(x1[0] / 10) * np.cos(2 * x1[0]) + 0.1 * rs.standard_normal(20)
for N in (128, 20, 64, 128, 20):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (8, N)); y = np.sin(X).sum(0)
    g = api.GP(X, y, "matern52")
    for _ in range(20): g.update(np.full(8, 0.5), 1.0, 0.1)
    t = time.perf_counter()
    for i in range(500): g.update(np.full(8, 0.5), 1.0, 0.1 + 1e-4 * (i % 7))
    dt = (time.perf_counter() - t) / 500
    print(f"N={N}: update {dt*1e6:.1f} us", flush=True)
    g.close()

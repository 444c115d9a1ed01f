"""Per-call cost of the entry points a BO iteration uses, at the reference's own sizes (N = 20, 100)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
rng = np.random.default_rng(0)
def t(f, K=100):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(K): f()
    return (time.perf_counter() - t0) / K * 1e3
for d, N in ((2, 20), (4, 100)):
    X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    g = api.GP(X, y, "matern52"); lam = np.full(d, 0.5)
    g.update(lam, 1.0, 0.05)
    for M in (1, 20, 224, 2000, 8192):
        Xs = rng.uniform(0, 1, (d, M))
        cand = api.Candidates(Xs)
        best = float(y.max())
        row = [f"d={d} N={N} M={M}:"]
        row.append(f"predict {t(lambda: g.predict(Xs)):.3f}")
        row.append(f"predict_grad {t(lambda: g.predict_grad(Xs)):.3f}")
        row.append(f"acq_ei(resident cand) {t(lambda: api.acq_ei([[g]], cand, [1.0], None, best, want_acq=False)):.3f}")
        try:
            row.append(f"acq_ei_grad {t(lambda: api.acq_ei_grad([g], Xs, [1.0], None, best)):.3f}")
        except Exception as e:
            row.append(f"acq_ei_grad n/a ({type(e).__name__})")
        print("  ".join(row) + "  ms", flush=True)
    print(f"d={d} N={N}: update {t(lambda: g.update(lam, 1.0, 0.05)):.3f} ms; update+predict(224) {t(lambda: (g.update(lam, 1.0, 0.05), g.predict(rng.uniform(0, 1, (d, 224))))):.3f} ms", flush=True)
    g.close()

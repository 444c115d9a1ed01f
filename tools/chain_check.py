"""Resident panel chain (csrc/chain.hpp): parity of the factor, z and logpdf against the oracle at sizes that take the chain
schedule, and the time per update (isolated loop) with and without it.  python tools/chain_check.py [N ...]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(Ns, reps):
    from boss_jl_amd import api
    from oracle import gp_oracle as O
    api.load_library()
    for N in Ns:
        rng = np.random.default_rng(N)
        d = 8
        X = rng.uniform(0, 1, (d, N))
        y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
        lam = np.full(d, 0.5)
        g = api.GP(X, y, "matern52")
        lp = g.update(lam, 1.0, 0.05)
        L, z = g.factor()
        msg = ""
        if N <= 4096 and not os.environ.get("SKIP_ORACLE"):
            post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
            import scipy.linalg as sla
            z_o = sla.solve_triangular(post.L, post.delta, lower=True)
            msg = f"|dL|={np.abs(L - post.L).max():.2e} |dz|={np.abs(z - z_o).max():.2e} dlogpdf={abs(lp - post.logpdf) / (1 + abs(post.logpdf)):.2e}"
        for _ in range(3):
            g.update(lam, 1.0, 0.05)
        t = time.perf_counter()
        for i in range(reps):
            g.update(lam, 1.0, 0.05 + 1e-4 * (i & 3))
        dt = (time.perf_counter() - t) / reps
        print(f"N={N} chain={'off' if os.environ.get('BOSS_NO_CHAIN') else 'on'} logpdf={lp:.6f} {msg} update {dt * 1e3:.3f} ms", flush=True)
        g.close()


if __name__ == "__main__":
    Ns = [int(a) for a in sys.argv[1:]] or [384, 640, 1024, 1408, 2048, 3000, 4096]
    run(Ns, 30)

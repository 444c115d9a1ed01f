"""Generates tests/golden/gp_golden.json: small fixed inputs + expected outputs for the hot path.

The reference (Julia) cannot run in this pipeline and holds no golden posterior numbers
(SURVEY §8c), so the expected values are produced by the fp64 oracle (oracle/gp_oracle.py) and,
for every case with N <= 32, cross-checked here against a 50-digit mpmath evaluation before
being written.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as O  # noqa: E402
from oracle import mp_oracle as MP  # noqa: E402

CASES = [
    # name, kernel, d, N, M, noise, amp, with_mean, discrete
    ("ref_fixture_3pt", "matern52", 2, 3, 5, 1e-4, 1.0, True, None),
    ("example_jl_1d", "matern32", 1, 20, 9, 0.1, 1.0, False, None),
    ("m52_d2_n16", "matern52", 2, 16, 6, 0.05, 1.3, False, None),
    ("se_d8_n24", "sqexp", 8, 24, 5, 0.02, 0.8, False, None),
    ("m32_d8_n32_mean", "matern32", 8, 32, 7, 1e-3, 1.1, True, None),
    ("m52_d3_n64_disc", "matern52", 3, 64, 8, 0.05, 1.0, False, [False, True, False]),
    ("m52_d8_n130", "matern52", 8, 130, 33, 0.05, 1.0, False, None),
    ("m52_d8_n256_mean", "matern52", 8, 256, 40, 0.01, 2.0, True, None),
]


def build(name, kernel, d, N, M, noise, amp, with_mean, discrete, seed):
    rng = np.random.default_rng(seed)
    if name == "ref_fixture_3pt":            # test/unit/test/models/gaussian_process.jl:57-73 data
        X = np.array([[2., 5., 8.], [2., 5., 8.]])
        y = X[0].copy()
        Xs = np.array([[1., 2., 3., 4., 100.], [1., 2., 3., 4., 100.]])
        lam = np.array([2.5, 3.5])
    elif name == "example_jl_1d":            # examples/example.jl:14-22,72: y = exp(x/10) cos(2x) + noise on [0,20]
        X = rng.uniform(0, 20, (1, N))
        y = np.exp(X[0] / 10) * np.cos(2 * X[0]) + 0.1 * rng.standard_normal(N)
        Xs = np.linspace(0, 20, M)[None, :]
        lam = np.array([1.5])
    else:
        scale = 4.0 if discrete else 1.0
        X = rng.uniform(0, scale, (d, N))
        y = np.sin(2 * np.pi * X / scale).sum(0) / np.sqrt(d) + noise * rng.standard_normal(N)
        Xs = rng.uniform(0, scale, (d, M))
        lam = rng.uniform(0.3, 0.9, d) * scale
    mX = (1.0 + 0.1 * X.sum(0)) if with_mean else None
    ms = (1.0 + 0.1 * Xs.sum(0)) if with_mean else None
    post = O.gp_fit(X, y, kernel, lam, amp, noise, mean=mX, discrete=discrete)
    mu, var = O.gp_mean_and_var(post, Xs, ms, clip=False)
    if N <= 32 and discrete is None:
        lp, mu_mp, var_mp = MP.posterior(X, y, O.KERNEL_NAMES[kernel], lam, amp, noise, Xs, mX, ms)
        Kc = np.linalg.cond(O.kernelmatrix(post.h, X) + post.h.noise_std ** 2 * np.eye(N))
        tol = max(1e-12, Kc * N * 2.0 ** -53 * 4)
        assert abs(lp - post.logpdf) <= tol * (1 + abs(lp)), (name, lp, post.logpdf)
        assert np.allclose(mu, mu_mp, rtol=0, atol=tol * (1 + np.abs(mu).max())), name
        assert np.allclose(var, var_mp, rtol=0, atol=tol * amp ** 2), name
    best = float(np.max(y))
    acq = O.ei_acquisition([post], Xs, [1.0], [np.inf], best, means_s=None if ms is None else [ms])
    return {
        "name": name, "kernel": kernel, "d": d, "N": N, "M": M,
        "X": X.tolist(), "y": y.tolist(), "Xs": Xs.tolist(), "lengthscale": lam.tolist(), "amplitude": amp,
        "noise_std": noise, "mean_X": None if mX is None else mX.tolist(), "mean_Xs": None if ms is None else ms.tolist(),
        "discrete": discrete, "logpdf": post.logpdf, "mu": mu.tolist(), "var": var.tolist(),
        "z": O.sla.solve_triangular(post.L, post.delta, lower=True).tolist(),
        "L_diag": np.diag(post.L).tolist(),
        # full factor, lower triangle packed column by column (kept to the cases where it stays small)
        "L_packed": None if N > 130 else np.concatenate([post.L[j:, j] for j in range(N)]).tolist(),
        "best": best, "acq_ei": acq.tolist(), "argmax": int(np.argmax(acq)),
    }


if __name__ == "__main__":
    out = [build(*c, seed=100 + i) for i, c in enumerate(CASES)]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gp_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")

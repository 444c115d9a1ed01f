"""A whole Bayesian-optimisation run on the MI355X through the plugin-trio mirror — what `bo!` does with
`GaussianProcess` + `SampleOptMAP` + `OptimizationAM` / `SequentialBatchAM` in the reference (src/bo.jl:30-48):
a constrained 2-output toy problem, hyper-parameters re-fitted every iteration, gradient-based multistart acquisition
maximisation, and at the end one sequential batch of four points.

    python examples/bo_loop.py [iterations]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import boss_jl_amd as B  # noqa: E402
from boss_jl_amd.bo import bo_step, estimate_parameters  # noqa: E402


def blackbox(x):
    """objective y1 (maximise) and constraint output y2 <= 0.5"""
    return np.array([np.sin(3 * x[0]) * np.cos(2 * x[1]) + 0.5 * x[0], x[0] * x[1]])


def main(iters=15, seed=0):
    rng = np.random.default_rng(seed)
    d, P, n0 = 2, 2, 8
    X0 = rng.uniform(0, 2, (d, n0))
    Y0 = np.stack([blackbox(X0[:, j]) for j in range(n0)], axis=1)
    model = B.HipGaussianProcess(lengthscale_priors=[B.MvLogNormal([-0.5, -0.5], [0.6, 0.6])] * P,
                                 amplitude_priors=[B.LogNormal(0.0, 0.7)] * P, noise_std_priors=[B.Dirac(1e-3)] * P)
    problem = B.BossProblem(blackbox, B.Domain((np.zeros(d), np.full(d, 2.0))), B.ExpectedImprovement(B.LinFitness([1.0, 0.0])),
                            model, B.ExperimentData(X0, Y0), y_max=[np.inf, 0.5])
    fitter = B.HipSampleOptMAP(samples=256, multistart=8, iters=15, seed=seed)
    am = B.HipGradientAM(x_prior=lambda r: r.uniform(0, 2, d), multistart=64, iters=20, seed=seed)
    t_fit = t_acq = 0.0
    for it in range(iters):
        t = time.perf_counter()
        fitted = estimate_parameters(problem, fitter)
        t_fit += time.perf_counter() - t
        t = time.perf_counter()
        x, val = bo_step(problem, fitter, am)
        t_acq += time.perf_counter() - t
        feas = problem.data.Y[1] <= 0.5
        best = problem.data.Y[0][feas].max() if feas.any() else float("nan")
        print(f"iter {it + 1:2d}: x = ({x[0]:.3f}, {x[1]:.3f})  acq {val:.3e}  logposterior {fitted.loglike:9.3f}  best feasible y {best:.4f}",
              flush=True)
    estimate_parameters(problem, fitter)
    batch = B.HipSequentialBatchAM(B.HipBatchAM(x_prior=lambda r: r.uniform(0, 2, d), samples=4096, seed=seed), batch_size=4)
    t = time.perf_counter()
    Xb, _ = batch.maximize_acquisition(problem)
    t_b = time.perf_counter() - t
    print("sequential batch of 4:", np.round(np.asarray(Xb).T, 3).tolist(), f"({t_b * 1e3:.1f} ms)")
    print(f"{iters} iterations: fitting {t_fit / iters * 1e3:.1f} ms, acquisition maximisation {t_acq / iters * 1e3:.1f} ms per iteration; "
          f"{problem.data.X.shape[1]} observations")
    feas = problem.data.Y[1] <= 0.5
    assert feas.any() and problem.data.Y[0][feas].max() > Y0[0][Y0[1] <= 0.5].max() - 1e-12
    return problem


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 15)

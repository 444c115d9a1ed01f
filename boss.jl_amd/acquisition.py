"""Host epilogue of the acquisition for the one case that cannot run on the device:
`NonlinFitness` (src/types/fitness.jl) — a user closure per Monte-Carlo sample
(expected_improvement(::NonlinFitness), src/acquisitions/expected_improvement.jl:104-111).
The GPU still produces the posterior mean/variance for every candidate, output and hyper-parameter
sample (boss_gp_predict); only the closure evaluation, the ε-average and the feasibility product
run here.  LinFitness never takes this path (boss_acq_ei evaluates it analytically on the device).
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import numpy as np
from scipy.special import erfc


def sample_eps(y_dim: int, count: int, rng) -> np.ndarray:
    """sample_ϵs (expected_improvement.jl:119): rand(Normal(), (y_dim, sample_count))."""
    return rng.standard_normal((y_dim, count))


def _normcdf(z):
    return 0.5 * erfc(-np.asarray(z, float) / math.sqrt(2.0))


def feas_prob_host(mu: np.ndarray, var: np.ndarray, y_max: Optional[Sequence[float]]) -> np.ndarray:
    """feas_prob (expected_improvement.jl:113-114) on P×M arrays; +Inf constraints count as factor 1
    (src/utils/inf.jl); σ == 0 and y_max == μ gives cdf = 1 (StatsFuns.normcdf)."""
    if y_max is None:
        return np.ones(mu.shape[1])
    fp = np.ones(mu.shape[1])
    for p, ym in enumerate(np.asarray(y_max, float)):
        if np.isinf(ym) and ym > 0:
            continue
        s = np.sqrt(var[p])
        with np.errstate(divide="ignore", invalid="ignore"):
            z = (ym - mu[p]) / s
        z = np.where((s == 0.0) & (ym == mu[p]), np.inf, z)
        fp = fp * _normcdf(z)
    return fp


def ei_nonlin_host(fitness: Callable, mu: np.ndarray, var: np.ndarray, eps: np.ndarray, best: float) -> np.ndarray:
    """expected_improvement(::NonlinFitness, mean, var, ϵ_samples::Matrix, best) (:104-107):
    mean_k max(0, fit(μ + √σ² ⊙ ε_k) − best); a vector ε is the single-sample form (:108-111)."""
    eps = np.asarray(eps, float)
    if eps.ndim == 1:
        eps = eps[:, None]
    sd = np.sqrt(var)
    M = mu.shape[1]
    out = np.zeros(M)
    for j in range(M):
        acc = 0.0
        for k in range(eps.shape[1]):
            acc += max(0.0, float(fitness(mu[:, j] + sd[:, j] * eps[:, k])) - best)
        out[j] = acc / eps.shape[1]
    return out


def acquisition_nonlin(fitness: Callable, mu_var_per_sample, y_max, best, eps: np.ndarray, valid_mask=None):
    """construct_ei for NonlinFitness (:68-90): mu_var_per_sample = [(mu P×M, var P×M), ...] over
    posterior samples; MAP: one posterior and the full ε matrix, BI: column s of ε for sample s."""
    S = len(mu_var_per_sample)
    constrained = y_max is not None
    acc = None
    for s, (mu, var) in enumerate(mu_var_per_sample):
        M = mu.shape[1]
        if (not constrained) and best is None:
            a = np.zeros(M)
        elif best is None:
            a = feas_prob_host(mu, var, y_max)
        else:
            e = eps if S == 1 else eps[:, s]
            a = ei_nonlin_host(fitness, mu, var, e, best)
            if constrained:
                a = a * feas_prob_host(mu, var, y_max)
        acc = a if acc is None else acc + a
    acq = acc / S
    if valid_mask is not None:
        acq = np.where(np.asarray(valid_mask, bool), acq, 0.0)
    return acq

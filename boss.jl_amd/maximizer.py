"""HipBatchAM — the AcquisitionMaximizer of the plugin trio.

Reproduces SamplingAM (src/acquisition_maximizers/sampling.jl:20-75: draw M candidates from
x_prior inside the domain, acq on each, arg-max) and GridAM (grid.jl:52-65: a fixed list of
points) with ONE device call for all candidates: posterior mean/variance for every output and
hyper-parameter sample, analytic EI × feasibility, make_safe mask, arg-max
(construct_acquisition(::ExpectedImprovement), src/acquisitions/expected_improvement.jl:49-90).

Multi-GPU (one process per GPU): every rank holds the (redundantly factorised) posterior,
evaluates its contiguous shard of the candidates, and the per-rank (max, global index) pairs are
combined with a single 16-byte all-gather over RCCL (distributed.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import numpy as np

from . import acquisition as acq_host
from . import api
from . import distributed as dist_util
from .model import HipGaussianProcessPosterior
from .problem import BossOptions, BossProblem, NonlinFitness, best_so_far, in_bounds, in_cons, in_domain


def _rand_in_domain(x_prior: Callable, domain, rng, max_attempts: int):
    """_rand_in_domain / _rand_in_discrete (sampling.jl:59-75)."""
    for _ in range(max_attempts):
        x = np.asarray(x_prior(rng), float)
        x = np.where(domain.discrete, np.rint(x), x)
        if in_domain(x, domain)[0]:
            return x
    return None


def posteriors_of(problem: BossProblem):
    """model_posterior(problem) (src/posterior.jl:2-3): list over BI samples of posteriors."""
    post = problem.model.model_posterior(problem.params, problem.data)
    return post if isinstance(post, list) else [post]


def acquisition_values(problem: BossProblem, posts: Sequence[HipGaussianProcessPosterior], Xs: np.ndarray,
                       cand: Optional[api.Candidates] = None, eps_seed: Optional[int] = 0):
    """vals = acq.(eachcol(xs)) for acq = construct_safe_acquisition(problem) — device-evaluated.
    Returns (acq[M], local argmax, local max)."""
    ei = problem.acquisition
    Xs = np.asfortranarray(Xs, dtype=np.float64)
    own = cand is None
    if own:
        cand = api.Candidates(Xs, problem.model.device)
    constrained = True      # BossProblem.y_max is always a vector (problem.jl:50), Inf entries count as factor 1
    b = best_so_far(ei.fitness, problem.data.Y, problem.y_max)
    mask = None
    if ei.cons_safe:                                                   # make_safe (expected_improvement.jl:58-65)
        mask = in_bounds(Xs, problem.domain.bounds) & in_cons(Xs, problem.domain.cons)
    if isinstance(ei.fitness, NonlinFitness):
        # expected_improvement(::NonlinFitness) needs a user closure per Monte-Carlo sample: the device
        # provides (μ, σ²) for every candidate / output / posterior sample, the ε-average runs on the host
        if own:
            cand.close()
        mv = [p.mean_and_var(Xs) for p in posts]
        rng = np.random.default_rng(eps_seed)
        count = ei.eps_samples if len(posts) == 1 else len(posts)      # ϵ_sample_count (expected_improvement.jl:116-117)
        eps = acq_host.sample_eps(problem.data.Y.shape[0], count, rng)
        acq = acq_host.acquisition_nonlin(ei.fitness, mv, problem.y_max if constrained else None, b, eps, mask)
        am = int(np.argmax(acq))
        return acq, am, float(acq[am])
    gps = [[s.gp for s in p.slices] for p in posts]
    means = None
    m0 = posts[0].slices[0]._mean_s(Xs)
    if m0 is not None:
        means = np.stack([np.stack([s._mean_s(Xs) for s in p.slices], axis=0) for p in posts], axis=0)   # S×P×M
    acq, am, mx = api.acq_ei(gps, cand, ei.fitness.coefs, problem.y_max if constrained else None, b, mask, means)
    if own:
        cand.close()
    return acq, am, mx


@dataclass
class HipBatchAM:
    """x_prior: rng -> x (the reference takes a MultivariateDistribution); points: optional fixed
    d×M grid (GridAM).  samples: number of candidates drawn when `points` is None."""
    x_prior: Optional[Callable] = None
    samples: int = 0
    points: Optional[np.ndarray] = None
    max_attempts: int = 200
    seed: Optional[int] = None
    group: object = None

    def candidates(self, problem: BossProblem) -> np.ndarray:
        if self.points is not None:
            return np.asfortranarray(self.points, dtype=np.float64)
        rng = np.random.default_rng(self.seed)
        xs = [_rand_in_domain(self.x_prior, problem.domain, rng, self.max_attempts) for _ in range(self.samples)]
        xs = [x for x in xs if x is not None]                          # _reduce_samples (sampling.jl:77-79)
        if not xs:
            raise RuntimeError("HipBatchAM: No samples were successfully drawn! Check the `x_prior` and the `Domain`.")
        return np.asfortranarray(np.stack(xs, axis=1))

    def maximize_acquisition(self, problem: BossProblem, options: BossOptions = BossOptions(), return_all: bool = False):
        """maximize_acquisition(::SamplingAM, problem, options) (sampling.jl:20-40) -> (x, val)."""
        Xs = self.candidates(problem)                                   # identical on every rank (seeded)
        M = Xs.shape[1]
        rank, world = dist_util.rank_world(self.group)
        lo, hi = dist_util.shard_range(M, rank, world)
        posts = posteriors_of(problem)
        if hi > lo:
            acq, am, mx = acquisition_values(problem, posts, Xs[:, lo:hi])
            am += lo
        else:
            acq, am, mx = np.zeros(0), M, -np.inf
        if return_all:
            return Xs, (dist_util.allgather_concat(acq, self.group) if world > 1 else acq)
        mx, am = dist_util.argmax_exchange(mx, am, self.group)
        return Xs[:, am].copy(), mx

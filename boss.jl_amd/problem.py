"""Minimal host-side containers mirroring the reference types the plugin trio is called with.

Only what the hot path's callers need: Domain (src/types/domain.jl:30-84),
ExperimentData (src/types/data.jl:17-31), LinFitness (src/types/fitness.jl), ExpectedImprovement
(src/acquisitions/expected_improvement.jl:39-45), BossProblem (src/types/problem.jl:38-58) and the
few prior distributions the reference's tests use.  Everything here is plain numpy host logic.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np


# ---------------------------------------------------------------- priors (Distributions.jl stand-ins)
@dataclass
class Dirac:
    value: float

    def rand(self, rng):
        return self.value

    def logpdf(self, x):
        return 0.0 if x == self.value else -math.inf

    def grad_logpdf(self, x):
        return 0.0


@dataclass
class LogNormal:
    mu: float = 0.0
    sigma: float = 1.0

    def rand(self, rng):
        return math.exp(self.mu + self.sigma * rng.standard_normal())

    def logpdf(self, x):
        if x <= 0:
            return -math.inf
        z = (math.log(x) - self.mu) / self.sigma
        return -math.log(x * self.sigma) - 0.5 * math.log(2 * math.pi) - 0.5 * z * z

    def grad_logpdf(self, x):
        """d logpdf / dx."""
        return -(1.0 + (math.log(x) - self.mu) / self.sigma ** 2) / x


@dataclass
class MvLogNormal:
    """BOSS.mvlognormal(μ, σ) = MvLogNormal(μ, Diagonal(σ²)) (src/utils/distributions.jl:21-22)."""
    mu: Sequence[float]
    sigma: Sequence[float]

    def rand(self, rng):
        mu, s = np.asarray(self.mu, float), np.asarray(self.sigma, float)
        return np.exp(mu + s * rng.standard_normal(mu.shape[0]))

    def logpdf(self, x):
        x = np.asarray(x, float)
        if np.any(x <= 0):
            return -math.inf
        mu, s = np.asarray(self.mu, float), np.asarray(self.sigma, float)
        z = (np.log(x) - mu) / s
        return float(np.sum(-np.log(x * s) - 0.5 * math.log(2 * math.pi) - 0.5 * z * z))

    def grad_logpdf(self, x):
        x = np.asarray(x, float)
        mu, s = np.asarray(self.mu, float), np.asarray(self.sigma, float)
        return -(1.0 + (np.log(x) - mu) / s ** 2) / x


@dataclass
class MvDirac:
    values: Sequence[float]

    def rand(self, rng):
        return np.asarray(self.values, float).copy()

    def logpdf(self, x):
        return 0.0 if np.array_equal(np.asarray(x, float), np.asarray(self.values, float)) else -math.inf

    def grad_logpdf(self, x):
        return np.zeros(len(self.values))


# ---------------------------------------------------------------- domain / data / fitness
@dataclass
class Domain:
    bounds: tuple                      # (lb, ub)
    discrete: Optional[Sequence[bool]] = None
    cons: Optional[Callable] = None    # feasible iff all(cons(x) >= 0)

    def __post_init__(self):
        lb, ub = (np.asarray(b, float) for b in self.bounds)
        self.bounds = (lb, ub)
        if self.discrete is None:
            self.discrete = np.zeros(lb.shape[0], dtype=bool)
        self.discrete = np.asarray(self.discrete, dtype=bool)
        assert lb.shape == ub.shape == self.discrete.shape      # domain.jl:41

    @property
    def x_dim(self):
        return self.discrete.shape[0]


def in_bounds(X, bounds):
    """src/types/domain.jl:73-78, vectorised over columns of a d×M matrix."""
    X = np.asarray(X, float)
    if X.ndim == 1:
        X = X[:, None]
    lb, ub = bounds
    return ~(np.any(X < lb[:, None], axis=0) | np.any(X > ub[:, None], axis=0))


def in_cons(X, cons):
    """src/types/domain.jl:83-84: user closure, evaluated per column on the host."""
    X = np.asarray(X, float)
    if X.ndim == 1:
        X = X[:, None]
    if cons is None:
        return np.ones(X.shape[1], dtype=bool)
    return np.array([bool(np.all(np.asarray(cons(X[:, j])) >= 0.0)) for j in range(X.shape[1])])


def in_discrete(X, discrete):
    X = np.asarray(X, float)
    if X.ndim == 1:
        X = X[:, None]
    if not np.any(discrete):
        return np.ones(X.shape[1], dtype=bool)
    return np.all(np.rint(X[discrete]) == X[discrete], axis=0)


def in_domain(X, domain: Domain):
    """src/types/domain.jl:66-71."""
    return in_bounds(X, domain.bounds) & in_discrete(X, domain.discrete) & in_cons(X, domain.cons)


@dataclass
class ExperimentData:
    X: np.ndarray      # d×N
    Y: np.ndarray      # P×N

    def __post_init__(self):
        self.X = np.asfortranarray(np.atleast_2d(np.asarray(self.X, float)))
        self.Y = np.asfortranarray(np.atleast_2d(np.asarray(self.Y, float)))
        assert self.X.shape[1] == self.Y.shape[1]


@dataclass
class LinFitness:
    coefs: Sequence[float]

    def __call__(self, y):
        return float(np.asarray(self.coefs, float) @ np.asarray(y, float))


@dataclass
class NonlinFitness:
    """NonlinFitness(fitness::Function) (src/types/fitness.jl): a user closure y -> score."""
    fitness: Callable

    def __call__(self, y):
        return float(self.fitness(np.asarray(y, float)))


@dataclass
class ExpectedImprovement:
    """ExpectedImprovement(; fitness, ϵ_samples = 200, cons_safe = true) (expected_improvement.jl:39-45)."""
    fitness: object
    eps_samples: int = 200
    cons_safe: bool = True


@dataclass
class BossOptions:
    info: bool = False
    debug: bool = False


def is_feasible(y, y_max) -> bool:
    """src/utils/utils.jl:33."""
    return bool(np.all(np.asarray(y) <= np.asarray(y_max)))


def best_so_far(fitness, Y, y_max):
    """best_so_far (src/acquisitions/expected_improvement.jl:134-140): best RAW feasible observation
    (vectorised: this runs once per selection of a sequential batch, over all N observations)."""
    Y = np.asarray(Y, float)
    if Y.size == 0:
        return None
    feas = np.all(Y <= np.asarray(y_max, float)[:, None], axis=0)        # is_feasible (src/utils/utils.jl:33)
    if not feas.any():
        return None
    if isinstance(fitness, LinFitness):
        return float(np.max(np.asarray(fitness.coefs, float) @ Y[:, feas]))
    return max(fitness(Y[:, j]) for j in np.flatnonzero(feas))


@dataclass
class BossProblem:
    """src/types/problem.jl:38-58 (fields the hot path's callers touch)."""
    f: Optional[Callable]
    domain: Domain
    acquisition: ExpectedImprovement
    model: object
    data: ExperimentData
    y_max: Optional[Sequence[float]] = None
    params: object = None
    consistent: bool = False

    def __post_init__(self):
        assert self.domain.x_dim == self.data.X.shape[0]           # problem.jl:49
        if self.y_max is None:
            self.y_max = np.full(self.data.Y.shape[0], np.inf)      # :50
        self.y_max = np.asarray(self.y_max, float)
        assert self.y_max.shape[0] == self.data.Y.shape[0]          # :51
        if np.any(self.domain.discrete) and hasattr(self.model, "make_discrete"):
            self.model = self.model.make_discrete(self.domain.discrete)   # :64-68

    def augment_dataset(self, X, Y):
        """augment_dataset! (problem.jl:191-198)."""
        X = np.atleast_2d(np.asarray(X, float))
        Y = np.atleast_2d(np.asarray(Y, float))
        if X.shape[0] != self.data.X.shape[0]:
            X = X.T
        if Y.shape[0] != self.data.Y.shape[0]:
            Y = Y.T
        self.data = ExperimentData(np.hstack([self.data.X, X]), np.hstack([self.data.Y, Y]))
        self.consistent = False

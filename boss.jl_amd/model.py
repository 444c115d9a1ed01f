"""HipGaussianProcess — the SurrogateModel of the plugin trio (host-side mirror).

Mirrors the reference's GaussianProcess / Semiparametric surface (src/models/gaussian_process.jl,
src/models/semiparametric.jl) over the C ABI:
  model_posterior / model_posterior_slice   -> boss_gp_create + boss_gp_update (resident factor)
  mean / var / mean_and_var / std / ...     -> boss_gp_predict
  data_loglike                              -> boss_gp_update's logpdf (or boss_gp_loglike_batch)
  params_loglike / params_sampler           -> host (prior bookkeeping, out of the GPU's scope)
The GP's prior mean (a user closure, or the Semiparametric parametric model m(x;θ)) is evaluated
on the host and crosses the ABI as dense vectors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import api
from .problem import ExperimentData


@dataclass
class HipGPParams:
    """GaussianProcessParams(λ, α, σ) (gaussian_process.jl:62-70) + optional Semiparametric θ."""
    lengthscales: np.ndarray      # d×P
    amplitudes: np.ndarray        # P
    noise_std: np.ndarray         # P
    theta: Optional[np.ndarray] = None

    def __post_init__(self):
        self.lengthscales = np.atleast_2d(np.asarray(self.lengthscales, float))
        self.amplitudes = np.asarray(self.amplitudes, float).reshape(-1)
        self.noise_std = np.asarray(self.noise_std, float).reshape(-1)

    def slice(self, i: int) -> "HipGPParams":
        """slice(::GaussianProcessParams, idx) (gaussian_process.jl:105-111)."""
        return HipGPParams(self.lengthscales[:, i:i + 1], self.amplitudes[i:i + 1], self.noise_std[i:i + 1], self.theta)


def join_slices(ps: Sequence[HipGPParams]) -> HipGPParams:
    """join_slices (gaussian_process.jl:113-119)."""
    return HipGPParams(np.hstack([p.lengthscales for p in ps]), np.concatenate([p.amplitudes for p in ps]),
                       np.concatenate([p.noise_std for p in ps]), ps[0].theta)


@dataclass
class HipGaussianProcess:
    """GaussianProcess(mean, kernel, lengthscale_priors, amplitude_priors, noise_std_priors)
    (gaussian_process.jl:34-42).  `mean`: None, a length-P vector, or a function x -> length-P
    vector.  `parametric`: optional (x, θ) -> length-P vector with `theta_priors` — the
    Semiparametric model (semiparametric.jl:79-92)."""
    lengthscale_priors: Sequence
    amplitude_priors: Sequence
    noise_std_priors: Sequence
    mean: object = None
    kernel: str = "matern52"                     # default per src/deprecated.jl:34
    discrete: Optional[Sequence[bool]] = None
    parametric: Optional[Callable] = None
    theta_priors: Optional[Sequence] = None
    device: int = 0

    sliceable = True                             # gaussian_process.jl:85

    def make_discrete(self, discrete):
        """make_discrete(m, discrete) (gaussian_process.jl:78-79): wrap the kernel in DiscreteKernel."""
        return HipGaussianProcess(self.lengthscale_priors, self.amplitude_priors, self.noise_std_priors, self.mean,
                                  self.kernel, np.asarray(discrete, bool), self.parametric, self.theta_priors, self.device)

    @property
    def y_dim(self):
        return len(self.amplitude_priors)

    # ------------------------------------------------------------------ prior mean on the host
    def mean_values(self, X, params: Optional[HipGPParams], i: int):
        """m_i(x_j) for all columns j — mean_getindex (gaussian_process.jl:101-103) / parametric(θ)."""
        X = np.asarray(X, float)
        if X.ndim == 1:
            X = X[:, None]
        n = X.shape[1]
        if self.parametric is not None:
            th = params.theta
            return np.array([float(np.asarray(self.parametric(X[:, j], th))[i]) for j in range(n)])
        if self.mean is None:
            return None
        if callable(self.mean):
            return np.array([float(np.asarray(self.mean(X[:, j]))[i]) for j in range(n)])
        return np.full(n, float(np.asarray(self.mean, float)[i]))

    # ------------------------------------------------------------------ priors (host bookkeeping)
    def params_sampler(self):
        """_params_sampler (gaussian_process.jl:291-298)."""
        def sample(rng):
            lam = np.stack([np.atleast_1d(p.rand(rng)) for p in self.lengthscale_priors], axis=1)
            amp = np.array([p.rand(rng) for p in self.amplitude_priors])
            sig = np.array([p.rand(rng) for p in self.noise_std_priors])
            th = None if self.theta_priors is None else np.array([p.rand(rng) for p in self.theta_priors])
            return HipGPParams(lam, amp, sig, th)
        return sample

    def params_loglike(self):
        """params_loglike (gaussian_process.jl:282-289)."""
        def ll(p: HipGPParams):
            v = sum(pr.logpdf(p.lengthscales[:, i]) for i, pr in enumerate(self.lengthscale_priors))
            v += sum(pr.logpdf(p.amplitudes[i]) for i, pr in enumerate(self.amplitude_priors))
            v += sum(pr.logpdf(p.noise_std[i]) for i, pr in enumerate(self.noise_std_priors))
            if self.theta_priors is not None:
                v += sum(pr.logpdf(p.theta[i]) for i, pr in enumerate(self.theta_priors))
            return v
        return ll

    # ------------------------------------------------------------------ likelihood
    def data_loglike(self, data: ExperimentData):
        """data_loglike (gaussian_process.jl:250-267): Σ over outputs of logpdf(FiniteGP_i, Y[i,:]).
        Keeps one resident handle per output; each call is one boss_gp_update."""
        gps = [api.GP(data.X, data.Y[i], self.kernel, self.discrete, self.device) for i in range(data.Y.shape[0])]

        def ll(p: HipGPParams):
            tot = 0.0
            for i, g in enumerate(gps):
                try:
                    tot += g.update(p.lengthscales[:, i], p.amplitudes[i], p.noise_std[i], self.mean_values(data.X, p, i))
                except api.PosDefException:
                    return -math.inf                    # safe_data_loglike (src/surrogate_model.jl:2-12)
            return tot
        ll.handles = gps
        return ll

    def data_loglike_batch(self, data: ExperimentData, samples: Sequence[HipGPParams]) -> np.ndarray:
        """`loglike.(samples)` (src/model_fitters/sampling.jl:64,77) in one batched call per output."""
        S = len(samples)
        tot = np.zeros(S)
        for i in range(data.Y.shape[0]):
            lam = np.stack([s.lengthscales[:, i] for s in samples], axis=1)
            amp = np.array([s.amplitudes[i] for s in samples])
            sig = np.array([s.noise_std[i] for s in samples])
            means = None
            if self.parametric is not None:
                means = np.stack([self.mean_values(data.X, s, i) for s in samples], axis=0)
            else:
                mv = self.mean_values(data.X, None, i)
                means = mv
            ll, _ = api.loglike_batch(data.X, data.Y[i], self.kernel, lam, amp, sig, means, self.discrete, self.device)
            tot += ll
        return tot

    # ------------------------------------------------------------------ posterior
    def model_posterior_slice(self, params: HipGPParams, data: ExperimentData, i: int,
                              reserve: int = 0) -> "HipGaussianProcessPosteriorSlice":
        """model_posterior_slice (gaussian_process.jl:133-141).  reserve: room for later appends."""
        g = api.fit(data.X, data.Y[i], self.kernel, params.lengthscales[:, i], params.amplitudes[i], params.noise_std[i],
                    self.mean_values(data.X, params, i), self.discrete, self.device, reserve)
        return HipGaussianProcessPosteriorSlice(self, params, i, g)

    def model_posterior(self, params, data: ExperimentData, reserve: int = 0):
        """model_posterior (src/posterior.jl:8-19,38-41): a vector of params broadcasts (BI samples)."""
        if isinstance(params, (list, tuple)):
            if len(params) > 1 and reserve == 0:
                # the S samples of a BI fit: per output ONE batched factorisation (boss_gp_fit_batch) — X and y uploaded once,
                # the S posteriors resident as members of one set.  A sample whose matrix is not positive definite raises, as
                # the reference's cholesky would inside the broadcast (src/posterior.jl:15-19)
                P, S = data.Y.shape[0], len(params)
                per_out = []
                for i in range(P):
                    means = [self.mean_values(data.X, p, i) for p in params]
                    mX = None if all(m is None for m in means) else np.stack([np.zeros(data.X.shape[1]) if m is None else np.asarray(m, float)
                                                                              for m in means])
                    gps, _, st = api.fit_batch(data.X, data.Y[i], self.kernel, np.stack([p.lengthscales[:, i] for p in params], axis=1),
                                               [p.amplitudes[i] for p in params], [p.noise_std[i] for p in params], mX, self.discrete,
                                               self.device)
                    if np.any(st != 0):
                        for g in gps:
                            g.close()
                        for row in per_out:
                            for g in row:
                                g.close()
                        bad = int(np.flatnonzero(st)[0])
                        if st[bad] == api.BOSS_E_NOT_PD:
                            raise api.PosDefException(api.BOSS_E_NOT_PD, f"sample {bad}, output {i}: matrix is not positive definite")
                        raise api.BossError(int(st[bad]), f"sample {bad}, output {i}: invalid hyper-parameters")
                    per_out.append(gps)
                return [HipGaussianProcessPosterior([HipGaussianProcessPosteriorSlice(self, params[s], i, per_out[i][s]) for i in range(P)])
                        for s in range(S)]
            return [self.model_posterior(p, data, reserve) for p in params]
        return HipGaussianProcessPosterior([self.model_posterior_slice(params, data, i, reserve)
                                            for i in range(data.Y.shape[0])])


def _clip(var):
    """_clip_var is applied device-side by boss_gp_predict; nothing to do on the host."""
    return var


@dataclass
class HipGaussianProcessPosteriorSlice:
    """GaussianProcessPosterior (gaussian_process.jl:127-131): one output dimension."""
    model: HipGaussianProcess
    params: HipGPParams
    idx: int
    gp: api.GP

    def _mean_s(self, X):
        return self.model.mean_values(X, self.params, self.idx)

    def mean_and_var(self, x):
        x = np.asarray(x, float)
        if x.ndim == 1:                      # vector -> scalars (gaussian_process.jl:169-173)
            mu, var = self.gp.predict(x[:, None], self._mean_s(x))
            return float(mu[0]), float(var[0])
        return self.gp.predict(x, self._mean_s(x))       # matrix -> vectors (:174-178)

    def mean(self, x):
        return self.mean_and_var(x)[0]

    def var(self, x):
        return self.mean_and_var(x)[1]

    def std(self, x):
        return np.sqrt(self.var(x))

    def mean_and_std(self, x):
        mu, var = self.mean_and_var(x)
        return mu, np.sqrt(var)

    def mean_and_var_grad(self, X, mean_grad=None):
        """mean_and_var(post, X) with its gradient w.r.t. the columns of X (analytic, on the device):
        (mu[M], var[M], dmu[d,M], dvar[d,M]).  mean_grad: d×M gradient of the prior mean, if it has one."""
        X = np.asarray(X, float)
        if X.ndim == 1:
            X = X[:, None]
        return self.gp.predict_grad(X, self._mean_s(X), mean_grad)

    def mean_and_cov(self, X):
        """mean_and_cov(post, X::Matrix) (gaussian_process.jl:180-184) -> (mu[M], Σ[M,M])."""
        X = np.asarray(X, float)
        return self.gp.predict_cov(X, self._mean_s(X))

    def cov(self, X):
        return self.mean_and_cov(X)[1]

    def append(self, x, y_i: float) -> float:
        """Posterior of the data augmented by (x, y_i) under the same hyper-parameters — what
        model_posterior returns after augment_dataset! (problem.jl:191-198) — as a block Cholesky
        append on the resident factor (boss_gp_append).  x: length-d vector or d×n matrix."""
        x = np.asarray(x, float)
        return self.gp.append(x, y_i, self._mean_s(x))

    def close(self):
        self.gp.close()


@dataclass
class HipGaussianProcessPosterior:
    """DefaultModelPosterior (src/posterior.jl:31-79): fan-out over output slices; rows = outputs."""
    slices: List[HipGaussianProcessPosteriorSlice]

    def mean_and_var(self, x):
        res = [s.mean_and_var(x) for s in self.slices]
        x = np.asarray(x, float)
        if x.ndim == 1:
            return np.array([r[0] for r in res]), np.array([r[1] for r in res])
        return np.vstack([r[0] for r in res]), np.vstack([r[1] for r in res])

    def mean(self, x):
        return self.mean_and_var(x)[0]

    def var(self, x):
        return self.mean_and_var(x)[1]

    def std(self, x):
        return np.sqrt(self.var(x))

    def mean_and_std(self, x):
        mu, var = self.mean_and_var(x)
        return mu, np.sqrt(var)

    def mean_and_cov(self, X):
        """src/posterior.jl:74-79: means stacked as rows, covariances along dims=3 (M×M×P)."""
        res = [s.mean_and_cov(X) for s in self.slices]
        return np.vstack([r[0] for r in res]), np.stack([r[1] for r in res], axis=2)

    def cov(self, X):
        return self.mean_and_cov(X)[1]

    def append(self, x, y):
        """Append one observation (x, y[P]) (or n of them: x d×n, y P×n) to every output slice."""
        y = np.asarray(y, float)
        for i, s in enumerate(self.slices):
            s.append(x, y[i])

    def close(self):
        for s in self.slices:
            s.close()


def average_mean(posts: Sequence[HipGaussianProcessPosterior], X):
    """average_mean (src/posterior.jl:177-179)."""
    return sum(p.mean(X) for p in posts) / len(posts)

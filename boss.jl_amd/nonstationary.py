"""HipNonstationaryGP — host-side mirror of the reference's NonstationaryGP posterior
(src/models/nonstationary_gp/nonstationary_gp.jl) over the C ABI.

The reference models the lengthscales / amplitudes / noise stds as functions of the input — posteriors of latent
ParametrizedGPs or constants (`_param_posterior_slice`, :198-212).  Those latent models are host-side closures here
(`f_lam(x) -> x_dim vector`, `f_amp(x) -> scalar`, `f_noise(x) -> scalar`, one triple per output); the dense work —
Gibbs Gram matrix, Cholesky, likelihood, prediction — runs on the device (`boss_ngp_*`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import api
from .problem import ExperimentData


def _cols(f: Callable, X: np.ndarray) -> np.ndarray:
    """f at every column of X: rows = points.  Closures flagged `vectorized` take the whole d×M matrix at once
    (the ParametrizedGP posteriors below: one device call for all points)."""
    if getattr(f, "vectorized", False):
        return np.asarray(f(X), float)
    return np.array([np.asarray(f(X[:, j]), float) for j in range(X.shape[1])])


# ---------------------------------------------------------------------------------------------
# ParametrizedGP (src/models/nonstationary_gp/parametrized_gp.jl): the latent model of one hyper-parameter
# of the NonstationaryGP — a zero-mean, unit-amplitude GP over the data points whose (whitened) outputs are the
# parameters; its posterior mean, pushed through Normal-cdf -> target quantile -> activation, is the value of
# the hyper-parameter at x.
# ---------------------------------------------------------------------------------------------
@dataclass
class HipParametrizedGPParams:
    """ParametrizedGPParams(X, μ, L, yϵ, λ) (parametrized_gp.jl:64-76)."""
    X: np.ndarray
    mu: np.ndarray
    L: np.ndarray
    yeps: np.ndarray
    lengthscale: np.ndarray


@dataclass
class HipParametrizedGP:
    """ParametrizedGP(kernel, lengthscale_prior, target_dist, act_func, noise_std) (:39-47).  The reference requires a
    Dirac lengthscale prior (:193): `lengthscale` holds its values.  `target_dist`: None or anything with `.ppf(u)`
    (a frozen scipy.stats distribution); `act_func` must accept arrays."""
    lengthscale: Sequence[float]
    kernel: str = "matern32"
    target_dist: object = None
    act_func: Callable = staticmethod(lambda z: z)
    noise_std: float = 0.0
    device: int = 0

    def transform(self, y):
        """construct_variable_transform (:122-134) then act_func: z = quantile(target, cdf(Normal(0,1), y))."""
        y = np.asarray(y, float)
        if self.target_dist is not None:
            from scipy.special import ndtr
            y = self.target_dist.ppf(ndtr(y))
        return self.act_func(y)

    def params_sampler(self, data: ExperimentData):
        """_params_sampler (:191-215): L = chol of the prior covariance at the data points (finite_param_gp, :136-158),
        yϵ ~ N(0, I)."""
        lam = np.asarray(self.lengthscale, float)
        n = data.X.shape[1]
        g = api.GP(data.X, np.zeros(n), self.kernel, None, self.device)
        try:
            g.update(lam, 1.0, self.noise_std)
            L, _ = g.factor()
        finally:
            g.close()
        mu = np.zeros(n)

        def sample(rng):
            return HipParametrizedGPParams(data.X, mu, L, rng.standard_normal(n), lam)
        return sample

    def params_loglike(self, data: ExperimentData = None):
        """params_loglike (:160-189): logpdf(MvNormal(0, I), yϵ)."""
        return lambda p: float(-0.5 * (p.yeps @ p.yeps) - 0.5 * len(p.yeps) * math.log(2.0 * math.pi))

    def model_posterior(self, params: HipParametrizedGPParams, data: ExperimentData = None):
        """model_posterior (:90-106): x -> act(ft(mean of the GP posterior conditioned on y = L yϵ + μ)).  The returned
        closure takes one point or a d×M matrix (one device call); `.close()` releases the handle."""
        y = params.L @ params.yeps + params.mu
        g = api.GP(params.X, y, self.kernel, None, self.device)
        g.update(params.lengthscale, 1.0, self.noise_std)

        def post(x):
            x = np.asarray(x, float)
            m, _ = g.predict(x[:, None] if x.ndim == 1 else x)
            z = self.transform(m)
            return float(z[0]) if x.ndim == 1 else z
        post.vectorized = True
        post.close = g.close
        return post

    def model_posterior_lookup(self, params: HipParametrizedGPParams, data: ExperimentData = None):
        """model_posterior_lookup (:108-120): the values at the data points only, no kernel matrix."""
        vals = self.transform(params.L @ params.yeps + params.mu)
        table = {tuple(params.X[:, j]): float(vals[j]) for j in range(params.X.shape[1])}

        def post(x):
            x = np.asarray(x, float)
            if x.ndim == 1:
                return table[tuple(x)]
            return np.array([table[tuple(x[:, j])] for j in range(x.shape[1])])
        post.vectorized = True
        return post


def stack_latents(posts: Sequence[Callable]) -> Callable:
    """x -> apply.(posts, Ref(x)) (nonstationary_gp.jl:209-212): the x_dim latent lengthscale models of one output as
    one closure returning a d-vector (or M×d for a matrix of points)."""
    def f(x):
        x = np.asarray(x, float)
        cols = [np.asarray(p(x), float) for p in posts]
        return np.array(cols) if x.ndim == 1 else np.stack(cols, axis=1)
    f.vectorized = all(getattr(p, "vectorized", False) for p in posts)
    return f


@dataclass
class HipNonstationaryPosteriorSlice:
    """GaussianProcessPosterior over the NonstationaryKernel (nonstationary_gp.jl:153-157)."""
    gp: api.GibbsGP
    f_lam: Callable
    f_amp: Callable
    mean_fn: Optional[Callable]
    discrete: Optional[np.ndarray]
    f_noise: Optional[Callable] = None                          # σ(·): needed by append only

    def append(self, x, y) -> float:
        """augment_dataset! (src/types/problem.jl:191-198) on the fitted slice: the latent models are evaluated at the new points, the
        system is rebuilt and factorised (boss_ngp_append).  x: d×m (or length d), y: m.  Returns the logpdf of all points."""
        if self.f_noise is None:
            raise ValueError("this slice was built without its noise model")
        X = np.asarray(x, float).reshape(self.gp.d, -1)
        Xr = self._round(X)
        ms = None if self.mean_fn is None else np.array([float(self.mean_fn(X[:, j])) for j in range(X.shape[1])])
        return self.gp.append(X, np.asarray(y, float).reshape(-1), _cols(self.f_lam, Xr).T, _cols(self.f_amp, Xr).reshape(-1),
                              _cols(self.f_noise, X).reshape(-1), ms)

    def loglike_grad(self):
        """(logpdf, dlam[d, N], damp[N], dnoise[N], dmean[N]): data_loglike_slice (nonstationary_gp.jl:237-245) of the fitted slice and
        its partial derivatives w.r.t. the latent models' values at the training points (boss_ngp_loglike_grad) — the cotangents a
        gradient-based fitter chains through its latent models."""
        return self.gp.loglike_grad()

    def _round(self, X):
        if self.discrete is None:
            return X
        X = X.copy()
        X[self.discrete] = np.rint(X[self.discrete])            # DiscreteKernel: the kernel, hence λ(·), α(·), sees rounded inputs
        return X

    def mean_and_var(self, x):
        x = np.asarray(x, float)
        vec = x.ndim == 1
        X = x[:, None] if vec else x
        Xr = self._round(X)
        ms = None if self.mean_fn is None else np.array([float(self.mean_fn(X[:, j])) for j in range(X.shape[1])])
        mu, var = self.gp.predict(X, _cols(self.f_lam, Xr).T, _cols(self.f_amp, Xr).reshape(-1), ms)
        return (float(mu[0]), float(var[0])) if vec else (mu, var)

    def mean_and_var_grad(self, X, lam_jac: Optional[Callable] = None, amp_jac: Optional[Callable] = None, mean_grad=None,
                          fd_step: float = 1e-6):
        """mean_and_var and its gradient w.r.t. the candidate columns (boss_ngp_predict_grad) — what ForwardDiff pushes through the
        posterior inside OptimizationAM (src/acquisition_maximizers/optimization.jl:36).  The candidate also enters through the latent
        λ(x*), α(x*): `lam_jac(x) -> d×d` ([l, m] = ∂λ_l/∂x_m) and `amp_jac(x) -> d` supply their Jacobians; without them central
        differences of the host closures are taken (step fd_step).  Returns (mu[M], var[M], dmu[d, M], dvar[d, M])."""
        X = np.asarray(X, float)
        if X.ndim == 1:
            X = X[:, None]
        d, M = X.shape
        Xr = self._round(X)

        def jac(f, x, n_out):
            J = np.zeros((n_out, d))
            for m in range(d):
                e = np.zeros(d)
                e[m] = fd_step
                J[:, m] = (np.atleast_1d(np.asarray(f(x + e), float)) - np.atleast_1d(np.asarray(f(x - e), float))) / (2 * fd_step)
            return J
        Dl = np.stack([np.asarray(lam_jac(Xr[:, j]), float) if lam_jac else jac(self.f_lam, Xr[:, j], d) for j in range(M)], axis=2)
        Da = np.stack([np.asarray(amp_jac(Xr[:, j]), float).reshape(-1) if amp_jac else jac(self.f_amp, Xr[:, j], 1)[0] for j in range(M)], axis=1)
        if self.discrete is not None:                        # rounded dimensions: the latent models are piecewise constant in them
            Dl[:, self.discrete, :] = 0.0
            Da[self.discrete, :] = 0.0
        ms = None if self.mean_fn is None else np.array([float(self.mean_fn(X[:, j])) for j in range(M)])
        return self.gp.predict_grad(X, _cols(self.f_lam, Xr).T, _cols(self.f_amp, Xr).reshape(-1), Dl, Da, ms, mean_grad)

    def mean(self, x):
        return self.mean_and_var(x)[0]

    def var(self, x):
        return self.mean_and_var(x)[1]

    def std(self, x):
        return np.sqrt(self.var(x))

    def close(self):
        self.gp.close()


@dataclass
class HipNonstationaryGP:
    """finite_nongp (nonstationary_gp.jl:183-196) for y_dim outputs: per output i the closures
    f_lam[i], f_amp[i], f_noise[i] and an optional prior mean function."""
    f_lam: Sequence[Callable]
    f_amp: Sequence[Callable]
    f_noise: Sequence[Callable]
    mean: Optional[Sequence[Optional[Callable]]] = None
    discrete: Optional[Sequence[bool]] = None
    device: int = 0

    def _latent_at_data(self, X, i):
        disc = None if self.discrete is None else np.asarray(self.discrete, bool)
        Xr = X.copy()
        if disc is not None:
            Xr[disc] = np.rint(Xr[disc])
        m = None if self.mean is None or self.mean[i] is None else np.array([float(self.mean[i](X[:, j])) for j in range(X.shape[1])])
        return _cols(self.f_lam[i], Xr).T, _cols(self.f_amp[i], Xr).reshape(-1), _cols(self.f_noise[i], X).reshape(-1), m, disc

    def model_posterior_slice(self, data: ExperimentData, i: int) -> HipNonstationaryPosteriorSlice:
        lam, amp, noi, m, disc = self._latent_at_data(data.X, i)
        g = api.GibbsGP(data.X, data.Y[i], disc, self.device)
        try:
            g.update(lam, amp, noi, m)
        except Exception:
            g.close()
            raise
        return HipNonstationaryPosteriorSlice(g, self.f_lam[i], self.f_amp[i], None if self.mean is None else self.mean[i], disc, self.f_noise[i])

    def model_posterior(self, data: ExperimentData) -> List[HipNonstationaryPosteriorSlice]:
        return [self.model_posterior_slice(data, i) for i in range(data.Y.shape[0])]

    def data_loglike(self, data: ExperimentData) -> float:
        """data_loglike (nonstationary_gp.jl:231-245): Σ over outputs of logpdf(FiniteGP_i, Y[i, :]); -Inf when not PD."""
        tot = 0.0
        for i in range(data.Y.shape[0]):
            lam, amp, noi, m, disc = self._latent_at_data(data.X, i)
            g = api.GibbsGP(data.X, data.Y[i], disc, self.device)
            try:
                tot += g.update(lam, amp, noi, m)
            except api.PosDefException:
                return -np.inf
            finally:
                g.close()
        return tot

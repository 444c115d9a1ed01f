"""HipNonstationaryGP — host-side mirror of the reference's NonstationaryGP posterior
(src/models/nonstationary_gp/nonstationary_gp.jl) over the C ABI.

The reference models the lengthscales / amplitudes / noise stds as functions of the input — posteriors of latent
ParametrizedGPs or constants (`_param_posterior_slice`, :198-212).  Those latent models are host-side closures here
(`f_lam(x) -> x_dim vector`, `f_amp(x) -> scalar`, `f_noise(x) -> scalar`, one triple per output); the dense work —
Gibbs Gram matrix, Cholesky, likelihood, prediction — runs on the device (`boss_ngp_*`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import api
from .problem import ExperimentData


def _cols(f: Callable, X: np.ndarray) -> np.ndarray:
    return np.array([np.asarray(f(X[:, j]), float) for j in range(X.shape[1])])


@dataclass
class HipNonstationaryPosteriorSlice:
    """GaussianProcessPosterior over the NonstationaryKernel (nonstationary_gp.jl:153-157)."""
    gp: api.GibbsGP
    f_lam: Callable
    f_amp: Callable
    mean_fn: Optional[Callable]
    discrete: Optional[np.ndarray]

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
        return HipNonstationaryPosteriorSlice(g, self.f_lam[i], self.f_amp[i], None if self.mean is None else self.mean[i], disc)

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

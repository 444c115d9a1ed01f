"""HipGradientGaussianProcess — host-side mirror of the reference's GradientGaussianProcess
(src/models/gradient_gp.jl) and GradientData (src/data/gradient_data.jl) over the C ABI.

Every evaluated point contributes its value and its gradient, so n points give an n(1+x_dim) system per output:
  model_posterior_slice / data_loglike   -> boss_ggp_create + boss_ggp_update (augmented Gram, Cholesky, α-solve, logpdf)
  mean / var / mean_and_var              -> boss_gp_predict on the augmented factor
The acquisition maximizers (HipBatchAM, …) take these posteriors unchanged; the entry points that assume
value-only observations (append, gradients w.r.t. candidates, covariance) are not available for this model.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import api
from .model import HipGaussianProcessPosterior, HipGaussianProcessPosteriorSlice
from .problem import ExperimentData


@dataclass
class GradientData(ExperimentData):
    """GradientData(X, Y, dY) (gradient_data.jl:25-38): dY is y_dim × x_dim × n, dY[i, k, j] = ∂y_i/∂x_k at X[:, j]."""
    dY: np.ndarray = None

    def __post_init__(self):
        super().__post_init__()
        self.dY = np.asarray(self.dY, float)
        if self.dY.ndim == 2:                                   # a single output given as x_dim × n
            self.dY = self.dY[None, :, :]
        assert self.dY.shape == (self.Y.shape[0], self.X.shape[0], self.X.shape[1]), "dY must be y_dim × x_dim × n"

    def augment(self, x, y, J) -> "GradientData":
        """augment_dataset (gradient_data.jl:47-68): J is the y_dim × x_dim Jacobian at x."""
        x = np.asarray(x, float).reshape(-1, 1)
        y = np.asarray(y, float).reshape(-1, 1)
        J = np.asarray(J, float).reshape(self.Y.shape[0], self.X.shape[0], 1)
        return GradientData(np.hstack([self.X, x]), np.hstack([self.Y, y]), np.concatenate([self.dY, J], axis=2))

    def slice(self, i: int) -> "GradientData":
        """slice(::GradientData, idx) (gradient_data.jl:76-82)."""
        return GradientData(self.X, self.Y[i:i + 1], self.dY[i:i + 1])


@dataclass
class HipGradientGPParams:
    """GradientGaussianProcessParams(λ, α, σ, σ_∂) (gradient_gp.jl:41-51)."""
    lengthscales: np.ndarray      # d×P
    amplitudes: np.ndarray        # P
    noise_std: np.ndarray         # P
    grad_noise_std: np.ndarray    # P

    def __post_init__(self):
        self.lengthscales = np.atleast_2d(np.asarray(self.lengthscales, float))
        self.amplitudes = np.asarray(self.amplitudes, float).reshape(-1)
        self.noise_std = np.asarray(self.noise_std, float).reshape(-1)
        self.grad_noise_std = np.asarray(self.grad_noise_std, float).reshape(-1)

    def slice(self, i: int) -> "HipGradientGPParams":
        """slice(::GradientGaussianProcessParams, idx) (gradient_gp.jl:94-101)."""
        return HipGradientGPParams(self.lengthscales[:, i:i + 1], self.amplitudes[i:i + 1], self.noise_std[i:i + 1],
                                   self.grad_noise_std[i:i + 1])


def join_gradient_slices(ps: Sequence[HipGradientGPParams]) -> HipGradientGPParams:
    """join_slices (gradient_gp.jl:103-110)."""
    return HipGradientGPParams(np.hstack([p.lengthscales for p in ps]), np.concatenate([p.amplitudes for p in ps]),
                               np.concatenate([p.noise_std for p in ps]), np.concatenate([p.grad_noise_std for p in ps]))


class HipGradientGPPosteriorSlice(HipGaussianProcessPosteriorSlice):
    """GradientGPPosteriorSlice (gradient_gp.jl:57-64): μ = k*·α, σ² = max(0, k(x,x) − ‖L⁻¹k*‖²)  (:334-361)."""

    def _mean_s(self, X):
        return None                                             # the model's mean is not used (:334-337)

    def _unavailable(self, *a, **k):
        raise NotImplementedError("not defined for gradient-observation posteriors")

    mean_and_cov = cov = _unavailable                        # (mean_and_var_grad: inherited — boss_gp_predict_grad takes these posteriors)

    def append(self, x, y, dy) -> float:
        """augment_dataset! (src/types/problem.jl:191-198) on the fitted slice: new points with values and gradients, hyper-parameters
        unchanged (boss_ggp_append: the augmented system is rebuilt and factorised, as in the reference)."""
        return self.gp.append(x, y, dy)


@dataclass
class HipGradientGaussianProcess:
    """GradientGaussianProcess(mean, kernel, lengthscale_priors, amplitude_priors, noise_std_priors,
    grad_noise_std_priors) (gradient_gp.jl:22-31).  `mean` is carried like the reference does and, like there,
    used by neither the posterior nor the likelihood."""
    lengthscale_priors: Sequence
    amplitude_priors: Sequence
    noise_std_priors: Sequence
    grad_noise_std_priors: Sequence
    mean: object = None
    kernel: str = "matern52"
    device: int = 0

    sliceable = True                                            # gradient_gp.jl:69

    @property
    def y_dim(self):
        return len(self.amplitude_priors)

    def params_sampler(self):
        def sample(rng):
            lam = np.stack([np.atleast_1d(p.rand(rng)) for p in self.lengthscale_priors], axis=1)
            return HipGradientGPParams(lam, np.array([p.rand(rng) for p in self.amplitude_priors]),
                                       np.array([p.rand(rng) for p in self.noise_std_priors]),
                                       np.array([p.rand(rng) for p in self.grad_noise_std_priors]))
        return sample

    def params_loglike(self):
        """params_loglike (gradient_gp.jl:403-411)."""
        def ll(p: HipGradientGPParams):
            v = sum(pr.logpdf(p.lengthscales[:, i]) for i, pr in enumerate(self.lengthscale_priors))
            v += sum(pr.logpdf(p.amplitudes[i]) for i, pr in enumerate(self.amplitude_priors))
            v += sum(pr.logpdf(p.noise_std[i]) for i, pr in enumerate(self.noise_std_priors))
            v += sum(pr.logpdf(p.grad_noise_std[i]) for i, pr in enumerate(self.grad_noise_std_priors))
            return v
        return ll

    def data_loglike(self, data: GradientData):
        """data_loglike (gradient_gp.jl:367-397), summed over the outputs; one resident handle per output,
        each call one boss_ggp_update.  A non-PD augmented matrix yields -Inf (safe_data_loglike)."""
        gps = [api.GradGP(data.X, data.Y[i], data.dY[i], self.kernel, self.device) for i in range(data.Y.shape[0])]

        def ll(p: HipGradientGPParams):
            tot = 0.0
            for i, g in enumerate(gps):
                try:
                    tot += g.update(p.lengthscales[:, i], p.amplitudes[i], p.noise_std[i], p.grad_noise_std[i])
                except api.PosDefException:
                    return -math.inf
            return tot
        ll.handles = gps
        return ll

    def data_loglike_grad(self, data: GradientData):
        """data_loglike with its gradient w.r.t. (λ, α, σ, σ_∂) — what ForwardDiff yields through gradient_gp.jl:367-397 inside
        OptimizationMAP (src/model_fitters/optimization.jl:146-164): p -> (ℓ, HipGradientGPParams of partial derivatives), per
        output one boss_ggp_update + boss_ggp_loglike_grad on a resident handle.  Non-PD: (-Inf, zeros)."""
        gps = [api.GradGP(data.X, data.Y[i], data.dY[i], self.kernel, self.device) for i in range(data.Y.shape[0])]

        def llg(p: HipGradientGPParams):
            d, P = p.lengthscales.shape
            gl, ga, gs, gd = np.zeros((d, P)), np.zeros(P), np.zeros(P), np.zeros(P)
            tot = 0.0
            for i, g in enumerate(gps):
                try:
                    g.update(p.lengthscales[:, i], p.amplitudes[i], p.noise_std[i], p.grad_noise_std[i])
                except api.PosDefException:
                    return -math.inf, HipGradientGPParams(np.zeros((d, P)), np.zeros(P), np.zeros(P), np.zeros(P))
                ll, gr = g.loglike_grad()
                tot += ll
                gl[:, i], ga[i], gs[i], gd[i] = gr[:d], gr[d], gr[d + 1], gr[d + 2]
            return tot, HipGradientGPParams(gl, ga, gs, gd)
        llg.handles = gps
        return llg

    def data_loglike_batch(self, data: GradientData, samples: Sequence[HipGradientGPParams]) -> np.ndarray:
        """`loglike.(samples)` (src/model_fitters/sampling.jl:64,77) — what HipBatchedMAP calls; the augmented
        factorisations are large enough to fill the device one at a time, so this is a loop on resident handles."""
        ll = self.data_loglike(data)
        try:
            return np.array([ll(p) for p in samples])
        finally:
            for g in ll.handles:
                g.close()

    def model_posterior_slice(self, params: HipGradientGPParams, data: GradientData, i: int) -> HipGradientGPPosteriorSlice:
        """model_posterior_slice (gradient_gp.jl:307-329)."""
        g = api.GradGP(data.X, data.Y[i], data.dY[i], self.kernel, self.device)
        try:
            g.update(params.lengthscales[:, i], params.amplitudes[i], params.noise_std[i], params.grad_noise_std[i])
        except Exception:
            g.close()
            raise
        return HipGradientGPPosteriorSlice(self, params, i, g)

    def model_posterior(self, params, data: GradientData):
        if isinstance(params, (list, tuple)):
            return [self.model_posterior(p, data) for p in params]
        return HipGaussianProcessPosterior([self.model_posterior_slice(params, data, i) for i in range(data.Y.shape[0])])

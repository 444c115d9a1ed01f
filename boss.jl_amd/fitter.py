"""HipBatchedMAP — the ModelFitter of the plugin trio: SamplingMAP semantics
(src/model_fitters/sampling.jl:17-78: draw S prior samples, evaluate loglike on each, keep the
arg-max) with the S data log-likelihoods evaluated in ONE batched device call per output
(boss_gp_loglike_batch) instead of S sequential Cholesky factorisations.

With torch.distributed initialised (one process per GPU) the S samples are sharded across ranks
with no data-path collective; the per-rank best (loglike, index) pairs are combined with one
all_gather (see distributed.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import api
from . import distributed as dist_util
from .model import HipGaussianProcess, HipGPParams
from .problem import BossOptions, BossProblem, Dirac, MvDirac


@dataclass
class MAPParams:
    """MAPParams(params, loglike) (src/types/parameters.jl:118-123)."""
    params: HipGPParams
    loglike: float


@dataclass
class HipBatchedMAP:
    samples: int
    seed: Optional[int] = None
    group: object = None            # torch.distributed process group or None

    def estimate_parameters(self, problem: BossProblem, options: BossOptions = BossOptions(), return_all: bool = False):
        """estimate_parameters(::SamplingMAP, problem, options; return_all) (sampling.jl:17-28)."""
        model: HipGaussianProcess = problem.model
        rng = np.random.default_rng(dist_util.shared_seed(self.seed, self.group))     # the same draws on every rank
        sampler = model.params_sampler()
        prior_ll = model.params_loglike()
        draws: List[HipGPParams] = [sampler(rng) for _ in range(self.samples)]     # same stream on every rank
        rank, world = dist_util.rank_world(self.group)
        lo, hi = dist_util.shard_range(self.samples, rank, world)
        mine = draws[lo:hi]
        vals = np.full(self.samples, -np.inf)
        if mine:
            vals[lo:hi] = model.data_loglike_batch(problem.data, mine) + np.array([prior_ll(p) for p in mine])
        if return_all:
            vals = dist_util.allgather_concat(vals[lo:hi], self.group) if world > 1 else vals
            return [MAPParams(p, float(v)) for p, v in zip(draws, vals)]
        # sampling_optim (sampling.jl:59-71): strict `>` keeps the FIRST best sample
        if hi > lo:
            j = int(np.argmax(vals[lo:hi])) + lo
            best_v, best_i = float(vals[j]), j
        else:
            best_v, best_i = -np.inf, self.samples
        best_v, best_i = dist_util.argmax_exchange(best_v, best_i, self.group)
        return MAPParams(draws[best_i], best_v)


@dataclass
class HipGradientMAP:
    """OptimizationMAP semantics (src/model_fitters/optimization.jl:13-164): multistart local maximisation
    of the log-posterior  loglike(data | θ) + logprior(θ)  over the GP hyper-parameters, the best local
    optimum wins.  The reference hands the objective to an Optimization.jl algorithm with automatic
    differentiation; here the starts advance in lockstep and every round is ONE device call per output —
    boss_gp_loglike_grad_batch: values and analytic gradients of all trial points — and the ascent runs in log-parameter space
    (the reference softplus/log-transforms positive parameters the same way, :120-144) with a
    Barzilai–Borwein-free backtracking step.  Parameters with a Dirac prior stay fixed (dirac.jl:36-77).
    Multi-GPU: the starts are sharded across ranks, 16-byte arg-max exchange (as HipBatchedMAP)."""
    multistart: int = 8
    iters: int = 40
    seed: Optional[int] = None
    group: object = None
    step0: float = 0.3
    starts: Optional[List[HipGPParams]] = None       # set_starts (optimization.jl): given starts replace the prior draws

    def _objective_batch(self, model, prior_ll, data, plist: List[HipGPParams]):
        """log-posterior and its gradient w.r.t. (λ[d,P], α[P], σ[P]) for every parameter set of `plist`: one batched
        device call per output (boss_gp_loglike_grad_batch) — all starts of a round advance together."""
        S = len(plist)
        d, P = plist[0].lengthscales.shape
        tot = np.array([prior_ll(p) for p in plist], dtype=float)
        gl, ga, gs = np.zeros((S, d, P)), np.zeros((S, P)), np.zeros((S, P))
        for i in range(P):
            lam = np.stack([p.lengthscales[:, i] for p in plist], axis=1)
            amp = np.array([p.amplitudes[i] for p in plist])
            sig = np.array([p.noise_std[i] for p in plist])
            ll, st, gr = api.loglike_batch(data.X, data.Y[i], model.kernel, lam, amp, sig, model.mean_values(data.X, None, i),
                                           model.discrete, model.device, want_grad=True)
            tot += np.where(st == 0, ll, -np.inf)
            gl[:, :, i], ga[:, i], gs[:, i] = gr[:d].T, gr[d], gr[d + 1]
            for k, p in enumerate(plist):
                gl[k, :, i] += np.atleast_1d(model.lengthscale_priors[i].grad_logpdf(p.lengthscales[:, i]))
                ga[k, i] += model.amplitude_priors[i].grad_logpdf(p.amplitudes[i])
                gs[k, i] += model.noise_std_priors[i].grad_logpdf(p.noise_std[i])
        tot = np.where(np.isfinite(tot), tot, -np.inf)
        return tot, gl, ga, gs

    def _objective_gradient_model(self, model, prior_ll, data, plist):
        """The same for a HipGradientGaussianProcess (values + gradients): per start and output one boss_ggp_update and one
        boss_ggp_loglike_grad on resident handles (an augmented factorisation fills the device on its own); a fourth parameter group,
        the gradient noise σ_∂ (what ForwardDiff yields through gradient_gp.jl:367-397 inside OptimizationMAP)."""
        S = len(plist)
        d, P = plist[0].lengthscales.shape
        tot = np.array([prior_ll(p) for p in plist], dtype=float)
        gl, ga, gs, gd = np.zeros((S, d, P)), np.zeros((S, P)), np.zeros((S, P)), np.zeros((S, P))
        llg = model.data_loglike_grad(data)
        try:
            for k, p in enumerate(plist):
                ll, g = llg(p)
                tot[k] += ll
                if not np.isfinite(ll):
                    continue
                gl[k], ga[k], gs[k], gd[k] = g.lengthscales, g.amplitudes, g.noise_std, g.grad_noise_std
                for i in range(P):
                    gl[k, :, i] += np.atleast_1d(model.lengthscale_priors[i].grad_logpdf(p.lengthscales[:, i]))
                    ga[k, i] += model.amplitude_priors[i].grad_logpdf(p.amplitudes[i])
                    gs[k, i] += model.noise_std_priors[i].grad_logpdf(p.noise_std[i])
                    gd[k, i] += model.grad_noise_std_priors[i].grad_logpdf(p.grad_noise_std[i])
        finally:
            for h in llg.handles:
                h.close()
        tot = np.where(np.isfinite(tot), tot, -np.inf)
        return tot, gl, ga, gs, gd

    def estimate_parameters(self, problem: BossProblem, options: BossOptions = BossOptions(), return_all: bool = False):
        model = problem.model
        grad_model = hasattr(model, "grad_noise_std_priors")        # HipGradientGaussianProcess (gradient_gp.py)
        if not grad_model and model.parametric is not None:
            raise NotImplementedError("HipGradientMAP treats the prior mean as fixed; use HipBatchedMAP for Semiparametric models")
        data = problem.data
        rng = np.random.default_rng(dist_util.shared_seed(self.seed, self.group))     # the same starts on every rank
        sampler, prior_ll = model.params_sampler(), model.params_loglike()
        if self.starts is not None:
            starts = list(self.starts)
        else:
            starts = [sampler(rng) for _ in range(self.multistart)]                      # same stream on every rank
        nstart = len(starts)
        rank, world = dist_util.rank_world(self.group)
        lo, hi = dist_util.shard_range(nstart, rank, world)
        P = data.Y.shape[0]
        # the parameter groups the ascent moves, and which of their entries are free (a Dirac prior fixes its parameter)
        free = [np.array([not isinstance(pr, (Dirac, MvDirac)) for pr in model.lengthscale_priors])[None, :],
                np.array([not isinstance(pr, Dirac) for pr in model.amplitude_priors]),
                np.array([not isinstance(pr, Dirac) for pr in model.noise_std_priors])]
        if grad_model:
            free.append(np.array([not isinstance(pr, Dirac) for pr in model.grad_noise_std_priors]))
            parts_of = lambda p: [p.lengthscales, p.amplitudes, p.noise_std, p.grad_noise_std]
            remake = lambda p, q: type(p)(q[0], q[1], q[2], q[3])
            objective = lambda pl: (lambda r: (r[0], list(r[1:])))(self._objective_gradient_model(model, prior_ll, data, pl))
        else:
            parts_of = lambda p: [p.lengthscales, p.amplitudes, p.noise_std]
            remake = lambda p, q: HipGPParams(q[0], q[1], q[2], p.theta)
            objective = lambda pl: (lambda r: (r[0], list(r[1:])))(self._objective_batch(model, prior_ll, data, pl))
        results = []
        if hi > lo:
            # every start runs the same backtracking ascent it would run alone; the starts advance in lockstep so that each
            # round's trial points are evaluated by ONE batched device call per output
            ps = [starts[k] for k in range(lo, hi)]
            f, grads = objective(ps)
            grads = [np.array(g) for g in grads]
            n = len(ps)
            step = np.full(n, self.step0)
            its = np.zeros(n, dtype=int)
            alive = np.isfinite(f) & (self.iters > 0)
            while alive.any():
                idx, trial = [], []
                for k in np.flatnonzero(alive):
                    parts = parts_of(ps[k])
                    # ascent direction in log-space: ∂f/∂log θ = θ ∂f/∂θ ; fixed (Dirac) parameters do not move
                    dirs = [np.where(fr, q * g[k], 0.0) for q, g, fr in zip(parts, grads, free)]
                    nrm = math.sqrt(float(sum((dq * dq).sum() for dq in dirs)))
                    if nrm < 1e-10:
                        alive[k] = False
                        continue
                    idx.append(k)
                    trial.append(remake(ps[k], [q * np.exp(step[k] * dq / nrm) for q, dq in zip(parts, dirs)]))
                if not idx:
                    break
                fq, gq = objective(trial)
                for j, k in enumerate(idx):
                    if fq[j] > f[k]:
                        ps[k], f[k] = trial[j], fq[j]
                        for g, gn in zip(grads, gq):
                            g[k] = gn[j]
                        step[k] = min(step[k] * 1.6, 2.0)
                        its[k] += 1
                        if its[k] >= self.iters:
                            alive[k] = False
                    else:
                        step[k] *= 0.4
                        if step[k] <= 1e-6:
                            alive[k] = False
            results = [(lo + k, ps[k], float(f[k])) for k in range(n)]
        if return_all:
            return [MAPParams(p, f) for _, p, f in results]
        best = max(results, key=lambda r: (r[2], -r[0])) if results else None
        best_v, best_i = (best[2], best[0]) if best else (-np.inf, nstart)
        best_v, gi = dist_util.argmax_exchange(best_v, best_i, self.group)
        if world == 1:
            return MAPParams(best[1], best_v)
        mine = best[1] if best and best[0] == gi else None
        shapes = [np.shape(q) for q in parts_of(starts[0])]
        sizes = [int(np.prod(sh)) for sh in shapes]
        flat = None if mine is None else np.concatenate([np.asarray(q).reshape(-1, order="F") for q in parts_of(mine)])
        flat = dist_util.broadcast_array(flat, (sum(sizes),), dist_util.owner_of_index(gi, nstart, world), self.group)
        offs = np.concatenate([[0], np.cumsum(sizes)])
        parts = [flat[offs[i]:offs[i + 1]].reshape(shapes[i], order="F") for i in range(len(shapes))]
        return MAPParams(remake(starts[0], parts), best_v)


@dataclass
class HipSampleOptMAP:
    """SampleOptMAP (src/model_fitters/sample_opt.jl:14-49): draw `samples` prior samples, score them with the
    batched likelihood (HipBatchedMAP, return_all), start `multistart` gradient ascents (HipGradientMAP) from the
    best ones."""
    samples: int = 2000
    multistart: int = 20
    iters: int = 40
    seed: Optional[int] = None
    group: object = None

    def __post_init__(self):
        assert self.samples >= self.multistart                       # sample_opt.jl:30

    def estimate_parameters(self, problem: BossProblem, options: BossOptions = BossOptions(), return_all: bool = False):
        scored = HipBatchedMAP(self.samples, self.seed, self.group).estimate_parameters(problem, options, return_all=True)
        order = sorted(range(len(scored)), key=lambda i: -scored[i].loglike)             # sortperm(...; rev=true), stable
        starts = [scored[i].params for i in order[:self.multistart]]
        opt = HipGradientMAP(self.multistart, self.iters, self.seed, self.group, starts=starts)
        return opt.estimate_parameters(problem, options, return_all)

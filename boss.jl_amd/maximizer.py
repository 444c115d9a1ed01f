"""HipBatchAM — the AcquisitionMaximizer of the plugin trio.

Reproduces SamplingAM (src/acquisition_maximizers/sampling.jl:20-75: draw M candidates from
x_prior inside the domain, acq on each, arg-max) and GridAM (grid.jl:52-65: a fixed list of
points) with ONE device call for all candidates: posterior mean/variance for every output and
hyper-parameter sample, analytic EI × feasibility, make_safe mask, arg-max
(construct_acquisition(::ExpectedImprovement), src/acquisitions/expected_improvement.jl:49-90).

Multi-GPU (one process per GPU), selected by `shard`:
  "candidates"  every rank holds the (redundantly factorised) posterior and evaluates its
                contiguous shard of the candidates; the per-rank (max, global index) pairs are
                combined with a single 16-byte all-gather over RCCL;
  "outputs"     output i is factorised and predicted on rank i mod G only; the (mu_i, var_i) rows
                are all-gathered and the EI x feasibility epilogue + arg-max run from those moments;
  "samples"     hyper-parameter sample s (BI) is factorised on its shard's rank only; the partial
                sums of acq_s(x_j) are combined with one all-reduce(sum) of M doubles.
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


def posteriors_of(problem: BossProblem, lo: Optional[int] = None, hi: Optional[int] = None, reserve: int = 0):
    """model_posterior(problem) (src/posterior.jl:2-3): list over BI samples of posteriors.
    [lo, hi) restricts the construction to a shard of the samples; reserve: room for later appends."""
    params = problem.params
    if lo is not None:
        params = list(params)[lo:hi]
        if not params:
            return []
    post = problem.model.model_posterior(params, problem.data, reserve) if reserve else \
        problem.model.model_posterior(params, problem.data)
    return post if isinstance(post, list) else [post]


def n_samples(problem: BossProblem) -> int:
    return len(problem.params) if isinstance(problem.params, (list, tuple)) else 1


def output_moments(problem: BossProblem, i: int, Xs: np.ndarray) -> np.ndarray:
    """(mu_i, var_i) of output i at all candidates for every hyper-parameter sample: [S][2][M].
    Only the owner rank of output i factorises it (model_posterior_slice, gaussian_process.jl:133-141)."""
    plist = list(problem.params) if isinstance(problem.params, (list, tuple)) else [problem.params]
    out = np.empty((len(plist), 2, Xs.shape[1]))
    for s, prm in enumerate(plist):
        sl = problem.model.model_posterior_slice(prm, problem.data, i)
        out[s, 0], out[s, 1] = sl.mean_and_var(Xs)
        sl.gp.close()
    return out


def moments_acquisition(problem: BossProblem, mu: np.ndarray, var: np.ndarray, Xs: np.ndarray):
    """EI x feasibility + arg-max on the device from gathered moments mu/var [S][P][M]."""
    ei = problem.acquisition
    b = best_so_far(ei.fitness, problem.data.Y, problem.y_max)
    mask = None
    if ei.cons_safe:
        mask = in_bounds(Xs, problem.domain.bounds) & in_cons(Xs, problem.domain.cons)
    return api.acq_ei_moments(mu, var, ei.fitness.coefs, problem.y_max, b, mask, problem.model.device)


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
    shard: str = "candidates"           # "candidates" | "outputs" | "samples"
    fused: bool = True                  # one output, one parameter sample, LinFitness: the posterior update and the acquisition in ONE device
    #                                     call (boss_gp_update_acq: the candidates' substitution rides along the factorisation)
    devices: Optional[Sequence[int]] = None   # ONE process driving GPUs 0..G-1: the shard modes run inside the library
    #                                           (boss_multi_*: one host thread per device, exchanges over RCCL) — what a Julia
    #                                           caller uses; None = one process per GPU over torch.distributed (or a single GPU)

    def candidates(self, problem: BossProblem) -> np.ndarray:
        if self.points is not None:
            return np.asfortranarray(self.points, dtype=np.float64)
        rng = np.random.default_rng(dist_util.shared_seed(self.seed, self.group))     # the same draws on every rank
        xs = [_rand_in_domain(self.x_prior, problem.domain, rng, self.max_attempts) for _ in range(self.samples)]
        xs = [x for x in xs if x is not None]                          # _reduce_samples (sampling.jl:77-79)
        if not xs:
            raise RuntimeError("HipBatchAM: No samples were successfully drawn! Check the `x_prior` and the `Domain`.")
        return np.asfortranarray(np.stack(xs, axis=1))

    def maximize_acquisition(self, problem: BossProblem, options: BossOptions = BossOptions(), return_all: bool = False,
                             posts=None):
        """maximize_acquisition(::SamplingAM, problem, options) (sampling.jl:20-40) -> (x, val).
        posts: already-built posteriors of `problem` (HipSequentialBatchAM keeps them resident and
        extends them by block Cholesky appends); default: build them from problem.params."""
        Xs = self.candidates(problem)                                   # identical on every rank (seeded)
        M = Xs.shape[1]
        rank, world = dist_util.rank_world(self.group)
        if self.devices is not None:
            assert world == 1 and posts is None, "devices=[...] is the single-process mode"
            return self._maximize_in_library(problem, Xs, return_all)
        if self.shard == "outputs":
            return self._maximize_by_outputs(problem, Xs, rank, world, return_all)
        if self.shard == "samples":
            return self._maximize_by_samples(problem, Xs, rank, world, return_all)
        assert self.shard == "candidates", self.shard
        lo, hi = dist_util.shard_range(M, rank, world)
        if posts is None and self.fused and hi > lo and self._fusable(problem):
            acq, am, mx = self._update_and_acquire(problem, Xs[:, lo:hi], return_all)
            am += lo
            if return_all:
                return Xs, (dist_util.allgather_concat(acq, self.group) if world > 1 else acq)
            mx, am = dist_util.argmax_exchange(mx, am, self.group)
            return Xs[:, am].copy(), mx
        if posts is None:
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

    @staticmethod
    def _fusable(problem: BossProblem) -> bool:
        """`model_posterior(problem)` is ONE factorisation (one output, one parameter sample) of the plain GP / semiparametric
        model and the acquisition is the analytic EI: the two halves of the BO iteration (src/bo.jl:30-48) fit one device call."""
        from .model import HipGaussianProcess
        return (problem is not None and type(problem.model) is HipGaussianProcess and problem.data.Y.shape[0] == 1 and
                not isinstance(problem.params, (list, tuple)) and not isinstance(problem.acquisition.fitness, NonlinFitness))

    def _update_and_acquire(self, problem: BossProblem, Xs, want_acq: bool):
        """model_posterior(problem) + acq.(eachcol(xs)) + argmax as boss_gp_update_acq: the posterior of the single output is
        built under problem.params while the candidates' forward substitution rides along its factorisation."""
        model, prm, data, ei = problem.model, problem.params, problem.data, problem.acquisition
        Xs = np.asfortranarray(Xs, dtype=np.float64)
        mask = (in_bounds(Xs, problem.domain.bounds) & in_cons(Xs, problem.domain.cons)) if ei.cons_safe else None
        b = best_so_far(ei.fitness, data.Y, problem.y_max)
        g = api.GP(data.X, data.Y[0], model.kernel, model.discrete, model.device)
        cand = api.Candidates(Xs, model.device)
        try:
            r = g.update_acq(prm.lengthscales[:, 0], prm.amplitudes[0], prm.noise_std[0], cand, fit_coef=float(np.asarray(ei.fitness.coefs)[0]),
                             y_max=float(np.asarray(problem.y_max, float)[0]), best=b, mean_X=model.mean_values(data.X, prm, 0),
                             mean_Xs=model.mean_values(Xs, prm, 0), valid_mask=mask, want_acq=want_acq)
        finally:
            cand.close()
            g.close()
        return r["acq"], r["argmax"], r["max"]

    def _maximize_in_library(self, problem: BossProblem, Xs, return_all):
        """The three shard modes with ONE process driving the devices: posteriors are created on the devices the mode
        assigns them to and the library shards / exchanges itself (boss_multi_acq_ei*)."""
        import dataclasses
        ei = problem.acquisition
        if isinstance(ei.fitness, NonlinFitness):
            raise NotImplementedError("devices=[...] needs the analytic EI of LinFitness")
        devs = list(self.devices)
        G = len(devs)
        if devs != list(range(G)) or api.init() < G:
            raise ValueError("devices must be [0, ..., G-1] with G <= the number of visible GPUs")
        P, S, M = problem.data.Y.shape[0], n_samples(problem), Xs.shape[1]
        plist = list(problem.params) if isinstance(problem.params, (list, tuple)) else [problem.params]
        on = lambda g: dataclasses.replace(problem.model, device=g)
        b = best_so_far(ei.fitness, problem.data.Y, problem.y_max)
        mask = (in_bounds(Xs, problem.domain.bounds) & in_cons(Xs, problem.domain.cons)) if ei.cons_safe else None
        means = None
        if problem.model.mean_values(Xs[:, :1], plist[0], 0) is not None:
            means = np.stack([np.stack([problem.model.mean_values(Xs, prm, i) for i in range(P)]) for prm in plist])   # S×P×M
        slices = []                                                      # everything to close afterwards
        try:
            if self.shard == "candidates":                               # replicas on every device, columns split M/G
                reps = []
                for g in range(G):
                    row = [[on(g).model_posterior_slice(prm, problem.data, i) for i in range(P)] for prm in plist]
                    slices += [sl for r in row for sl in r]
                    reps.append([[sl.gp for sl in r] for r in row])
                acq, am, mx = api.multi_acq_ei(reps, Xs, ei.fitness.coefs, problem.y_max, b, mask, means)
            elif self.shard == "outputs":                                # output i on device i mod G
                row = [[on(dist_util.owner_of(i, G)).model_posterior_slice(prm, problem.data, i) for i in range(P)] for prm in plist]
                slices += [sl for r in row for sl in r]
                acq, am, mx = api.multi_acq_ei_outputs([[sl.gp for sl in r] for r in row], Xs, ei.fitness.coefs, problem.y_max, b, mask, means)
            else:                                                        # sample s on the device of its shard
                assert self.shard == "samples", self.shard
                row = []
                for g in range(G):
                    lo, hi = dist_util.shard_range(S, g, G)
                    row += [[on(g).model_posterior_slice(prm, problem.data, i) for i in range(P)] for prm in plist[lo:hi]]
                slices += [sl for r in row for sl in r]
                acq, am, mx = api.multi_acq_ei_samples([[sl.gp for sl in r] for r in row], Xs, ei.fitness.coefs, problem.y_max, b, mask, means)
        finally:
            for sl in slices:
                sl.close()
        if return_all:
            return Xs, acq
        return Xs[:, am].copy(), mx

    # (the two modes below factorise only a shard of the outputs / samples on each rank)
    def _maximize_by_outputs(self, problem: BossProblem, Xs, rank, world, return_all):
        if isinstance(problem.acquisition.fitness, NonlinFitness):
            raise NotImplementedError("shard='outputs' needs the analytic EI of LinFitness; use shard='candidates'")
        P, S, M = problem.data.Y.shape[0], n_samples(problem), Xs.shape[1]
        mine = {i: output_moments(problem, i, Xs) for i in range(P) if dist_util.owner_of(i, world) == rank}
        rows = dist_util.allgather_owned(mine, P, (S, 2, M), self.group)        # [P][S][2][M]
        mu = np.ascontiguousarray(rows[:, :, 0, :].transpose(1, 0, 2))          # [S][P][M]
        var = np.ascontiguousarray(rows[:, :, 1, :].transpose(1, 0, 2))
        acq, am, mx = moments_acquisition(problem, mu, var, Xs)                  # identical on every rank
        if return_all:
            return Xs, acq
        return Xs[:, am].copy(), mx

    def _maximize_by_samples(self, problem: BossProblem, Xs, rank, world, return_all):
        if isinstance(problem.acquisition.fitness, NonlinFitness):
            raise NotImplementedError("shard='samples' needs the analytic EI of LinFitness; use shard='candidates'")
        S, M = n_samples(problem), Xs.shape[1]
        lo, hi = dist_util.shard_range(S, rank, world)
        part = np.zeros(M)
        if hi > lo:
            posts = posteriors_of(problem, lo, hi)
            acq, _, _ = acquisition_values(problem, posts, Xs)                   # mean over the local samples
            part = acq * (hi - lo)
        acq = dist_util.allreduce_sum(part, self.group) / S                      # (:87-90) mean over all samples
        if return_all:
            return Xs, acq
        am = int(np.argmax(acq))                                                 # Julia argmax: first maximum, NaN largest
        return Xs[:, am].copy(), float(acq[am])


@dataclass
class HipSequentialBatchAM:
    """SequentialBatchAM(am, batch_size) (src/acquisition_maximizers/batch.jl:18-38): select
    `batch_size` candidates one after the other, augmenting a COPY of the problem with the
    'speculative' observation (x, posterior mean at x) after each selection.

    The reference rebuilds the posterior from scratch (an O(N^3) Cholesky) for every selected
    point; here the posteriors stay resident and each speculative observation is a block Cholesky
    append (boss_gp_append, O(N^2)) under the unchanged hyper-parameters — same posterior.  When the
    inner maximiser's candidate set is fixed (`points`, or a seeded sample) the candidates' predictive
    state stays resident too (boss_track_*) and is extended in O(N·M) per selection."""
    am: HipBatchAM
    batch_size: int

    def maximize_acquisition(self, problem: BossProblem, options: BossOptions = BossOptions()):
        import copy
        prob = copy.copy(problem)                                        # problem_ = deepcopy(problem) (batch.jl:27)
        prob.data = type(problem.data)(problem.data.X.copy(), problem.data.Y.copy())
        posts = posteriors_of(prob, reserve=self.batch_size)             # no re-allocation when the appends arrive
        fixed = self.am.points is not None or self.am.seed is not None   # same candidate set at every selection
        lin = not isinstance(prob.acquisition.fitness, NonlinFitness)
        try:
            if fixed and lin and self.am.shard == "candidates":
                return self._tracked(prob, posts), None
            xs = []
            for _ in range(self.batch_size):                             # speculative_evaluation! (batch.jl:32-38)
                x, _ = self.am.maximize_acquisition(prob, options, posts=posts)
                y = sum(p.mean(x) for p in posts) / len(posts)           # mean(post, x); BI: average_mean
                prob.augment_dataset(x, y)
                for p in posts:
                    p.append(x, y)
                xs.append(x)
            return np.stack(xs, axis=1), None
        finally:
            for p in posts:
                p.close()

    def _tracked(self, prob: BossProblem, posts):
        """Fixed candidate set: the candidates' V slabs stay resident (api.Track) and every speculative
        observation extends them by one row — O(N·M) per selection instead of the O(N²M) re-solve."""
        Xs = self.am.candidates(prob)                                    # identical on every rank
        M = Xs.shape[1]
        rank, world = dist_util.rank_world(self.am.group)
        lo, hi = dist_util.shard_range(M, rank, world)
        ei = prob.acquisition
        dev = prob.model.device
        cand = api.Candidates(Xs[:, lo:hi], dev) if hi > lo else None
        tracks = []
        if cand is not None:
            tracks = [[api.Track(s.gp, cand, s._mean_s(Xs[:, lo:hi])) for s in p.slices] for p in posts]
        mask = None
        if ei.cons_safe and hi > lo:
            mask = in_bounds(Xs[:, lo:hi], prob.domain.bounds) & in_cons(Xs[:, lo:hi], prob.domain.cons)
        xs = []
        try:
            for _ in range(self.batch_size):
                b = best_so_far(ei.fitness, prob.data.Y, prob.y_max)
                if cand is not None:
                    _, am, mx = api.acq_ei_tracks(tracks, ei.fitness.coefs, prob.y_max, b, mask, want_acq=False)
                    am += lo
                else:
                    am, mx = M, -np.inf
                mx, am = dist_util.argmax_exchange(mx, am, self.am.group)
                x = Xs[:, am].copy()
                yl = None
                if lo <= am < hi:                                        # the owner reads ŷ = mean(post, x) off its tracks
                    yl = sum(np.array([t.moments(am - lo, 1)[0][0] for t in ts]) for ts in tracks) / len(tracks)
                P = prob.data.Y.shape[0]                                 # one fixed-size broadcast from the owner
                y = dist_util.broadcast_array(yl, (P,), dist_util.owner_of_index(am, M, world), self.am.group) if world > 1 else yl
                prob.augment_dataset(x, y)
                for p in posts:
                    p.append(x, y)
                xs.append(x)
        finally:
            for ts in tracks:
                for t in ts:
                    t.close()
            if cand is not None:
                cand.close()
        return np.stack(xs, axis=1)


@dataclass
class HipGradientAM:
    """OptimizationAM semantics (src/acquisition_maximizers/optimization.jl:13-118): multistart LOCAL
    optimisation of the acquisition, started from `multistart` x_prior samples inside the domain, the
    best local optimum wins (optim_multistart.jl).  The reference differentiates the acquisition with
    ForwardDiff (optimization.jl:36) one start at a time; here the gradient is analytic and evaluated
    on the device (boss_acq_ei_grad) for ALL starts in one call per iteration: projected gradient
    ascent with a per-start Barzilai–Borwein step and backtracking.  Discrete dimensions are rounded
    (their gradient is zero), `cons` is honoured through make_safe (acq = 0 outside) and a final
    in-domain filter.  Multi-GPU: the starts are sharded across ranks, 16-byte arg-max exchange.
    `mean_grad`: optional x -> P×d Jacobian of the GP's prior mean (zero / constant means need none)."""
    x_prior: Callable = None
    multistart: int = 200
    iters: int = 30
    max_attempts: int = 200
    seed: Optional[int] = None
    group: object = None
    mean_grad: Optional[Callable] = None

    def _value_and_grad(self, problem: BossProblem, posts, X):
        ei = problem.acquisition
        if isinstance(ei.fitness, NonlinFitness):
            raise NotImplementedError("HipGradientAM needs the analytic EI of LinFitness")
        b = best_so_far(ei.fitness, problem.data.Y, problem.y_max)
        mask = (in_bounds(X, problem.domain.bounds) & in_cons(X, problem.domain.cons)) if ei.cons_safe else None
        acc, gacc = 0.0, 0.0
        for post in posts:                                   # BI: the acquisition (and its gradient) is the sample mean
            gps = [s.gp for s in post.slices]
            ms = None
            if post.slices[0]._mean_s(X) is not None:
                ms = np.stack([s._mean_s(X) for s in post.slices])
            mg = None
            if self.mean_grad is not None:
                J = np.stack([np.asarray(self.mean_grad(X[:, j]), float) for j in range(X.shape[1])], axis=2)   # P×d×M
                mg = J
            a, g = api.acq_ei_grad(gps, X, ei.fitness.coefs, problem.y_max, b, mask, ms, mg)
            acc, gacc = acc + a, gacc + g
        return acc / len(posts), gacc / len(posts)

    def maximize_acquisition(self, problem: BossProblem, options: BossOptions = BossOptions(), posts=None):
        rng = np.random.default_rng(dist_util.shared_seed(self.seed, self.group))     # the same starts on every rank
        dom = problem.domain
        starts = [_rand_in_domain(self.x_prior, dom, rng, self.max_attempts) for _ in range(self.multistart)]
        starts = [x for x in starts if x is not None]
        if not starts:
            raise RuntimeError("HipGradientAM: No samples were successfully drawn! Check the `x_prior` and the `Domain`.")
        X0 = np.asfortranarray(np.stack(starts, axis=1))
        rank, world = dist_util.rank_world(self.group)
        lo, hi = dist_util.shard_range(X0.shape[1], rank, world)
        if posts is None:
            posts = posteriors_of(problem)
        best_v, best_i, Xf = -np.inf, X0.shape[1], None
        if hi > lo:
            X = X0[:, lo:hi].copy()
            lb, ub = dom.bounds
            span = np.where(np.isfinite(ub - lb), ub - lb, 1.0)[:, None]
            cont = ~dom.discrete
            f, g = self._value_and_grad(problem, posts, X)
            step = np.full(X.shape[1], 0.05)                 # relative to the box, per start
            for _ in range(self.iters):
                gn = g * span
                gn[~cont] = 0.0
                nrm = np.maximum(np.sqrt((gn * gn).sum(0)), 1e-300)
                Xn = X + (gn / nrm) * span * step[None, :]
                Xn = np.minimum(np.maximum(Xn, lb[:, None]), ub[:, None])
                fn, gnw = self._value_and_grad(problem, posts, Xn)
                better = fn > f
                X[:, better], f[better], g[:, better] = Xn[:, better], fn[better], gnw[:, better]
                step = np.where(better, np.minimum(step * 1.5, 0.5), step * 0.4)
                if np.all(step < 1e-7):
                    break
            ok = in_domain(X, dom)
            fm = np.where(ok, f, -np.inf)
            j = int(np.argmax(fm))
            best_v, best_i, Xf = float(fm[j]), lo + j, X
        best_v, gi = dist_util.argmax_exchange(best_v, best_i, self.group)
        if world > 1:                                        # the winner's refined point: one d-double broadcast from its owner
            owner = dist_util.owner_of_index(gi, X0.shape[1], world)
            mine = Xf[:, gi - lo] if owner == rank else None
            return dist_util.broadcast_array(mine, (X0.shape[0],), owner, self.group), best_v
        return Xf[:, gi - lo].copy(), best_v

"""HipBatchedMAP — the ModelFitter of the plugin trio: SamplingMAP semantics
(src/model_fitters/sampling.jl:17-78: draw S prior samples, evaluate loglike on each, keep the
arg-max) with the S data log-likelihoods evaluated in ONE batched device call per output
(boss_gp_loglike_batch) instead of S sequential Cholesky factorisations.

With torch.distributed initialised (one process per GPU) the S samples are sharded across ranks
with no data-path collective; the per-rank best (loglike, index) pairs are combined with one
all_gather (see distributed.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import distributed as dist_util
from .model import HipGaussianProcess, HipGPParams
from .problem import BossOptions, BossProblem


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
        rng = np.random.default_rng(self.seed)
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

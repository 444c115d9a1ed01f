"""One BO iteration with the plugin trio — the harness equivalent of the body of `bo!`
(src/bo.jl:30-48): estimate_parameters! -> maximize_acquisition -> eval_objective! ->
augment_dataset!.  The reference's loop itself is out of scope; this mirrors
just enough of it to exercise the trio the way an unmodified `bo!` would."""
from __future__ import annotations

import numpy as np

from .problem import BossOptions, BossProblem


def estimate_parameters(problem: BossProblem, fitter, options: BossOptions = BossOptions()):
    """estimate_parameters! (src/bo.jl:80-84) + update_parameters! (src/types/problem.jl:177-182)."""
    fitted = fitter.estimate_parameters(problem, options)
    problem.params = fitted.params
    problem.consistent = True
    return fitted


def bo_step(problem: BossProblem, fitter, maximizer, options: BossOptions = BossOptions()):
    if not problem.consistent or problem.params is None:
        estimate_parameters(problem, fitter, options)
    x, val = maximizer.maximize_acquisition(problem, options)
    if problem.f is not None:                                   # f === missing -> recommender mode (bo.jl:50-59)
        y = np.asarray(problem.f(x), float).reshape(-1)
        problem.augment_dataset(x[:, None], y[:, None])
    return x, val


def bo(problem: BossProblem, fitter, maximizer, iters: int, options: BossOptions = BossOptions()):
    """IterLimit(iters) loop (src/bo.jl:40-48)."""
    estimate_parameters(problem, fitter, options)
    for _ in range(iters):
        bo_step(problem, fitter, maximizer, options)
        estimate_parameters(problem, fitter, options)
    return problem

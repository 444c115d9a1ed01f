"""boss.jl_amd — MI355X-native GP-posterior + acquisition hot path for BOSS.jl.

Holds only what the path needs: csrc/ (HIP kernels + the C ABI of include/bosship.h), api.py
(ctypes twin of the Julia ccall layer) and the host-side mirror of BOSS's plugin interface
(model.py / fitter.py / maximizer.py).  The directory name contains a dot, so import it through
the repo-root shim:  `import boss_jl_amd`.
"""
from . import api  # noqa: F401
from .api import (BossError, Candidates, DomainError, GP, PosDefException, acq_ei, fit,  # noqa: F401
                  load_library, loglike_batch)
from .problem import (BossOptions, BossProblem, Dirac, Domain, ExperimentData, ExpectedImprovement,  # noqa: F401,E402
                      LinFitness, LogNormal, MvDirac, MvLogNormal, NonlinFitness)
from .model import HipGaussianProcess, HipGPParams, average_mean  # noqa: F401,E402
from .gradient_gp import (GradientData, HipGradientGaussianProcess, HipGradientGPParams,  # noqa: F401,E402
                          join_gradient_slices)
from .nonstationary import HipNonstationaryGP, HipParametrizedGP, stack_latents  # noqa: F401,E402
from .fitter import HipBatchedMAP, HipGradientMAP, HipSampleOptMAP, MAPParams  # noqa: F401,E402
from .maximizer import HipBatchAM, HipGradientAM, HipSequentialBatchAM  # noqa: F401,E402

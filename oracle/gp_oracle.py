"""CPU oracle for the BOSS.jl GP-posterior + acquisition hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product path (boss.jl_amd/) never
imports anything from oracle/ and fails loudly when the HIP library is missing.

It is a plain numpy/scipy (OpenBLAS LAPACK) fp64 restatement of what the reference computes
on this path.  Every function cites the reference file:line it follows (paths relative to
/root/reference).  The reference delegates the arithmetic to packages that are NOT vendored
under /root/reference:

    AbstractGPs.jl   compat 0.5.21   (Project.toml:25)   posterior / logpdf / mean_and_var
    KernelFunctions.jl (transitive, unpinned)            Matern32/52, SqExponential, ARD
    Distances.jl       (transitive)                      pairwise Euclidean
    Distributions.jl 0.25.109 / StatsFuns                Normal cdf/pdf

so those parts restate the packages' published algorithms, anchored on the reference's call
sites (src/models/gaussian_process.jl:199-248,279) and on src/models/gradient_gp.jl:323-361,403
which spells the same algebra out in-repo.

PARITY PIN STATUS: the reference is Julia; no Julia toolchain exists in this pipeline and the
reference's tests hold NO golden posterior numbers (they are property / known-answer tests).
The oracle is therefore pinned by (i) every known-answer and property test the reference holds
for this path (tests/test_oracle_reference_pins.py), (ii) an independent 50-digit mpmath
evaluation of the same formulas, (iii) scikit-learn's GaussianProcessRegressor as an independent
third-party implementation.  Absolute numeric parity with AbstractGPs beyond that is UNPINNED.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import scipy.linalg as sla
from scipy.special import erfc

# src/models/gaussian_process.jl:5   const MIN_PARAM_VALUE = 1e-8
MIN_PARAM_VALUE = 1e-8
# src/models/gaussian_process.jl:13  const MAX_NEG_VAR = 1e-8
MAX_NEG_VAR = 1e-8
# AbstractGPs: `post(X*)` without a noise argument builds FiniteGP(f, x, Σy=1e-18)
# (AbstractGPs/src/finite_gp_projection.jl default_σ² = 1e-18) -> var(post(X*)) adds it.
PREDICT_JITTER = 1e-18

MATERN32, MATERN52, SQEXP = 0, 1, 2
KERNEL_NAMES = {"matern32": MATERN32, "matern52": MATERN52, "sqexp": SQEXP}

_SQRT3 = math.sqrt(3.0)
_SQRT5 = math.sqrt(5.0)
_LOG2PI = math.log(2.0 * math.pi)


class DomainError(ValueError):
    """Mirror of Julia's DomainError thrown by `_clip_var` (gaussian_process.jl:191-193)."""


class PosDefException(np.linalg.LinAlgError):
    """Mirror of LinearAlgebra.PosDefException raised by `cholesky` on a non-PD matrix."""


# --------------------------------------------------------------------------------------
# kernels  (KernelFunctions.jl: Matern32Kernel / Matern52Kernel / SqExponentialKernel;
#           reference use: src/deprecated.jl:34 default Matern52, examples/example.jl:88 Matern32,
#           src/models/nonstationary_gp/nonstationary_gp.jl:438 GaussianKernel)
# --------------------------------------------------------------------------------------
def kappa(kernel: int, r: np.ndarray) -> np.ndarray:
    """Radial profile of the base kernel on Euclidean distance r (KernelFunctions `kappa`)."""
    if kernel == MATERN32:
        s = _SQRT3 * r
        return (1.0 + s) * np.exp(-s)
    if kernel == MATERN52:
        s = _SQRT5 * r
        return (1.0 + s + s * s / 3.0) * np.exp(-s)
    if kernel == SQEXP:
        return np.exp(-0.5 * r * r)
    raise ValueError(f"unknown kernel id {kernel}")


def discrete_round(X: np.ndarray, discrete: Optional[Sequence[bool]]) -> np.ndarray:
    """src/utils/utils.jl `discrete_round` + src/models/utils/kernels.jl:56-59 (DiscreteKernel):
    flagged dims of BOTH kernel arguments are rounded (Julia `round` = half-to-even = np.rint)."""
    if discrete is None:
        return X
    discrete = np.asarray(discrete, dtype=bool)
    if not discrete.any():
        return X
    X = np.array(X, dtype=np.float64, copy=True)
    X[discrete, :] = np.rint(X[discrete, :])
    return X


def scaled_distance(Xa: np.ndarray, Xb: np.ndarray, lengthscale: np.ndarray,
                    form: str = "direct") -> np.ndarray:
    """Pairwise Euclidean distance of ARD-scaled columns, r_ij = ||(xa_i - xb_j) ./ λ||.

    `with_lengthscale(kernel, λ::Vector)` = kernel ∘ ARDTransform(1 ./ λ)
    (gaussian_process.jl:243), i.e. inputs are multiplied by inv(λ) first.
    form="direct": sum of squared differences (what the HIP kernels do; most accurate).
    form="gemm":   ||a||²+||b||²-2a·b, clamped at 0 — the Distances.jl `pairwise(Euclidean())`
                   formulation KernelFunctions uses for ColVecs inputs (threshold 0).
    The two differ by O(eps·||x/λ||²) in r²; tests quantify it.
    """
    inv = 1.0 / np.asarray(lengthscale, dtype=np.float64)
    A = np.asarray(Xa, dtype=np.float64) * inv[:, None]
    B = np.asarray(Xb, dtype=np.float64) * inv[:, None]
    if form == "direct":
        r2 = np.zeros((A.shape[1], B.shape[1]))
        for k in range(A.shape[0]):
            diff = A[k][:, None] - B[k][None, :]
            r2 += diff * diff
        return np.sqrt(r2)
    if form == "gemm":
        sa = np.sum(A * A, axis=0)
        sb = np.sum(B * B, axis=0)
        r2 = sa[:, None] + sb[None, :] - 2.0 * (A.T @ B)
        return np.sqrt(np.maximum(r2, 0.0))
    raise ValueError(form)


@dataclass
class GPHyper:
    """Hyper-parameters of one output slice AFTER the +1e-8 offsets (gaussian_process.jl:239-241)."""
    kernel: int
    lengthscale: np.ndarray
    amplitude: float
    noise_std: float
    discrete: Optional[np.ndarray] = None


def finite_gp_params(kernel, d, lengthscale, amplitude, noise_std, discrete=None) -> GPHyper:
    """src/models/gaussian_process.jl:216-245 `finite_gp`: validate, then ADD 1e-8 to λ, α, σ."""
    if isinstance(kernel, str):
        kernel = KERNEL_NAMES[kernel]
    lengthscale = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64))
    # :227-229  @assert all(lengthscales .>= 0); amplitude >= 0; noise_std >= 0
    assert np.all(lengthscale >= 0), "lengthscales must be >= 0"
    assert amplitude >= 0, "amplitude must be >= 0"
    assert noise_std >= 0, "noise_std must be >= 0"
    # :233  @assert length(lengthscales) == size(X, 1)
    assert lengthscale.shape[0] == d, "length(lengthscales) must equal x_dim"
    return GPHyper(kernel, lengthscale + MIN_PARAM_VALUE, float(amplitude) + MIN_PARAM_VALUE,
                   float(noise_std) + MIN_PARAM_VALUE,
                   None if discrete is None else np.asarray(discrete, dtype=bool))


def kernelmatrix(h: GPHyper, Xa: np.ndarray, Xb: Optional[np.ndarray] = None,
                 form: str = "direct") -> np.ndarray:
    """k(x,x') = α² κ(||(x-x')./λ||)  (gaussian_process.jl:243: `(amplitude^2) * with_lengthscale`)."""
    Xa = discrete_round(np.asarray(Xa, dtype=np.float64), h.discrete)
    Xb = Xa if Xb is None else discrete_round(np.asarray(Xb, dtype=np.float64), h.discrete)
    r = scaled_distance(Xa, Xb, h.lengthscale, form)
    return (h.amplitude ** 2) * kappa(h.kernel, r)


def kernelmatrix_blas(h: GPHyper, X: np.ndarray) -> np.ndarray:
    """kernelmatrix(k, X) the way the reference's stack evaluates it (src/models/utils/kernels.jl:35 prints the
    metric: Distances.Euclidean): pairwise distances as ‖a‖²+‖b‖²−2a·b with the cross term from ONE BLAS-3 call
    (dgemm, multithreaded), then the radial profile broadcast over the N×N matrix — in place, so that a timing of it
    measures arithmetic and not temporaries.  Same values as kernelmatrix(..., form="gemm") up to the order of the
    three additions.  Used by bench.py's cpu_baseline leg."""
    Xa = discrete_round(np.asarray(X, dtype=np.float64), h.discrete)
    A = Xa * (1.0 / h.lengthscale)[:, None]
    sa = np.einsum("ij,ij->j", A, A)
    R = A.T @ A                                   # dgemm
    R *= -2.0
    R += sa[:, None]
    R += sa[None, :]
    np.maximum(R, 0.0, out=R)
    amp2 = h.amplitude ** 2
    if h.kernel == SQEXP:
        R *= -0.5
        np.exp(R, out=R)
        R *= amp2
        return R
    np.sqrt(R, out=R)
    R *= _SQRT3 if h.kernel == MATERN32 else _SQRT5       # s
    E = np.exp(-R)
    if h.kernel == MATERN32:
        R += 1.0                                          # 1 + s
    else:
        T = R * R
        T *= 1.0 / 3.0
        R += T
        R += 1.0                                          # 1 + s + s²/3
    R *= E
    R *= amp2
    return R


# --------------------------------------------------------------------------------------
# posterior construction   (gaussian_process.jl:199-211 -> AbstractGPs.posterior(fx, y))
# --------------------------------------------------------------------------------------
@dataclass
class GPPosterior:
    """What AbstractGPs.PosteriorGP stores: (α = C\\δ, C, x, δ) + the prior definition."""
    h: GPHyper
    X: np.ndarray            # d×N, one observation per column (src/types/data.jl:10-12)
    L: np.ndarray            # lower Cholesky factor of K + σ²I
    a: np.ndarray            # (K+σ²I)^{-1} (y - m(X))
    delta: np.ndarray        # y - m(X)
    logpdf: float            # log marginal likelihood of y under the FiniteGP


def _mean_vec(mean, X: np.ndarray) -> np.ndarray:
    """Prior mean over columns: nothing -> 0, Real -> ConstMean, Function -> CustomMean mapped
    over columns (gaussian_process.jl:101-103,247-248); a precomputed vector is accepted as-is
    (that is the form in which user closures cross the C ABI)."""
    n = X.shape[1]
    if mean is None:
        return np.zeros(n)
    if callable(mean):
        return np.array([float(mean(X[:, j])) for j in range(n)], dtype=np.float64)
    m = np.asarray(mean, dtype=np.float64)
    if m.ndim == 0:
        return np.full(n, float(m))
    assert m.shape == (n,)
    return m


def gp_fit(X, y, kernel, lengthscale, amplitude, noise_std, mean=None, discrete=None,
           form: str = "direct", timings: Optional[dict] = None) -> GPPosterior:
    """posterior_gp (gaussian_process.jl:199-211): K = kernelmatrix + σ²I; C = cholesky(K);
    δ = y - m(X); a = C \\ δ.  logpdf(FiniteGP, y) = -(N log 2π + logdet C + ||C.U' \\ δ||²)/2
    (gaussian_process.jl:279; algebra spelled out in gradient_gp.jl:325-326,403)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    d, N = X.shape
    assert y.shape[0] == N
    import time
    t0 = time.perf_counter()
    h = finite_gp_params(kernel, d, lengthscale, amplitude, noise_std, discrete)
    K = kernelmatrix_blas(h, X) if form == "blas" else kernelmatrix(h, X, form=form)
    K[np.diag_indices(N)] += h.noise_std ** 2
    t1 = time.perf_counter()
    if timings is not None:
        # bench.py's cpu_baseline leg: LAPACK dpotrf IN PLACE, as Julia's cholesky(Symmetric(K)) does on its copy.  K is symmetric and
        # C-ordered, so K.T is the same matrix as a Fortran-ordered view: no conversion copy, no zeroing of the other triangle
        # (scipy.linalg.cholesky spends more time on those two passes than on the factorisation); the consumers below read the
        # lower triangle only.
        from scipy.linalg.lapack import dpotrf
        L, info = dpotrf(K.T, lower=1, overwrite_a=1, clean=0)
        if info != 0:
            raise PosDefException(f"{info}-th leading minor not positive definite")
    else:
        try:
            L = sla.cholesky(K, lower=True, check_finite=False)
        except np.linalg.LinAlgError as e:  # PosDefException in Julia
            raise PosDefException(str(e))
    t2 = time.perf_counter()
    delta = y - _mean_vec(mean, X)
    z = sla.solve_triangular(L, delta, lower=True, check_finite=False)
    a = sla.solve_triangular(L, z, lower=True, trans="T", check_finite=False)
    logdet = 2.0 * float(np.sum(np.log(np.diag(L))))
    logpdf = -0.5 * (N * _LOG2PI + logdet + float(z @ z))
    if timings is not None:                  # bench.py's cpu_baseline leg: where one posterior update spends its time
        timings["gram"] = timings.get("gram", 0.0) + (t1 - t0)
        timings["potrf"] = timings.get("potrf", 0.0) + (t2 - t1)
        timings["trsv_logdet"] = timings.get("trsv_logdet", 0.0) + (time.perf_counter() - t2)
    return GPPosterior(h, X, L, a, delta, logpdf)


def gp_data_loglike_slice(X, y, kernel, lengthscale, amplitude, noise_std, mean=None,
                          discrete=None) -> float:
    """gp_data_loglike_slice (gaussian_process.jl:269-280); -Inf on a non-PD matrix is what
    `safe_data_loglike` (src/surrogate_model.jl:2-12) turns the PosDefException into."""
    try:
        return gp_fit(X, y, kernel, lengthscale, amplitude, noise_std, mean, discrete).logpdf
    except PosDefException:
        return -math.inf


def data_loglike(X, Y, kernel, lengthscales, amplitudes, noise_stds, means=None,
                 discrete=None) -> float:
    """data_loglike (gaussian_process.jl:250-267): sum of per-output slices.
    Y is P×N, lengthscales d×P, amplitudes/noise_stds length P; means: None or P entries."""
    Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
    lengthscales = np.asarray(lengthscales, dtype=np.float64)
    total = 0.0
    for i in range(Y.shape[0]):
        m = None if means is None else means[i]
        total += gp_data_loglike_slice(X, Y[i], kernel, lengthscales[:, i], amplitudes[i],
                                       noise_stds[i], m, discrete)
    return total


# --------------------------------------------------------------------------------------
# prediction   (gaussian_process.jl:143-194 -> AbstractGPs mean_and_var(post(X*)))
# --------------------------------------------------------------------------------------
def gp_data_loglike_grad(X, y, kernel, lengthscale, amplitude, noise_std, mean=None, discrete=None):
    """logpdf(FiniteGP, y) (gp_data_loglike_slice, gaussian_process.jl:269-280) and its gradient w.r.t. the
    INPUT hyper-parameters (λ_1..λ_d, α, σ) — what ForwardDiff / Zygote produce when OptimizationMAP
    maximises the likelihood (src/model_fitters/optimization.jl:146-164).  With K = α²κ + σ²I (after the
    +1e-8 offsets), a = K⁻¹(y−m), G = a aᵀ − K⁻¹:
        ∂ℓ/∂θ = ½ Σ_ij G_ij ∂K_ij/∂θ ,
        ∂K/∂λ_m = −α² h(r) Δ_m² / λ_m³ ,  ∂K/∂α = 2α κ ,  ∂K/∂σ = 2σ I .
    Returns (logpdf, grad[d+2])."""
    X = np.asarray(X, dtype=np.float64)
    d, N = X.shape
    post = gp_fit(X, y, kernel, lengthscale, amplitude, noise_std, mean=mean, discrete=discrete)
    h = post.h
    Xa = discrete_round(X, h.discrete)
    r = scaled_distance(Xa, Xa, h.lengthscale)
    amp2 = h.amplitude ** 2
    Kinv = sla.cho_solve((post.L, True), np.eye(N), check_finite=False)
    G = np.outer(post.a, post.a) - Kinv
    Q = amp2 * kappa_prime_over_r(h.kernel, r)
    grad = np.zeros(d + 2)
    for m in range(d):
        diff = Xa[m][:, None] - Xa[m][None, :]
        grad[m] = 0.5 * np.sum(G * (-Q * diff * diff / h.lengthscale[m] ** 3))
    grad[d] = 0.5 * np.sum(G * (2.0 * h.amplitude * kappa(h.kernel, r)))
    grad[d + 1] = 0.5 * np.trace(G) * 2.0 * h.noise_std
    return post.logpdf, grad


def clip_var(var, threshold: float = MAX_NEG_VAR):
    """_clip_var (gaussian_process.jl:186-194): >=0 keep; [-thr,0) -> 0; < -thr -> DomainError."""
    v = np.asarray(var, dtype=np.float64)
    bad = v < -threshold
    if np.any(bad):
        first = float(v.reshape(-1)[np.argmax(bad.reshape(-1))])
        raise DomainError(
            f"The posterior GP predicted variance {first} but only values above -{threshold} are tolerated.")
    out = np.where(v >= 0.0, v, 0.0)
    return float(out) if np.ndim(var) == 0 else out


def gp_mean_and_var(post: GPPosterior, Xs, mean_s=None, clip: bool = True, form: str = "direct"):
    """mean_and_var(post, X::Matrix) (gaussian_process.jl:174-178):
    μ = m(X*) + K*ᵀ a ;  V = C.U' \\ K* ;  σ² = k(x*,x*) - Σ_i V_ij² + 1e-18 ; then _clip_var."""
    Xs = np.asarray(Xs, dtype=np.float64)
    if Xs.ndim == 1:
        Xs = Xs[:, None]
    Ks = kernelmatrix(post.h, post.X, Xs, form=form)            # N×M
    mu = _mean_vec(mean_s, Xs) + Ks.T @ post.a
    V = sla.solve_triangular(post.L, Ks, lower=True, check_finite=False)
    kss = (post.h.amplitude ** 2) * kappa(post.h.kernel, np.zeros(Xs.shape[1]))
    var = kss - np.sum(V * V, axis=0) + PREDICT_JITTER
    if clip:
        var = clip_var(var)
    return mu, var


def gp_mean_and_cov(post: GPPosterior, Xs, mean_s=None):
    """mean_and_cov (gaussian_process.jl:180-184): Σ = K** - VᵀV (+1e-18 I), diagonal clipped."""
    Xs = np.asarray(Xs, dtype=np.float64)
    Ks = kernelmatrix(post.h, post.X, Xs)
    mu = _mean_vec(mean_s, Xs) + Ks.T @ post.a
    V = sla.solve_triangular(post.L, Ks, lower=True, check_finite=False)
    Kss = kernelmatrix(post.h, Xs)
    S = Kss - V.T @ V + PREDICT_JITTER * np.eye(Xs.shape[1])
    S[np.diag_indices_from(S)] = clip_var(np.diag(S))
    return mu, S


def kappa_prime_over_r(kernel: int, r: np.ndarray) -> np.ndarray:
    """h(r) = κ'(r)/r of the radial profiles above (finite at r = 0):
    Matern32 -3 e^{-√3 r};  Matern52 -(5/3)(1+√5 r) e^{-√5 r};  SqExp -e^{-r²/2}.
    What ForwardDiff computes through KernelFunctions when the reference differentiates the
    acquisition (src/acquisition_maximizers/optimization.jl:36 AutoForwardDiff, :89-118)."""
    if kernel == MATERN32:
        return -3.0 * np.exp(-_SQRT3 * r)
    if kernel == MATERN52:
        s = _SQRT5 * r
        return -(5.0 / 3.0) * (1.0 + s) * np.exp(-s)
    if kernel == SQEXP:
        return -np.exp(-0.5 * r * r)
    raise ValueError(f"unknown kernel id {kernel}")


def gp_mean_and_var_grad(post: GPPosterior, Xs, mean_s=None, mean_grad_s=None):
    """Posterior mean / (unclipped) variance at the columns of Xs AND their gradients w.r.t. the
    candidate coordinates — the derivative of `mean_and_var(post, x)` (gaussian_process.jl:169-173):
        ∇μ(x*)  = ∇m(x*) + Σ_i a_i ∇k(x_i, x*) ,         a = (K+σ²I)⁻¹ (y − m)
        ∇σ²(x*) = −2 Σ_i w_i ∇k(x_i, x*) ,                w = (K+σ²I)⁻¹ k*      (k(x*,x*) is constant)
        ∇k(x_i, x*)_m = α² h(r_i) (x*_m − x_i,m) / λ_m²
    Dimensions flagged discrete are rounded inside the kernel (DiscreteKernel), so their gradient is 0.
    Returns mu[M], var[M] (with the 1e-18 jitter, not clipped), dmu[d,M], dvar[d,M]."""
    Xs = np.asarray(Xs, dtype=np.float64)
    if Xs.ndim == 1:
        Xs = Xs[:, None]
    h = post.h
    Xa = discrete_round(post.X, h.discrete)
    Xb = discrete_round(Xs, h.discrete)
    r = scaled_distance(Xa, Xb, h.lengthscale)                       # N×M
    amp2 = h.amplitude ** 2
    Ks = amp2 * kappa(h.kernel, r)
    Q = amp2 * kappa_prime_over_r(h.kernel, r)                       # N×M
    V = sla.solve_triangular(post.L, Ks, lower=True, check_finite=False)
    W = sla.solve_triangular(post.L, V, lower=True, trans="T", check_finite=False)    # (K+σ²I)⁻¹ k*
    m = np.zeros(Xs.shape[1]) if mean_s is None else np.asarray(mean_s, dtype=np.float64)
    mu = m + Ks.T @ post.a
    var = amp2 * kappa(h.kernel, np.zeros(Xs.shape[1])) - np.sum(V * V, axis=0) + PREDICT_JITTER
    lam2 = h.lengthscale ** 2
    d = Xs.shape[0]
    dmu = np.zeros((d, Xs.shape[1]))
    dvar = np.zeros((d, Xs.shape[1]))
    QA = Q * post.a[:, None]
    QW = Q * W
    for k in range(d):
        diff = Xb[k][None, :] - Xa[k][:, None]                       # x*_k − x_i,k   (N×M)
        dmu[k] = np.sum(QA * diff, axis=0) / lam2[k]
        dvar[k] = -2.0 * np.sum(QW * diff, axis=0) / lam2[k]
    if h.discrete is not None:
        dmu[h.discrete] = 0.0
        dvar[h.discrete] = 0.0
    if mean_grad_s is not None:
        dmu = dmu + np.asarray(mean_grad_s, dtype=np.float64)
    return mu, var, dmu, dvar


def expected_improvement_lin_grad(fit_coefs, mu, var, dmu, dvar, best_yet: float):
    """EI(LinFitness) (expected_improvement.jl:93-101) and its gradient w.r.t. the candidate, by the
    chain rule through μf = cᵀμ, σf = sqrt(c²ᵀσ²):  ∂EI/∂μf = Φ(z), ∂EI/∂σf = φ(z), z = (μf − b)/σf.
    mu, var: P×M;  dmu, dvar: P×d×M.  Returns (ei[M], dei[d,M]); where σf = 0 the σ-term is dropped."""
    c = np.asarray(fit_coefs, dtype=np.float64)
    mu, var = np.atleast_2d(mu), np.atleast_2d(var)
    muf = c @ mu
    vf = (c * c) @ var
    sf = np.sqrt(vf)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (muf - best_yet) / sf
    ei = expected_improvement_lin(fit_coefs, mu, var, best_yet)
    dmuf = np.einsum("p,pdm->dm", c, dmu)
    dvf = np.einsum("p,pdm->dm", c * c, dvar)
    with np.errstate(divide="ignore", invalid="ignore"):
        dsf = np.where(sf > 0, dvf / (2.0 * sf), 0.0)
    dei = normcdf(z) * dmuf + normpdf(z) * dsf
    return ei, dei


def model_mean_and_var(posts: Sequence[GPPosterior], Xs, means_s=None):
    """DefaultModelPosterior fan-out over outputs (src/posterior.jl:67-72): rows = outputs -> P×M."""
    mus, vars_ = [], []
    for i, p in enumerate(posts):
        m, v = gp_mean_and_var(p, Xs, None if means_s is None else means_s[i])
        mus.append(m)
        vars_.append(v)
    return np.vstack(mus), np.vstack(vars_)


def average_mean(sample_posts: Sequence[Sequence[GPPosterior]], Xs):
    """average_mean (src/posterior.jl:177-179): arithmetic mean over BI samples of mean(post, X)."""
    acc = None
    for posts in sample_posts:
        mu, _ = model_mean_and_var(posts, Xs)
        acc = mu if acc is None else acc + mu
    return acc / len(sample_posts)


# --------------------------------------------------------------------------------------
# Expected improvement  (src/acquisitions/expected_improvement.jl)
# --------------------------------------------------------------------------------------
def normcdf(z):
    """Distributions.cdf(Normal(), z) = StatsFuns.normcdf(z) = erfc(-z/√2)/2."""
    return 0.5 * erfc(-np.asarray(z, dtype=np.float64) / math.sqrt(2.0))


def normpdf(z):
    """Distributions.pdf(Normal(), z) = exp(-z²/2)/√(2π)."""
    z = np.asarray(z, dtype=np.float64)
    return np.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)


def is_feasible(y, y_max) -> bool:
    """src/utils/utils.jl:33  all(y .<= y_max)."""
    return bool(np.all(np.asarray(y) <= np.asarray(y_max)))


def best_so_far(fit_coefs, Y, y_max):
    """best_so_far (expected_improvement.jl:134-140), LinFitness (src/types/fitness.jl):
    max fitness over FEASIBLE raw data columns; None ("nothing") if Y empty or none feasible."""
    Y = np.asarray(Y, dtype=np.float64)
    if Y.size == 0:
        return None
    Y = np.atleast_2d(Y)
    c = np.asarray(fit_coefs, dtype=np.float64)
    best = None
    for j in range(Y.shape[1]):
        if is_feasible(Y[:, j], y_max):
            f = float(c @ Y[:, j])
            best = f if best is None or f > best else best
    return best


def expected_improvement_lin(fit_coefs, mu, var, best_yet: float):
    """expected_improvement(::LinFitness) (expected_improvement.jl:93-101), vectorised over
    candidates: mu, var are P×M.  IEEE semantics when σf == 0 (diff/0 = ±Inf) are preserved."""
    c = np.asarray(fit_coefs, dtype=np.float64)
    mu = np.atleast_2d(np.asarray(mu, dtype=np.float64).T).T
    var = np.atleast_2d(np.asarray(var, dtype=np.float64).T).T
    muf = c @ mu
    sf = np.sqrt((c * c) @ var)
    diff = muf - best_yet
    with np.errstate(divide="ignore", invalid="ignore"):
        z = diff / sf
        ei = diff * normcdf(z) + sf * normpdf(z)
    return np.where((diff == 0.0) & (sf == 0.0), 0.0, ei)


def _normcdf_musigma(mu, sigma, x):
    """StatsFuns.normcdf(μ, σ, x) as used by cdf(Normal(μ, σ), x): z = (x-μ)/σ, except that
    for σ == 0 and x == μ the z-value is +Inf (cdf = 1).  Restated from the public StatsFuns
    source (not in /root/reference)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (x - mu) / sigma
    z = np.where((sigma == 0.0) & (x == mu), np.inf, z)
    return normcdf(z)


def feas_prob(mu, var, y_max):
    """feas_prob (expected_improvement.jl:113-114): Π_i cdf(Normal(μ_i, √σ²_i), y_max_i);
    constraints === nothing -> 1.  Entries y_max_i = Inf are the `Infinity` singleton whose cdf
    is hard-wired to 1 (src/utils/inf.jl:5-15, src/types/problem.jl:71-73)."""
    mu = np.atleast_2d(np.asarray(mu, dtype=np.float64).T).T
    var = np.atleast_2d(np.asarray(var, dtype=np.float64).T).T
    if y_max is None:
        return np.ones(mu.shape[1])
    y_max = np.asarray(y_max, dtype=np.float64)
    fp = np.ones(mu.shape[1])
    for i in range(mu.shape[0]):
        if np.isinf(y_max[i]) and y_max[i] > 0:
            continue
        fp = fp * _normcdf_musigma(mu[i], np.sqrt(var[i]), y_max[i])
    return fp


def in_bounds(Xs, lb, ub):
    """in_bounds (src/types/domain.jl:73-78), vectorised over columns."""
    Xs = np.asarray(Xs, dtype=np.float64)
    lb = np.asarray(lb, dtype=np.float64)[:, None]
    ub = np.asarray(ub, dtype=np.float64)[:, None]
    return ~(np.any(Xs < lb, axis=0) | np.any(Xs > ub, axis=0))


def ei_acquisition(posts_or_samples, Xs, fit_coefs, y_max, best_yet, valid_mask=None,
                   means_s=None, constrained: Optional[bool] = None):
    """construct_ei, all four single-posterior variants + the BI average
    (expected_improvement.jl:68-90) wrapped by make_safe(acq, domain) (:58-65).

    posts_or_samples: list of P GPPosterior (one per output), or list over S samples of such lists.
    y_max: length-P vector (Inf allowed) or None for `constraints === nothing`.
    best_yet: float or None ("nothing": no feasible observation yet).
    valid_mask: bool[M], False -> acq = 0.0 (outside bounds / violated cons, evaluated by caller).
    """
    if len(posts_or_samples) and isinstance(posts_or_samples[0], GPPosterior):
        samples = [posts_or_samples]
    else:
        samples = list(posts_or_samples)
    Xs = np.asarray(Xs, dtype=np.float64)
    M = Xs.shape[1]
    if constrained is None:
        constrained = y_max is not None
    acc = np.zeros(M)
    for posts in samples:
        if (not constrained) and best_yet is None:
            acq = np.zeros(M)                                           # :68-70
        else:
            mu, var = model_mean_and_var(posts, Xs, means_s)
            if best_yet is None:
                acq = feas_prob(mu, var, y_max)                         # :71-73
            elif not constrained:
                acq = expected_improvement_lin(fit_coefs, mu, var, best_yet)   # :74-76
            else:
                acq = expected_improvement_lin(fit_coefs, mu, var, best_yet) * feas_prob(mu, var, y_max)  # :77-84
        acc += acq
    acq = acc / len(samples)                                            # :87-90
    if valid_mask is not None:
        acq = np.where(np.asarray(valid_mask, dtype=bool), acq, 0.0)    # :58-65
    return acq


def feas_prob_grad(mu, var, dmu, dvar, y_max):
    """feas_prob (expected_improvement.jl:113-114) and its gradient w.r.t. the candidate:
    FP = Π_p Φ(t_p), t_p = (ymax_p − μ_p)/sqrt(σ²_p); outputs with ymax = +Inf are skipped (factor 1).
    mu, var: P×M; dmu, dvar: P×d×M.  Returns (fp[M], dfp[d,M])."""
    mu, var = np.atleast_2d(mu), np.atleast_2d(var)
    P, M = mu.shape
    d = dmu.shape[1]
    y_max = np.asarray(y_max, dtype=np.float64)
    Phi = np.ones((P, M))
    fac = np.zeros((P, d, M))
    for p in range(P):
        if np.isposinf(y_max[p]):
            continue
        v = np.maximum(var[p], 0.0)
        s = np.sqrt(v)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where((s == 0) & (y_max[p] == mu[p]), np.inf, (y_max[p] - mu[p]) / s)
            dt = np.where(v > 0, -dmu[p] / s - (y_max[p] - mu[p]) * dvar[p] / (2.0 * s * v), 0.0)
        Phi[p] = normcdf(t)
        fac[p] = np.where(v > 0, normpdf(t), 0.0) * dt
    fp = np.prod(Phi, axis=0)
    dfp = np.zeros((d, M))
    for p in range(P):
        others = np.prod(np.delete(Phi, p, axis=0), axis=0) if P > 1 else np.ones(M)
        dfp += others * fac[p]
    return fp, dfp


def ei_acquisition_grad(posts: Sequence[GPPosterior], Xs, fit_coefs, y_max, best_yet, valid_mask=None,
                        means_s=None, mean_grads_s=None):
    """construct_ei for ONE posterior sample (expected_improvement.jl:68-84) wrapped by make_safe (:58-65),
    with its gradient w.r.t. the candidates (what ForwardDiff yields inside OptimizationAM).
    Returns (acq[M], dacq[d,M])."""
    Xs = np.asarray(Xs, dtype=np.float64)
    d, M = Xs.shape
    P = len(posts)
    mu, var = np.zeros((P, M)), np.zeros((P, M))
    dmu, dvar = np.zeros((P, d, M)), np.zeros((P, d, M))
    for p, post in enumerate(posts):
        ms = None if means_s is None else means_s[p]
        mg = None if mean_grads_s is None else mean_grads_s[p]
        mu[p], v, dmu[p], dvar[p] = gp_mean_and_var_grad(post, Xs, ms, mg)
        var[p] = clip_var(v)
        dvar[p] = np.where(var[p] > 0, dvar[p], 0.0)
    constrained = y_max is not None
    if (not constrained) and best_yet is None:
        acq, dacq = np.zeros(M), np.zeros((d, M))
    elif best_yet is None:
        acq, dacq = feas_prob_grad(mu, var, dmu, dvar, y_max)
    elif not constrained:
        acq, dacq = expected_improvement_lin_grad(fit_coefs, mu, var, dmu, dvar, best_yet)
    else:
        ei, dei = expected_improvement_lin_grad(fit_coefs, mu, var, dmu, dvar, best_yet)
        fp, dfp = feas_prob_grad(mu, var, dmu, dvar, y_max)
        acq, dacq = ei * fp, dei * fp + ei * dfp
    if valid_mask is not None:
        vm = np.asarray(valid_mask, dtype=bool)
        acq = np.where(vm, acq, 0.0)
        dacq = np.where(vm[None, :], dacq, 0.0)
    return acq, dacq


def argmax_first(vals):
    """Julia `argmax` (sampling.jl:36-39): first index of the maximum."""
    return int(np.argmax(np.asarray(vals)))


# ------------------------------------------------------------------------------------------
# GradientGaussianProcess: a GP conditioned on function values AND gradients,
# src/models/gradient_gp.jl.  The reference obtains the derivative blocks by ForwardDiff through
# KernelFunctions (:143-159); ForwardDiff is exact to rounding, so the analytic derivatives of the
# radial profiles below ARE what it computes.  The reference holds no tests or fixtures for this
# model: the restatement is pinned by finite differences of the plain kernel only
# (tests/test_oracle_crosscheck.py) — PARITY UNPINNED beyond that.
# ------------------------------------------------------------------------------------------
def kappa_second(kernel: int, r: np.ndarray) -> np.ndarray:
    """g(r) = h'(r)/r with h = κ'/r:  SqExp e^{-r²/2};  Matern52 (25/3) e^{-√5 r};
    Matern32 3√3 e^{-√3 r}/r (singular at r = 0 — the Matérn-3/2 cusp the reference steps around
    by perturbing one argument, gradient_gp.jl:151-152)."""
    r = np.asarray(r, dtype=np.float64)
    if kernel == SQEXP:
        return np.exp(-0.5 * r * r)
    if kernel == MATERN52:
        return (25.0 / 3.0) * np.exp(-_SQRT5 * r)
    if kernel == MATERN32:
        return 3.0 * _SQRT3 * np.exp(-_SQRT3 * r) / r
    raise ValueError(f"unknown kernel id {kernel}")


_ISAPPROX_RTOL = math.sqrt(np.finfo(np.float64).eps)     # Julia isapprox default for Float64 vectors


def _isapprox(xi: np.ndarray, xj: np.ndarray) -> bool:
    """Julia `xi ≈ xj` for vectors: norm(xi - xj) <= rtol * max(norm(xi), norm(xj)), atol = 0."""
    return float(np.linalg.norm(xi - xj)) <= _ISAPPROX_RTOL * max(float(np.linalg.norm(xi)), float(np.linalg.norm(xj)))


def _kernel_and_derivs(kernel: int, lam: np.ndarray, amp2: float, xi: np.ndarray, xj: np.ndarray):
    """gradient_gp.jl:143-159: (k, ∂k/∂xi, ∂k/∂xj, ∂²k/∂xi∂xjᵀ) of k = α² κ(‖(xi−xj)⊘λ‖); the value at the
    given points, the derivatives at (xi, xj + 1e-8) when xi ≈ xj."""
    u = xi - xj
    k_val = amp2 * float(kappa(kernel, np.sqrt(np.sum((u / lam) ** 2))))
    if _isapprox(xi, xj):
        u = xi - (xj + MIN_PARAM_VALUE)
    r = float(np.sqrt(np.sum((u / lam) ** 2)))
    h = float(kappa_prime_over_r(kernel, np.float64(r)))
    g = float(kappa_second(kernel, np.float64(r)))
    s = u / lam ** 2
    dxi = amp2 * h * s
    d2 = -amp2 * (h * np.diag(1.0 / lam ** 2) + g * np.outer(s, s))
    return k_val, dxi, -dxi, d2


def augmented_kernel_matrix(kernel: int, X, lengthscale, amplitude: float, noise_std: float, grad_noise_std: float):
    """gradient_gp.jl:175-210 `_build_augmented_kernel`: the n(1+d) square matrix over the observation
    ordering [f(x_1..n), ∂f/∂x_1(x_1..n), …, ∂f/∂x_d(x_1..n)], noise σ² on the function block's diagonal
    and σ_∂² on the gradient blocks' (+1e-8 on every parameter, :128-131,:200-204)."""
    kernel = KERNEL_NAMES.get(kernel, kernel) if isinstance(kernel, str) else kernel
    X = np.asarray(X, dtype=np.float64)
    d, n = X.shape
    lam = np.asarray(lengthscale, dtype=np.float64) + MIN_PARAM_VALUE
    amp2 = (float(amplitude) + MIN_PARAM_VALUE) ** 2
    N = n * (1 + d)
    K = np.zeros((N, N))
    for i in range(n):
        for j in range(n):
            k_val, dxi, dxj, d2 = _kernel_and_derivs(kernel, lam, amp2, X[:, i], X[:, j])
            K[i, j] = k_val
            for l in range(d):
                K[i, n + l * n + j] = dxj[l]
                K[n + l * n + i, j] = dxi[l]
                for m in range(d):
                    K[n + l * n + i, n + m * n + j] = d2[l, m]
    K[np.arange(n), np.arange(n)] += (float(noise_std) + MIN_PARAM_VALUE) ** 2
    K[np.arange(n, N), np.arange(n, N)] += (float(grad_noise_std) + MIN_PARAM_VALUE) ** 2
    # `Symmetric(K)` (:209) reads the UPPER triangle; the perturbed entries of coincident pairs are not symmetric
    return np.triu(K) + np.triu(K, 1).T


def augmented_cross_cov(kernel: int, X, lengthscale, amplitude: float, Xs):
    """gradient_gp.jl:213-243 `_build_cross_cov` for every column of Xs: N_aug × M matrix
    [k(x*, x_j); ∂k(x*, x_j)/∂(x_j)_l], the derivative taken at x_j + 1e-8 when x* ≈ x_j."""
    kernel = KERNEL_NAMES.get(kernel, kernel) if isinstance(kernel, str) else kernel
    X = np.asarray(X, dtype=np.float64)
    Xs = np.asarray(Xs, dtype=np.float64)
    if Xs.ndim == 1:
        Xs = Xs[:, None]
    d, n = X.shape
    lam = np.asarray(lengthscale, dtype=np.float64) + MIN_PARAM_VALUE
    amp2 = (float(amplitude) + MIN_PARAM_VALUE) ** 2
    out = np.zeros((n * (1 + d), Xs.shape[1]))
    for c in range(Xs.shape[1]):
        for j in range(n):
            k_val, _, dxj, _ = _kernel_and_derivs(kernel, lam, amp2, Xs[:, c], X[:, j])
            out[j, c] = k_val
            out[n + np.arange(d) * n + j, c] = dxj
    return out


def augmented_obs_vector(y, dY):
    """gradient_gp.jl:288-302 `_build_obs_vector`: [y_1..n, ∂y/∂x_1 (1..n), …, ∂y/∂x_d (1..n)];  dY is d×n."""
    return np.concatenate([np.asarray(y, dtype=np.float64)] + [np.asarray(dY, dtype=np.float64)[l, :] for l in range(np.shape(dY)[0])])


@dataclass
class GradientGPPosterior:
    kernel: int
    X: np.ndarray
    lengthscale: np.ndarray
    amplitude: float
    L: np.ndarray           # lower Cholesky factor of the augmented matrix
    alpha: np.ndarray       # K_aug⁻¹ ỹ
    logpdf: float


def gradient_gp_fit(X, y, dY, kernel, lengthscale, amplitude, noise_std, grad_noise_std) -> GradientGPPosterior:
    """gradient_gp.jl:307-329 `model_posterior_slice` and :367-397 `data_loglike` (the model's mean is
    not used by either)."""
    kernel = KERNEL_NAMES.get(kernel, kernel) if isinstance(kernel, str) else kernel
    yt = augmented_obs_vector(y, dY)
    K = augmented_kernel_matrix(kernel, X, lengthscale, amplitude, noise_std, grad_noise_std)
    try:
        L = sla.cholesky(K, lower=True, check_finite=False)
    except np.linalg.LinAlgError as e:
        raise PosDefException(str(e))
    alpha = sla.cho_solve((L, True), yt, check_finite=False)
    ll = -0.5 * (float(yt @ alpha) + 2.0 * float(np.sum(np.log(np.diag(L)))) + len(yt) * math.log(2.0 * math.pi))
    return GradientGPPosterior(kernel, np.asarray(X, dtype=np.float64), np.asarray(lengthscale, dtype=np.float64),
                               float(amplitude), L, alpha, ll)


def gradient_gp_mean_and_var(post: GradientGPPosterior, Xs):
    """gradient_gp.jl:334-361: μ = k*·α,  σ² = max(0, k(x*,x*) − ‖L⁻¹k*‖²)  (no jitter, no mean function)."""
    Ks = augmented_cross_cov(post.kernel, post.X, post.lengthscale, post.amplitude, Xs)
    mu = Ks.T @ post.alpha
    V = sla.solve_triangular(post.L, Ks, lower=True, check_finite=False)
    amp2 = (post.amplitude + MIN_PARAM_VALUE) ** 2
    return mu, np.maximum(0.0, amp2 - np.sum(V * V, axis=0))


def gradient_gp_mean_and_var_grad(post: "GradientGPPosterior", Xs):
    """What ForwardDiff yields when OptimizationAM differentiates the acquisition through a GradientGaussianProcess posterior
    (src/acquisition_maximizers/optimization.jl:36,89-118 through src/models/gradient_gp.jl:334-361): the moments of
    `gradient_gp_mean_and_var` (variance UNclipped here) and their gradients w.r.t. the candidate columns,
        ∇μ = Σ_rows α_row ∇k*_row ,   ∇σ² = −2 Σ_rows w_row ∇k*_row ,  w = K⁻¹k* ,
    with, for t = (x* − x_j) ⊘ λ², h = κ'(r)/r, g = h'(r)/r:
        value row j:        ∇_{x*} k(x*, x_j)              = α² h t
        derivative row (j, l):  ∇_{x*} ∂k(x*, x_j)/∂(x_j)_l = −α² (g t_l t + h e_l/λ_l²)
    (the derivative rows at x_j + 1e-8 when x* ≈ x_j, as `_build_cross_cov` evaluates them).  No reference test covers this
    (PARITY UNPINNED beyond the finite differences of tests/test_oracle_crosscheck.py).
    Returns (mu[M], var[M], dmu[d, M], dvar[d, M])."""
    kernel = KERNEL_NAMES.get(post.kernel, post.kernel) if isinstance(post.kernel, str) else post.kernel
    X = np.asarray(post.X, dtype=np.float64)
    Xs = np.asarray(Xs, dtype=np.float64)
    if Xs.ndim == 1:
        Xs = Xs[:, None]
    d, n = X.shape
    M = Xs.shape[1]
    lam = np.asarray(post.lengthscale, dtype=np.float64) + MIN_PARAM_VALUE
    amp2 = (float(post.amplitude) + MIN_PARAM_VALUE) ** 2
    Ks = augmented_cross_cov(post.kernel, X, post.lengthscale, post.amplitude, Xs)
    mu = Ks.T @ post.alpha
    V = sla.solve_triangular(post.L, Ks, lower=True, check_finite=False)
    W = sla.solve_triangular(post.L, V, lower=True, trans="T", check_finite=False)
    var = amp2 - np.sum(V * V, axis=0)
    dmu, dvar = np.zeros((d, M)), np.zeros((d, M))
    for c in range(M):
        for j in range(n):
            xs, xj = Xs[:, c], X[:, j]
            u0 = xs - xj
            r0 = float(np.sqrt(np.sum((u0 / lam) ** 2)))
            gk = amp2 * float(kappa_prime_over_r(kernel, np.float64(r0))) * (u0 / lam ** 2)          # value row (as given)
            u = xs - (xj + MIN_PARAM_VALUE) if _isapprox(xs, xj) else u0
            r = float(np.sqrt(np.sum((u / lam) ** 2)))
            h = float(kappa_prime_over_r(kernel, np.float64(r)))
            g = float(kappa_second(kernel, np.float64(r)))
            t = u / lam ** 2
            rows = n + np.arange(d) * n + j
            J = -amp2 * (g * np.outer(t, t) + h * np.diag(1.0 / lam ** 2))                            # J[l, m] = ∂(row l)/∂x*_m
            dmu[:, c] += post.alpha[j] * gk + J.T @ post.alpha[rows]
            dvar[:, c] += -2.0 * (W[j, c] * gk + J.T @ W[rows, c])
    return mu, var, dmu, dvar


def kappa_third(kernel: int, r: np.ndarray) -> np.ndarray:
    """q(r) = g'(r)/r with g = kappa_second:  SqExp −e^{-r²/2};  Matern52 −(25√5/3) e^{-√5 r}/r;
    Matern32 −3√3 (1+√3 r) e^{-√3 r}/r³.  (Only ever evaluated at r > 0: coincident points are perturbed like in
    `_kernel_and_derivs`.)"""
    r = np.asarray(r, dtype=np.float64)
    if kernel == SQEXP:
        return -np.exp(-0.5 * r * r)
    if kernel == MATERN52:
        return -(25.0 * _SQRT5 / 3.0) * np.exp(-_SQRT5 * r) / r
    if kernel == MATERN32:
        return -3.0 * _SQRT3 * (1.0 + _SQRT3 * r) * np.exp(-_SQRT3 * r) / r ** 3
    raise ValueError(f"unknown kernel id {kernel}")


def _kernel_derivs_dlam(kernel: int, lam: np.ndarray, amp2: float, xi: np.ndarray, xj: np.ndarray):
    """∂/∂λ_m of the four blocks `_kernel_and_derivs` returns, m = 0..d-1 in the leading axis:
    (dk[m], d(∂k/∂xi)[m, l], d(∂k/∂xj)[m, l], d(∂²k)[m, l, l']).  With u = xi − xj (perturbed for the derivative blocks when
    xi ≈ xj), s = u ⊘ λ², w_m = −u_m²/λ_m³ (= r ∂r/∂λ_m) and the profiles h = κ'/r, g = h'/r, q = g'/r:
        ∂k/∂λ_m            = α² h(r₀) w⁰_m                                   (value block: unperturbed points)
        ∂(α² h s_l)/∂λ_m   = α² (g w_m s_l − 2 h s_l δ_lm/λ_l)
        ∂(−α²(g s_l s_l' + h δ_ll'/λ_l²))/∂λ_m
                           = −α² (q w_m s_l s_l' − 2 g s_l s_l' (δ_lm/λ_l + δ_l'm/λ_l') + g w_m δ_ll'/λ_l² − 2 h δ_ll' δ_lm/λ_l³)."""
    d = len(lam)
    u0 = xi - xj
    r0 = float(np.sqrt(np.sum((u0 / lam) ** 2)))
    dk = amp2 * float(kappa_prime_over_r(kernel, np.float64(r0))) * (-(u0 ** 2) / lam ** 3)
    u = xi - (xj + MIN_PARAM_VALUE) if _isapprox(xi, xj) else u0
    r = float(np.sqrt(np.sum((u / lam) ** 2)))
    h = float(kappa_prime_over_r(kernel, np.float64(r)))
    g = float(kappa_second(kernel, np.float64(r)))
    q = float(kappa_third(kernel, np.float64(r)))
    s = u / lam ** 2
    w = -(u ** 2) / lam ** 3
    ddxi = np.zeros((d, d))
    dd2 = np.zeros((d, d, d))
    for m in range(d):
        ddxi[m] = amp2 * g * w[m] * s
        ddxi[m, m] += -2.0 * amp2 * h * s[m] / lam[m]
        blk = q * w[m] * np.outer(s, s) + g * w[m] * np.diag(1.0 / lam ** 2)
        blk[m, :] += -2.0 * g * s[m] * s / lam[m]
        blk[:, m] += -2.0 * g * s * s[m] / lam[m]
        blk[m, m] += -2.0 * h / lam[m] ** 3
        dd2[m] = -amp2 * blk
    return dk, ddxi, -ddxi, dd2


def gradient_gp_loglike_grad(X, y, dY, kernel, lengthscale, amplitude, noise_std, grad_noise_std):
    """What ForwardDiff yields when OptimizationMAP differentiates `data_loglike` of a GradientGaussianProcess
    (src/model_fitters/optimization.jl:146-164 through src/models/gradient_gp.jl:367-397): the log-likelihood of the augmented
    observation vector and its gradient w.r.t. (λ_1..λ_d, α, σ, σ_∂),
        ∂ℓ/∂θ = ½ Σ_ab G_ab ∂K_ab/∂θ ,   G = a aᵀ − K⁻¹ ,  a = K⁻¹ỹ ,
    over the matrix `cholesky(Symmetric(K))` factorises (upper triangle of `_build_augmented_kernel`, mirrored).  No reference test
    covers this (PARITY UNPINNED beyond the finite differences of tests/test_oracle_crosscheck.py).
    Returns (logpdf, grad[d + 3])."""
    kernel = KERNEL_NAMES.get(kernel, kernel) if isinstance(kernel, str) else kernel
    X = np.asarray(X, dtype=np.float64)
    d, n = X.shape
    lam = np.asarray(lengthscale, dtype=np.float64) + MIN_PARAM_VALUE
    amp = float(amplitude) + MIN_PARAM_VALUE
    amp2 = amp * amp
    N = n * (1 + d)
    post = gradient_gp_fit(X, y, dY, kernel, lengthscale, amplitude, noise_std, grad_noise_std)
    Kinv = sla.cho_solve((post.L, True), np.eye(N), check_finite=False)
    G = np.outer(post.alpha, post.alpha) - Kinv
    dK = np.zeros((d + 1, N, N))                             # λ_1..λ_d, then α (noise-free part · 2/α)
    for i in range(n):
        for j in range(n):
            k_val, dxi, dxj, d2 = _kernel_and_derivs(kernel, lam, amp2, X[:, i], X[:, j])
            dk, ddxi, ddxj, dd2 = _kernel_derivs_dlam(kernel, lam, amp2, X[:, i], X[:, j])
            dK[:d, i, j] = dk
            dK[d, i, j] = k_val
            for l in range(d):
                dK[:d, i, n + l * n + j] = ddxj[:, l]
                dK[d, i, n + l * n + j] = dxj[l]
                dK[:d, n + l * n + i, j] = ddxi[:, l]
                dK[d, n + l * n + i, j] = dxi[l]
                for m in range(d):
                    dK[:d, n + l * n + i, n + m * n + j] = dd2[:, l, m]
                    dK[d, n + l * n + i, n + m * n + j] = d2[l, m]
    dK[d] *= 2.0 / amp
    grad = np.zeros(d + 3)
    for t in range(d + 1):
        Ks = np.triu(dK[t]) + np.triu(dK[t], 1).T            # `Symmetric(K)` reads the upper triangle
        grad[t] = 0.5 * float(np.sum(G * Ks))
    dg = np.diag(G)
    grad[d + 1] = (float(noise_std) + MIN_PARAM_VALUE) * float(np.sum(dg[:n]))
    grad[d + 2] = (float(grad_noise_std) + MIN_PARAM_VALUE) * float(np.sum(dg[n:]))
    return post.logpdf, grad


# ------------------------------------------------------------------------------------------
# NonstationaryGP: src/models/nonstationary_gp/nonstationary_gp.jl.  λ(·), α(·), σ(·) are
# functions of the input (posteriors of latent ParametrizedGPs, or constants, :198-212); here they arrive
# evaluated: lam_X d×N, amp_X N, noise_X N at the training points, lam_Xs d×M, amp_Xs M at the candidates.
# The reference's tests hold nothing for this model.  Pin: with constant λ, α, σ the Gibbs kernel IS the
# squared-exponential ARD kernel, so the restatement must reproduce the (pinned) plain-GP oracle there
# (tests/test_oracle_crosscheck.py); with varying parameters PARITY IS UNPINNED.
# ------------------------------------------------------------------------------------------
def gibbs_kernel_matrix(Xa, lam_a, amp_a, Xb, lam_b, amp_b):
    """nonstat_kernel / gibbs_kernel (:66-107):
    k(x, y) = ((α(x)+α(y))/2)² Π_i sqrt(2 λ_i(x) λ_i(y) / (λ_i(x)² + λ_i(y)²)) exp(−(x_i − y_i)² / (λ_i(x)² + λ_i(y)²))."""
    Xa, Xb = np.asarray(Xa, np.float64), np.asarray(Xb, np.float64)
    la, lb = np.asarray(lam_a, np.float64), np.asarray(lam_b, np.float64)
    K = (0.5 * (np.asarray(amp_a, np.float64)[:, None] + np.asarray(amp_b, np.float64)[None, :])) ** 2
    for i in range(Xa.shape[0]):                                       # the per-dimension product of :93-99
        s = la[i][:, None] ** 2 + lb[i][None, :] ** 2
        K = K * (np.sqrt(2.0 * la[i][:, None] * lb[i][None, :] / s) * np.exp(-((Xa[i][:, None] - Xb[i][None, :]) ** 2) / s))
    return K


@dataclass
class NonstationaryPosterior:
    X: np.ndarray
    lam_X: np.ndarray
    amp_X: np.ndarray
    discrete: Optional[np.ndarray]
    L: np.ndarray
    a: np.ndarray
    logpdf: float


def nonstationary_fit(X, y, lam_X, amp_X, noise_X, mean=None, discrete=None) -> NonstationaryPosterior:
    """finite_nongp (:183-196) + AbstractGPs.posterior / logpdf (:153-157, :237-245): K = Gibbs Gram matrix of the
    (rounded where discrete) points + diag(σ(x_j)²)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    N = X.shape[1]
    Xr = discrete_round(X, discrete)
    K = gibbs_kernel_matrix(Xr, lam_X, amp_X, Xr, lam_X, amp_X)
    K[np.diag_indices(N)] += np.asarray(noise_X, np.float64) ** 2
    try:
        L = sla.cholesky(K, lower=True, check_finite=False)
    except np.linalg.LinAlgError as e:
        raise PosDefException(str(e))
    delta = y - _mean_vec(mean, X)
    z = sla.solve_triangular(L, delta, lower=True, check_finite=False)
    a = sla.solve_triangular(L, z, lower=True, trans="T", check_finite=False)
    logpdf = -0.5 * (N * _LOG2PI + 2.0 * float(np.sum(np.log(np.diag(L)))) + float(z @ z))
    return NonstationaryPosterior(X, np.asarray(lam_X, np.float64), np.asarray(amp_X, np.float64),
                                  None if discrete is None else np.asarray(discrete, bool), L, a, logpdf)


def nonstationary_loglike_grad(X, y, lam_X, amp_X, noise_X, mean=None, discrete=None):
    """What ForwardDiff carries back to the latent models when a fitter differentiates `data_loglike_slice` of a NonstationaryGP
    (nonstationary_gp.jl:237-245 through finite_nongp :183-196): the log-likelihood and its partial derivatives w.r.t. the latent
    models' VALUES at the training points — λ(x_i) (d×N), α(x_i), σ(x_i), m(x_i); the caller chains them through its latent models.
    With G = a aᵀ − K⁻¹, a = K⁻¹(y − m), K⁰ the noise-free Gibbs Gram matrix, q_l = λ_il² + λ_jl², Δ_l = x_il − x_jl:
        ∂ℓ/∂λ_il = Σ_j G_ij K⁰_ij [½ (1/λ_il − 2 λ_il/q_l) + 2 λ_il Δ_l²/q_l²] ,   ∂ℓ/∂α_i = Σ_j G_ij K⁰_ij 2/(α_i + α_j) ,
        ∂ℓ/∂σ_i = σ_i G_ii ,   ∂ℓ/∂m_i = a_i
    (the j = i terms: the bracket vanishes, and K⁰_ii 2/(2α_i) = α_i is ½ ∂K_ii/∂α_i).  No reference test covers this (PARITY
    UNPINNED beyond the finite differences of tests/test_oracle_crosscheck.py).  Returns (logpdf, dlam[d, N], damp[N], dnoise[N], dmean[N])."""
    X = np.asarray(X, dtype=np.float64)
    lam = np.asarray(lam_X, dtype=np.float64)
    amp = np.asarray(amp_X, dtype=np.float64).reshape(-1)
    noi = np.asarray(noise_X, dtype=np.float64).reshape(-1)
    d, N = X.shape
    post = nonstationary_fit(X, y, lam, amp, noi, mean=mean, discrete=discrete)
    Xr = discrete_round(X, discrete)
    K0 = gibbs_kernel_matrix(Xr, lam, amp, Xr, lam, amp)
    Kinv = sla.cho_solve((post.L, True), np.eye(N), check_finite=False)
    W = (np.outer(post.a, post.a) - Kinv) * K0
    dlam = np.zeros((d, N))
    for l in range(d):
        q = lam[l][:, None] ** 2 + lam[l][None, :] ** 2
        dl2 = (Xr[l][:, None] - Xr[l][None, :]) ** 2
        D = 0.5 * (1.0 / lam[l][:, None] - 2.0 * lam[l][:, None] / q) + 2.0 * lam[l][:, None] * dl2 / q ** 2
        dlam[l] = np.sum(W * D, axis=1)
    damp = np.sum(W * (2.0 / (amp[:, None] + amp[None, :])), axis=1)
    dnoise = noi * (post.a ** 2 - np.diag(Kinv))
    return post.logpdf, dlam, damp, dnoise, post.a.copy()


def nonstationary_mean_and_var(post: NonstationaryPosterior, Xs, lam_Xs, amp_Xs, mean_s=None, clip: bool = True):
    """mean_and_var of the GaussianProcessPosterior built over the NonstationaryKernel (gaussian_process.jl:174-178):
    k(x*,x*) = α(x*)², + 1e-18, then _clip_var."""
    Xs = np.asarray(Xs, dtype=np.float64)
    Ks = gibbs_kernel_matrix(discrete_round(post.X, post.discrete), post.lam_X, post.amp_X,
                             discrete_round(Xs, post.discrete), lam_Xs, amp_Xs)
    mu = _mean_vec(mean_s, Xs) + Ks.T @ post.a
    V = sla.solve_triangular(post.L, Ks, lower=True, check_finite=False)
    var = np.asarray(amp_Xs, np.float64) ** 2 - np.sum(V * V, axis=0) + PREDICT_JITTER
    return mu, (clip_var(var) if clip else var)


def nonstationary_mean_and_var_grad(post: NonstationaryPosterior, Xs, lam_Xs, amp_Xs, dlam_Xs=None, damp_Xs=None, mean_s=None,
                                    mean_grad_s=None):
    """What ForwardDiff yields when OptimizationAM (src/acquisition_maximizers/optimization.jl:36,89-118) differentiates through a
    NonstationaryGP posterior (src/models/nonstationary_gp/nonstationary_gp.jl:153-196): the candidate enters the Gibbs kernel
    directly AND through the latent λ(x*), α(x*); their Jacobians arrive evaluated, as the values do —
        dlam_Xs[l, m, j] = ∂λ_l/∂x_m at candidate j (None: constant λ),  damp_Xs[m, j] = ∂α/∂x_m (None: constant α).
    With q_l = λ_il² + λ*_l², Δ_l = x_il − x*_l:
        ∂ln k_i/∂x*_m (explicit) = 2 Δ_m / q_m ,   ∂ln k_i/∂λ*_l = ½ (1/λ*_l − 2 λ*_l/q_l) + 2 λ*_l Δ_l²/q_l² ,   ∂ln k_i/∂α* = 2/(α_i + α*)
        ∇μ = ∇m + Σ_i a_i ∇k_i ,   ∇σ² = 2 α* ∇α* − 2 Σ_i w_i ∇k_i ,  w = K⁻¹k*.
    The reference holds no test for this model (PARITY UNPINNED; pinned here on finite differences, tests/test_oracle_crosscheck.py).
    Discrete dimensions: the caller passes zero Jacobian columns and gets a zero explicit part.
    Returns (mu[M], var[M] unclipped, dmu[d, M], dvar[d, M])."""
    Xs = np.asarray(Xs, dtype=np.float64)
    if Xs.ndim == 1:
        Xs = Xs[:, None]
    d, M = Xs.shape
    lam_Xs = np.asarray(lam_Xs, np.float64).reshape(d, M)
    amp_Xs = np.asarray(amp_Xs, np.float64).reshape(M)
    Xr, Xsr = discrete_round(post.X, post.discrete), discrete_round(Xs, post.discrete)
    Ks = gibbs_kernel_matrix(Xr, post.lam_X, post.amp_X, Xsr, lam_Xs, amp_Xs)
    mu = _mean_vec(mean_s, Xs) + Ks.T @ post.a
    V = sla.solve_triangular(post.L, Ks, lower=True, check_finite=False)
    W = sla.solve_triangular(post.L, V, lower=True, trans="T", check_finite=False)
    var = amp_Xs ** 2 - np.sum(V * V, axis=0) + PREDICT_JITTER
    Dl = np.zeros((d, d, M)) if dlam_Xs is None else np.asarray(dlam_Xs, np.float64).reshape(d, d, M)
    Da = np.zeros((d, M)) if damp_Xs is None else np.asarray(damp_Xs, np.float64).reshape(d, M)
    disc = np.zeros(d, bool) if post.discrete is None else np.asarray(post.discrete, bool)
    dmu = np.zeros((d, M)) if mean_grad_s is None else np.array(mean_grad_s, dtype=np.float64).reshape(d, M)
    dvar = np.zeros((d, M))
    for j in range(M):
        q = post.lam_X ** 2 + lam_Xs[:, j:j + 1] ** 2                      # d × N
        dlt = Xr - Xsr[:, j:j + 1]
        e = np.where(disc[:, None], 0.0, 2.0 * dlt / q)                    # explicit part
        c = 0.5 * (1.0 / lam_Xs[:, j:j + 1] - 2.0 * lam_Xs[:, j:j + 1] / q) + 2.0 * lam_Xs[:, j:j + 1] * dlt ** 2 / q ** 2
        sa = 2.0 / (post.amp_X + amp_Xs[j])                                # N
        G = e + Dl[:, :, j].T @ c + np.outer(Da[:, j], sa)                 # d × N: ∇_{x*} ln k_i
        gk = G * Ks[:, j][None, :]
        dmu[:, j] += gk @ post.a
        dvar[:, j] = 2.0 * amp_Xs[j] * Da[:, j] - 2.0 * (gk @ W[:, j])
    return mu, var, dmu, dvar

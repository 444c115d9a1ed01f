"""Independent cross-checks of the fp64 oracle: 50-digit mpmath and scikit-learn's GPR.

The reference holds no golden posterior numbers, so absolute values are pinned by
two independent evaluations of the published formulas instead.
"""
import numpy as np
import pytest

from oracle import gp_oracle as O
from oracle import mp_oracle as MP

CASES = [
    # (kernel, d, N, M, noise, with_mean)
    ("matern32", 1, 3, 4, 0.1, False),
    ("matern52", 2, 16, 5, 0.05, True),
    ("sqexp", 8, 24, 4, 0.02, False),
    ("matern52", 8, 32, 4, 1e-3, True),
]


def make_case(kernel, d, N, M, noise, with_mean, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (d, N))
    Xs = rng.uniform(0, 1, (d, M))
    lam = rng.uniform(0.3, 0.9, d)
    amp = 1.3
    y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + noise * rng.standard_normal(N)
    mX = 0.3 + X.sum(0) * 0.1 if with_mean else None
    ms = 0.3 + Xs.sum(0) * 0.1 if with_mean else None
    return X, y, Xs, lam, amp, noise, mX, ms


@pytest.mark.parametrize("case", CASES)
def test_oracle_vs_mpmath(case):
    kernel, d, N, M, noise, with_mean = case
    X, y, Xs, lam, amp, sig, mX, ms = make_case(*case)
    post = O.gp_fit(X, y, kernel, lam, amp, sig, mean=mX)
    mu, var = O.gp_mean_and_var(post, Xs, ms, clip=False)
    lp, mu_mp, var_mp = MP.posterior(X, y, O.KERNEL_NAMES[kernel], lam, amp, sig, Xs, mX, ms)
    # condition-aware fp64 bound: err <~ cond(K)·N·2^-53
    K = O.kernelmatrix(post.h, X) + (sig + 1e-8) ** 2 * np.eye(N)
    tol = max(1e-12, np.linalg.cond(K) * N * 2.0 ** -53 * 4)
    assert abs(post.logpdf - lp) <= tol * (1 + abs(lp))
    assert np.allclose(mu, mu_mp, rtol=0, atol=tol * (1 + np.abs(mu_mp).max()))
    assert np.allclose(var, var_mp, rtol=0, atol=tol * amp ** 2)


@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
def test_oracle_vs_sklearn(kernel):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import ConstantKernel, Matern, RBF
    X, y, Xs, lam, amp, sig, _, _ = make_case(kernel, 4, 40, 7, 0.05, False, seed=5)
    lam_e, amp_e, sig_e = lam + 1e-8, amp + 1e-8, sig + 1e-8
    base = {"matern32": Matern(length_scale=lam_e, nu=1.5), "matern52": Matern(length_scale=lam_e, nu=2.5),
            "sqexp": RBF(length_scale=lam_e)}[kernel]
    gpr = GaussianProcessRegressor(ConstantKernel(amp_e ** 2) * base, alpha=sig_e ** 2, optimizer=None)
    gpr.fit(X.T, y)
    mu_sk, std_sk = gpr.predict(Xs.T, return_std=True)
    post = O.gp_fit(X, y, kernel, lam, amp, sig)
    mu, var = O.gp_mean_and_var(post, Xs)
    assert np.allclose(mu, mu_sk, rtol=0, atol=1e-9)
    assert np.allclose(var, std_sk ** 2, rtol=0, atol=1e-9)
    assert abs(post.logpdf - gpr.log_marginal_likelihood_value_) <= 1e-8


def test_gemm_distance_form_is_close():
    """Distances.jl computes pairwise Euclidean via ||a||²+||b||²-2ab (SURVEY §7 hard parts); the
    oracle/HIP path uses direct differences.  Quantify the gap on the bench-shaped problem."""
    X, y, Xs, lam, amp, sig, _, _ = make_case("matern52", 8, 256, 16, 0.05, False, seed=9)
    pa = O.gp_fit(X, y, "matern52", lam, amp, sig, form="direct")
    pb = O.gp_fit(X, y, "matern52", lam, amp, sig, form="gemm")
    ma, va = O.gp_mean_and_var(pa, Xs, form="direct")
    mb, vb = O.gp_mean_and_var(pb, Xs, form="gemm")
    assert np.allclose(ma, mb, rtol=0, atol=1e-9) and np.allclose(va, vb, rtol=0, atol=1e-9)
    assert abs(pa.logpdf - pb.logpdf) <= 1e-9 * (1 + abs(pa.logpdf))


def test_non_pd_reports_posdef_and_neg_inf():
    X = np.array([[1., 1., 1.]])           # three identical points, ~zero noise -> singular
    y = np.array([1., 2., 3.])
    with pytest.raises(O.PosDefException):
        O.gp_fit(X, y, "sqexp", [1.], 1.0, 0.0)
    assert O.gp_data_loglike_slice(X, y, "sqexp", [1.], 1.0, 0.0) == -np.inf


@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
def test_gradient_restatement_against_finite_differences(kernel):
    """gp_mean_and_var_grad / expected_improvement_lin_grad (the analytic derivatives the device path
    of SURVEY §8f3 must reproduce) against central finite differences of the oracle itself."""
    rng = np.random.default_rng(5)
    d, N, M = 3, 40, 7
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(3 * X).sum(0) + 0.05 * rng.standard_normal(N)
    lam = np.array([0.4, 0.7, 0.5])
    post = O.gp_fit(X, y, kernel, lam, 1.3, 0.05, mean=np.full(N, 0.2))
    Xs = rng.uniform(0.05, 0.95, (d, M))
    ms = np.full(M, 0.2)
    mu, var, dmu, dvar = O.gp_mean_and_var_grad(post, Xs, ms)
    mu0, var0 = O.gp_mean_and_var(post, Xs, ms, clip=False)
    assert np.allclose(mu, mu0, rtol=0, atol=1e-13) and np.allclose(var, var0, rtol=0, atol=1e-13)
    eps = 1e-6
    b = float(y.max()) - 0.3
    ei, dei = O.expected_improvement_lin_grad([1.0], mu[None], var[None], dmu[None], dvar[None], b)
    for k in range(d):
        Xp, Xm = Xs.copy(), Xs.copy()
        Xp[k] += eps
        Xm[k] -= eps
        mp, vp = O.gp_mean_and_var(post, Xp, ms, clip=False)
        mm, vm = O.gp_mean_and_var(post, Xm, ms, clip=False)
        assert np.allclose(dmu[k], (mp - mm) / (2 * eps), rtol=1e-6, atol=1e-7)
        assert np.allclose(dvar[k], (vp - vm) / (2 * eps), rtol=1e-6, atol=1e-7)
        eip = O.expected_improvement_lin([1.0], mp[None], vp[None], b)
        eim = O.expected_improvement_lin([1.0], mm[None], vm[None], b)
        assert np.allclose(dei[k], (eip - eim) / (2 * eps), rtol=1e-5, atol=1e-8)
    # discrete dimensions: rounded inside the kernel -> zero gradient
    postd = O.gp_fit(X * 4, y, kernel, lam * 4, 1.3, 0.05, discrete=[True, False, False])
    _, _, dmud, dvard = O.gp_mean_and_var_grad(postd, Xs * 4)
    assert np.all(dmud[0] == 0) and np.all(dvard[0] == 0) and np.any(dmud[1] != 0)


def test_acquisition_gradient_restatement_against_finite_differences():
    """ei_acquisition_grad (EI × feasibility, two constrained outputs, prior-mean gradient) against
    central finite differences of ei_acquisition."""
    rng = np.random.default_rng(12)
    d, N, M, P = 2, 35, 9, 2
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), X[0] - X[1]])
    lam = [np.array([0.4, 0.6]), np.array([0.5, 0.5])]
    mean_f = [lambda Z: 0.1 + 0.2 * Z[0], lambda Z: -0.1 * Z[1]]
    mean_g = [np.array([0.2, 0.0]), np.array([0.0, -0.1])]
    posts = [O.gp_fit(X, Y[p], "matern52", lam[p], 1.0 + 0.2 * p, 0.05, mean=mean_f[p](X)) for p in range(P)]
    Xs = rng.uniform(0.05, 0.95, (d, M))
    y_max = [np.inf, 0.3]
    coefs = [1.0, 0.2]
    b = O.best_so_far(coefs, Y, y_max)
    ms = lambda Z: [mean_f[p](Z) for p in range(P)]
    mgs = [np.repeat(mean_g[p][:, None], M, axis=1) for p in range(P)]
    for ym, bb in ((y_max, b), (None, b), (y_max, None)):
        acq, dacq = O.ei_acquisition_grad(posts, Xs, coefs, ym, bb, means_s=ms(Xs), mean_grads_s=mgs)
        ref = O.ei_acquisition(posts, Xs, coefs, ym, bb, means_s=ms(Xs))
        assert np.allclose(acq, ref, rtol=0, atol=1e-14)
        eps = 1e-6
        for k in range(d):
            Xp, Xm = Xs.copy(), Xs.copy()
            Xp[k] += eps
            Xm[k] -= eps
            fd = (O.ei_acquisition(posts, Xp, coefs, ym, bb, means_s=ms(Xp)) -
                  O.ei_acquisition(posts, Xm, coefs, ym, bb, means_s=ms(Xm))) / (2 * eps)
            assert np.allclose(dacq[k], fd, rtol=1e-5, atol=1e-8), (ym, bb, k)


@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
def test_loglike_gradient_restatement_against_finite_differences(kernel):
    rng = np.random.default_rng(3)
    d, N = 3, 30
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(3 * X).sum(0) + 0.1 * rng.standard_normal(N)
    mean = 0.1 + 0.2 * X[0]
    lam, amp, sig = np.array([0.4, 0.7, 0.5]), 1.3, 0.1
    ll, g = O.gp_data_loglike_grad(X, y, kernel, lam, amp, sig, mean=mean)
    assert abs(ll - O.gp_data_loglike_slice(X, y, kernel, lam, amp, sig, mean=mean)) <= 1e-12
    th = np.concatenate([lam, [amp, sig]])
    eps = 1e-6
    for k in range(d + 2):
        tp, tm = th.copy(), th.copy()
        tp[k] += eps
        tm[k] -= eps
        fd = (O.gp_data_loglike_slice(X, y, kernel, tp[:d], tp[d], tp[d + 1], mean=mean) -
              O.gp_data_loglike_slice(X, y, kernel, tm[:d], tm[d], tm[d + 1], mean=mean)) / (2 * eps)
        assert abs(g[k] - fd) <= 1e-5 * (1 + abs(fd)), (k, g[k], fd)


# ------------------------------------------------------------------------------------------
# GradientGaussianProcess restatement: the reference holds no tests or fixtures for
# this model, so the oracle is pinned by finite differences of the plain kernel and by the
# properties the construction must have.
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", [O.MATERN32, O.MATERN52, O.SQEXP])
def test_augmented_kernel_blocks_are_kernel_derivatives(kernel):
    rng = np.random.default_rng(0)
    d, n = 3, 6
    X = rng.uniform(0, 1, (d, n))
    lam, amp = np.array([0.5, 0.8, 1.1]), 1.3
    K = O.augmented_kernel_matrix(kernel, X, lam, amp, 0.0, 0.0)
    amp2, lm = (amp + 1e-8) ** 2, lam + 1e-8

    def k(a, b):
        return amp2 * float(O.kappa(kernel, np.sqrt(np.sum(((a - b) / lm) ** 2))))
    h = 1e-5
    for i in range(n):
        for j in range(n):
            assert abs(K[i, j] - k(X[:, i], X[:, j])) <= 1e-14
            if i == j:
                continue
            for l in range(d):
                e = np.zeros(d)
                e[l] = h
                fd = (k(X[:, i], X[:, j] + e) - k(X[:, i], X[:, j] - e)) / (2 * h)
                assert abs(fd - K[i, n + l * n + j]) <= 1e-8                      # ∂k/∂(x_j)_l
                for m in range(d):
                    f = np.zeros(d)
                    f[m] = h
                    fd2 = (k(X[:, i] + e, X[:, j] + f) - k(X[:, i] + e, X[:, j] - f) - k(X[:, i] - e, X[:, j] + f)
                           + k(X[:, i] - e, X[:, j] - f)) / (4 * h * h)
                    assert abs(fd2 - K[n + l * n + i, n + m * n + j]) <= 5e-6     # ∂²k/∂(x_i)_l∂(x_j)_m
    assert np.array_equal(K, K.T)                                   # Symmetric(K): the upper triangle mirrored
    assert np.linalg.eigvalsh(K).min() > 0


def test_augmented_cross_cov_matches_the_matrix_columns():
    """k*(x*) for x* equal to a training point (shifted off the coincidence tolerance) is that point's column of
    the function block and of the gradient blocks."""
    rng = np.random.default_rng(1)
    d, n = 2, 5
    X = rng.uniform(0, 1, (d, n))
    lam = np.array([0.6, 0.9])
    K = O.augmented_kernel_matrix(O.MATERN52, X, lam, 1.0, 0.0, 0.0)
    xs = X[:, [3]] + 1e-6
    Kx = O.augmented_kernel_matrix(O.MATERN52, np.hstack([xs, X]), lam, 1.0, 0.0, 0.0)
    ks = O.augmented_cross_cov(O.MATERN52, X, lam, 1.0, xs)[:, 0]
    n1 = n + 1
    want = np.concatenate([Kx[0, 1:n1]] + [Kx[0, n1 + l * n1 + 1:n1 + (l + 1) * n1] for l in range(d)])
    assert np.allclose(ks, want, rtol=0, atol=1e-14)
    assert np.allclose(ks, np.concatenate([K[3, :n]] + [K[3, n + l * n:n + (l + 1) * n] for l in range(d)]), atol=1e-4)


def test_gradient_gp_interpolates_values_and_gradients():
    rng = np.random.default_rng(2)
    d, n = 2, 25
    X = rng.uniform(0, 1, (d, n))
    y = np.sin(3 * X[0]) * np.cos(2 * X[1])
    dY = np.stack([3 * np.cos(3 * X[0]) * np.cos(2 * X[1]), -2 * np.sin(3 * X[0]) * np.sin(2 * X[1])])
    post = O.gradient_gp_fit(X, y, dY, O.MATERN52, [0.5, 0.5], 1.0, 1e-3, 1e-3)
    mu, var = O.gradient_gp_mean_and_var(post, X)
    assert np.allclose(mu, y, atol=1e-4) and var.max() < 1e-5
    h = 1e-5
    for l in range(d):
        e = np.zeros((d, 1))
        e[l] = h
        g = (O.gradient_gp_mean_and_var(post, X + e)[0] - O.gradient_gp_mean_and_var(post, X - e)[0]) / (2 * h)
        assert np.allclose(g, dY[l], atol=5e-3)
    # gradients carry information: the same data without them predict worse between the points
    Xt = rng.uniform(0.1, 0.9, (d, 200))
    truth = np.sin(3 * Xt[0]) * np.cos(2 * Xt[1])
    plain = O.gp_fit(X, y, O.MATERN52, [0.5, 0.5], 1.0, 1e-3)
    err_g = np.abs(O.gradient_gp_mean_and_var(post, Xt)[0] - truth).max()
    err_p = np.abs(O.gp_mean_and_var(plain, Xt)[0] - truth).max()
    assert err_g < err_p


def test_gradient_gp_with_uninformative_gradients_reduces_to_the_plain_gp():
    rng = np.random.default_rng(3)
    d, n = 2, 20
    X = rng.uniform(0, 1, (d, n))
    y = np.sin(3 * X).sum(0)
    dY = rng.standard_normal((d, n))
    lam = np.array([0.4, 0.7])
    post = O.gradient_gp_fit(X, y, dY, O.MATERN52, lam, 1.1, 0.05, 1e6)
    plain = O.gp_fit(X, y, O.MATERN52, lam, 1.1, 0.05)
    Xt = rng.uniform(0, 1, (d, 30))
    mu_g, var_g = O.gradient_gp_mean_and_var(post, Xt)
    mu_p, var_p = O.gp_mean_and_var(plain, Xt, clip=False)
    assert np.allclose(mu_g, mu_p, atol=1e-8) and np.allclose(var_g, var_p, atol=1e-8)


# ------------------------------------------------------------------------------------------
# NonstationaryGP restatement: constant latent parameters reduce the Gibbs kernel to the ARD
# squared-exponential one, i.e. to the pinned plain-GP oracle.
# ------------------------------------------------------------------------------------------
def test_gibbs_kernel_with_constant_parameters_is_the_sqexp_gp():
    rng = np.random.default_rng(4)
    d, N, M = 3, 40, 25
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(3 * X).sum(0)
    Xs = rng.uniform(0, 1, (d, M))
    lam, amp, sig = np.array([0.4, 0.7, 1.1]), 1.3, 0.05
    plain = O.gp_fit(X, y, O.SQEXP, lam - 1e-8, amp - 1e-8, sig - 1e-8)          # the plain model adds 1e-8, this one does not
    ones = np.ones
    post = O.nonstationary_fit(X, y, lam[:, None] * ones((d, N)), amp * ones(N), sig * ones(N))
    assert abs(post.logpdf - plain.logpdf) <= 1e-9 * (1 + abs(plain.logpdf))
    mu, var = O.nonstationary_mean_and_var(post, Xs, lam[:, None] * ones((d, M)), amp * ones(M))
    mu_p, var_p = O.gp_mean_and_var(plain, Xs)
    assert np.allclose(mu, mu_p, rtol=0, atol=1e-9) and np.allclose(var, var_p, rtol=0, atol=1e-9)


def test_gibbs_kernel_is_positive_definite_and_symmetric_for_varying_lengthscales():
    rng = np.random.default_rng(5)
    d, N = 2, 60
    X = rng.uniform(0, 1, (d, N))
    lam = 0.1 + X ** 2                                  # λ_i(x) varies with the input
    amp = 1.0 + 0.5 * X[0]
    K = O.gibbs_kernel_matrix(X, lam, amp, X, lam, amp)
    assert np.allclose(K, K.T, rtol=0, atol=1e-15) and np.allclose(np.diag(K), amp ** 2)
    assert np.linalg.eigvalsh(K).min() > -1e-10
    # the scalar definition (nonstationary_gp.jl:101-103) for one pair
    k = ((amp[3] + amp[7]) / 2) ** 2
    for i in range(d):
        s = lam[i, 3] ** 2 + lam[i, 7] ** 2
        k *= np.sqrt(2 * lam[i, 3] * lam[i, 7] / s) * np.exp(-(X[i, 3] - X[i, 7]) ** 2 / s)
    assert abs(K[3, 7] - k) <= 1e-15


def test_gradient_gp_candidate_gradients_match_finite_differences():
    """oracle.gradient_gp_mean_and_var_grad (the analytic ∇μ, ∇σ² of a GradientGaussianProcess posterior,
    /root/reference/src/models/gradient_gp.jl:334-361 under the automatic differentiation of
    src/acquisition_maximizers/optimization.jl:36) against central differences of gradient_gp_mean_and_var's formulas."""
    import scipy.linalg as sla
    from oracle import gp_oracle as O
    rng = np.random.default_rng(0)
    for kern in ("matern32", "matern52", "sqexp"):
        d, n, M = 3, 12, 5
        X = rng.uniform(0, 1, (d, n))
        w = np.linspace(1, 2, d)[:, None]
        y = np.sin(2 * np.pi * w * X).sum(0)
        dY = 2 * np.pi * w * np.cos(2 * np.pi * w * X)
        post = O.gradient_gp_fit(X, y, dY, kern, np.array([0.4, 0.5, 0.6]), 1.2, 1e-2, 2e-2)
        Xs = rng.uniform(0, 1, (d, M))
        mu, var, dmu, dvar = O.gradient_gp_mean_and_var_grad(post, Xs)
        mu0, var0 = O.gradient_gp_mean_and_var(post, Xs)
        assert np.allclose(mu, mu0, rtol=0, atol=1e-12) and np.allclose(np.maximum(var, 0), var0, rtol=0, atol=1e-12)

        def f(Z):
            Ks = O.augmented_cross_cov(post.kernel, post.X, post.lengthscale, post.amplitude, Z)
            V = sla.solve_triangular(post.L, Ks, lower=True)
            return Ks.T @ post.alpha, (post.amplitude + 1e-8) ** 2 - np.sum(V * V, axis=0)
        e = 1e-6
        for m in range(d):
            Zp, Zm = Xs.copy(), Xs.copy()
            Zp[m] += e
            Zm[m] -= e
            (mp, vp), (mm, vm) = f(Zp), f(Zm)
            assert np.abs((mp - mm) / (2 * e) - dmu[m]).max() <= 1e-6 * (1 + np.abs(dmu[m]).max())
            assert np.abs((vp - vm) / (2 * e) - dvar[m]).max() <= 1e-6 * (1 + np.abs(dvar[m]).max())


def test_nonstationary_candidate_gradients_match_finite_differences():
    """oracle.nonstationary_mean_and_var_grad (Gibbs kernel with the candidate entering through λ(x*), α(x*) as well,
    /root/reference/src/models/nonstationary_gp/nonstationary_gp.jl:61-107,153-196) against central differences."""
    from oracle import gp_oracle as O
    rng = np.random.default_rng(1)
    d, N, M = 3, 25, 6
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(3 * X).sum(0)
    A = rng.uniform(0.2, 0.5, (d, d))
    b0 = np.array([0.4, 0.5, 0.6])
    lamf = lambda Z: b0[:, None] + 0.2 * np.sin(A @ Z)                       # noqa: E731
    dlamf = lambda z: 0.2 * np.cos(A @ z)[:, None] * A                       # noqa: E731
    ca = np.array([0.3, -0.2, 0.1])
    ampf = lambda Z: 1.0 + 0.3 * np.tanh(ca @ Z)                             # noqa: E731
    dampf = lambda z: 0.3 * (1 - np.tanh(ca @ z) ** 2) * ca                  # noqa: E731
    post = O.nonstationary_fit(X, y, lamf(X), ampf(X), np.full(N, 0.05), mean=0.1 * X[0])
    Xs = rng.uniform(0, 1, (d, M))
    Dl = np.stack([dlamf(Xs[:, j]) for j in range(M)], axis=2)
    Da = np.stack([dampf(Xs[:, j]) for j in range(M)], axis=1)
    mg = np.zeros((d, M))
    mg[0] = 0.1
    mu, var, dmu, dvar = O.nonstationary_mean_and_var_grad(post, Xs, lamf(Xs), ampf(Xs), Dl, Da, 0.1 * Xs[0], mg)
    f = lambda Z: O.nonstationary_mean_and_var(post, Z, lamf(Z), ampf(Z), 0.1 * Z[0], clip=False)   # noqa: E731
    mu0, var0 = f(Xs)
    assert np.allclose(mu, mu0, rtol=0, atol=1e-12) and np.allclose(var, var0, rtol=0, atol=1e-12)
    e = 1e-6
    for m in range(d):
        Zp, Zm = Xs.copy(), Xs.copy()
        Zp[m] += e
        Zm[m] -= e
        (mp, vp), (mm, vm) = f(Zp), f(Zm)
        assert np.abs((mp - mm) / (2 * e) - dmu[m]).max() <= 1e-6 * (1 + np.abs(dmu[m]).max())
        assert np.abs((vp - vm) / (2 * e) - dvar[m]).max() <= 1e-6 * (1 + np.abs(dvar[m]).max())


@pytest.mark.parametrize("kernel", [O.MATERN32, O.MATERN52, O.SQEXP])
def test_gradient_gp_likelihood_gradient_matches_finite_differences(kernel):
    """oracle.gradient_gp_loglike_grad (∂ℓ/∂(λ, α, σ, σ_∂) of the gradient-observation model, incl. the third radial profile and the
    upper-triangle convention of `Symmetric(K)`) against central differences of gradient_gp_fit(...).logpdf — also with a repeated
    training point, whose derivative blocks the reference evaluates at x_j + 1e-8 (gradient_gp.jl:148-152)."""
    rng = np.random.default_rng(7)
    for d, n, dup in ((1, 5, False), (3, 9, False), (2, 8, True)):
        X = rng.uniform(0, 1, (d, n))
        if dup:
            X[:, 1] = X[:, 0]
        y = np.sin(3 * X).sum(0)
        dY = 3 * np.cos(3 * X)
        th = np.concatenate([rng.uniform(0.4, 0.8, d), [1.2, 0.05, 0.08]])
        f = lambda t: O.gradient_gp_fit(X, y, dY, kernel, t[:d], t[d], t[d + 1], t[d + 2]).logpdf
        ll, g = O.gradient_gp_loglike_grad(X, y, dY, kernel, th[:d], th[d], th[d + 1], th[d + 2])
        assert ll == f(th)
        fd = np.zeros(d + 3)
        for i in range(d + 3):
            e = np.zeros(d + 3)
            e[i] = 1e-6 * max(1.0, abs(th[i]))
            fd[i] = (f(th + e) - f(th - e)) / (2 * e[i])
        assert np.abs(g - fd).max() <= 2e-6 * (1 + np.abs(fd).max()), (kernel, d, n, dup, g, fd)


@pytest.mark.parametrize("disc", [False, True])
def test_nonstationary_likelihood_gradient_matches_finite_differences(disc):
    """oracle.nonstationary_loglike_grad (∂ℓ/∂ of the latent values λ(x_i), α(x_i), σ(x_i), m(x_i) at the training points) against
    central differences of nonstationary_fit(...).logpdf, with and without a rounded dimension."""
    rng = np.random.default_rng(3)
    d, N = 3, 12
    discrete = np.array([False, True, False]) if disc else None
    X = rng.uniform(0, 1, (d, N))
    if disc:
        X[1] *= 4
    y = np.sin(3 * X).sum(0)
    lam, amp, noi, m = rng.uniform(0.3, 0.9, (d, N)), rng.uniform(0.8, 1.3, N), rng.uniform(0.03, 0.1, N), 0.2 * X[0]
    ll, dl, da, dn, dm = O.nonstationary_loglike_grad(X, y, lam, amp, noi, mean=m, discrete=discrete)
    f = lambda lam, amp, noi, m: O.nonstationary_fit(X, y, lam, amp, noi, mean=m, discrete=discrete).logpdf   # noqa: E731
    assert ll == f(lam, amp, noi, m)
    e = 1e-6

    def fd(arr, idx, which):
        p, q = arr.copy(), arr.copy()
        p[idx] += e
        q[idx] -= e
        args = {"lam": (p, amp, noi, m), "amp": (lam, p, noi, m), "noi": (lam, amp, p, m), "m": (lam, amp, noi, p)}[which]
        args2 = {"lam": (q, amp, noi, m), "amp": (lam, q, noi, m), "noi": (lam, amp, q, m), "m": (lam, amp, noi, q)}[which]
        return (f(*args) - f(*args2)) / (2 * e)
    for l in range(d):
        for i in (0, 5, N - 1):
            assert abs(fd(lam, (l, i), "lam") - dl[l, i]) <= 1e-6 * (1 + abs(dl[l, i]))
    for i in (0, 4, N - 1):
        assert abs(fd(amp, i, "amp") - da[i]) <= 1e-6 * (1 + abs(da[i]))
        assert abs(fd(noi, i, "noi") - dn[i]) <= 1e-6 * (1 + abs(dn[i]))
        assert abs(fd(m, i, "m") - dm[i]) <= 1e-6 * (1 + abs(dm[i]))

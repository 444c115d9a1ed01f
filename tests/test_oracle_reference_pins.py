"""Pins the CPU oracle to every known-answer / property test the reference holds for the hot path.

Ports (file:line in /root/reference):
  test/unit/test/models/gaussian_process.jl:57-148,150-239   posterior properties
  test/unit/test/models/gaussian_process.jl:241-259          _clip_var exact thresholds
  test/unit/test/models/gaussian_process.jl:261-316          likelihood orderings
  test/unit/test/acquisitions/expected_improvement.jl:26-40,55-179   EI / feas_prob / best_so_far
  test/unit/test/models/semiparametric.jl:2-193              same properties with a parametric mean
  test/unit/test/models/utils/kernels.jl:13-46               DiscreteKernel rounding identities
  test/unit/test/posterior.jl:2-30                           average_mean
"""
import math

import numpy as np
import pytest

from oracle import gp_oracle as O

X3 = np.array([[2., 5., 8.], [2., 5., 8.]])          # [2.;2.;; 5.;5.;; 8.;8.;;]
Y3 = X3.copy()


def sampling_map(X, Y, mean, rng, samples=200, noise=1e-4, kernel="matern52"):
    """SamplingMAP(samples=200) (src/model_fitters/sampling.jl:59-78) with the test's priors:
    amplitude LogNormal(), lengthscale mvlognormal([1,1],[1,1]), noise Dirac(1e-4)."""
    d, P = X.shape[0], Y.shape[0]
    best, best_v = None, -math.inf
    for _ in range(samples):
        lam = np.exp(1.0 + rng.standard_normal((d, P)))
        amp = np.exp(rng.standard_normal(P))
        sig = np.full(P, noise)
        v = O.data_loglike(X, Y, kernel, lam, amp, sig, means=mean)
        # params_loglike: LogNormal / MvLogNormal prior logpdfs (gaussian_process.jl:282-289)
        v += float(np.sum(-np.log(amp) - 0.5 * np.log(amp) ** 2 - 0.5 * math.log(2 * math.pi)))
        v += float(np.sum(-np.log(lam) - 0.5 * (np.log(lam) - 1.0) ** 2 - 0.5 * math.log(2 * math.pi)))
        if v > best_v:
            best, best_v = (lam, amp, sig), v
    return best


@pytest.fixture(scope="module")
def fitted_posts():
    rng = np.random.default_rng(7)
    means = [np.ones(3), np.ones(3)]                   # mean = x -> [1., 1.]
    lam, amp, sig = sampling_map(X3, Y3, means, rng)
    posts = [O.gp_fit(X3, Y3[i], "matern52", lam[:, i], amp[i], sig[i], mean=1.0) for i in range(2)]
    return posts


def mv(posts, x):
    x = np.asarray(x, dtype=float)
    Xs = x[:, None] if x.ndim == 1 else x
    ms = [np.ones(Xs.shape[1])] * len(posts)
    mu, var = O.model_mean_and_var(posts, Xs, ms)
    return (mu[:, 0], var[:, 0]) if x.ndim == 1 else (mu, var)


def test_gp_posterior_properties(fitted_posts):
    p = fitted_posts
    # gaussian_process.jl:94-101
    for pt in ([2., 2.], [5., 5.], [8., 8.]):
        assert np.allclose(mv(p, pt)[0], pt, atol=0.01)
    assert np.all(mv(p, [1., 1.])[0] < 2.0)
    assert np.all(mv(p, [4., 4.])[0] < 5.0)
    assert np.allclose(mv(p, [100., 100.])[0], [1., 1.], atol=0.01)
    assert np.all(mv(p, [2., 2.])[1] <= mv(p, [3., 3.])[1])
    assert np.all(mv(p, [10., 10.])[1] <= mv(p, [11., 11.])[1])
    # matrix vs vector consistency, :115-126
    Xm = np.array([[1., 2., 3.], [1., 2., 3.]])
    mu, var = mv(p, Xm)
    assert mu.shape == (2, 3) and var.shape == (2, 3)
    for j in range(3):
        muj, varj = mv(p, Xm[:, j])
        assert np.allclose(mu[:, j], muj, atol=1e-8) and np.allclose(var[:, j], varj, atol=1e-8)
    # single-column matrix, :128-138
    mu1, var1 = mv(p, np.array([[1.], [1.]]))
    assert mu1.shape == (2, 1) and var1.shape == (2, 1)
    # mean_and_cov diagonal == var (:119-120)
    for i in range(2):
        m, S = O.gp_mean_and_cov(p[i], Xm, np.ones(3))
        assert S.shape == (3, 3)
        assert np.allclose(np.diag(S), var[i], atol=1e-8) and np.allclose(m, mu[i], atol=1e-8)


def test_clip_var_exact():
    # gaussian_process.jl:241-259
    for v in (0., 1e-9, 1e-8, 1e-7, 1.):
        assert O.clip_var(v) == v
    for v in (-1e-9, -1e-8):
        assert O.clip_var(v) == 0.
    with pytest.raises(O.DomainError):
        O.clip_var(-1e-7)


def test_model_loglike_orderings():
    # gaussian_process.jl:261-283 (data part; prior part is host-side bookkeeping)
    means = [np.ones(3), np.ones(3)]

    def ll(lam, amp, sig):
        return O.data_loglike(X3, Y3, "matern52", np.full((2, 2), lam), [amp, amp], [sig, sig], means=means)
    assert ll(1., 1., 1.) < 0.
    assert ll(1., 1., 5.) > ll(1., 1., 100.)
    assert ll(1., 5., 1.) > ll(1., 100., 1.)
    assert ll(5., 1., 1.) > ll(100., 1., 1.)


def test_data_loglike_orderings():
    # gaussian_process.jl:287-316; note noise 0 -> 1e-8 via MIN_PARAM_VALUE
    X = np.array([[1., 2., 3.]])

    def out(Y, lam=1., amp=1., sig=0.):
        return O.data_loglike(X, np.array([Y]), "matern52", np.array([[lam]]), [amp], [sig], means=[np.zeros(3)])
    alt = [1., -1., 1.]
    assert isinstance(out([1., 2., 3.], 1., 1., 1.), float)
    assert out(alt, lam=0.1) > out(alt, lam=1.) > out(alt, lam=10.)
    assert out(alt, amp=1.) > out(alt, amp=0.1)
    assert out(alt, sig=1.) > out(alt, sig=0.1)
    assert out(alt, sig=1.) > out(alt, sig=10.)
    assert out([99.9, 100., 100.1]) > out([99., 100., 101.]) > out([90., 100., 110.])


def test_min_param_value_offsets():
    # gaussian_process.jl:239-241: zeros become 1e-8 rather than erroring
    h = O.finite_gp_params("matern32", 2, [0., 1.], 0., 0.)
    assert np.allclose(h.lengthscale, [1e-8, 1. + 1e-8]) and h.amplitude == 1e-8 and h.noise_std == 1e-8
    with pytest.raises(AssertionError):
        O.finite_gp_params("matern32", 2, [-1., 1.], 1., 1.)
    with pytest.raises(AssertionError):
        O.finite_gp_params("matern32", 2, [1.], 1., 1.)      # :233 length(lengthscales)==x_dim


# ------------------------------------------------------------------ expected improvement
def test_expected_improvement_known_answers():
    # expected_improvement.jl(test):108-142, LinFitness([1., 0.])
    c = [1., 0.]
    ei = lambda mu, var, b: float(O.expected_improvement_lin(c, np.array(mu)[:, None], np.array(var)[:, None], b)[0])
    assert ei([0., 0.], [1., 1.], 0.) > 0.
    assert ei([0., 0.], [0., 0.], 0.) == 0.
    assert ei([1., 1.], [0., 0.], 0.) == 1.
    assert abs(ei([-10., -10.], [1., 1.], 0.)) <= 1e-20


def test_feas_prob_known_answers():
    # expected_improvement.jl(test):144-163
    fp = lambda mu, var, ym: float(O.feas_prob(np.array(mu)[:, None], np.array(var)[:, None], ym)[0])
    assert fp([0., 0.], [0., 0.], None) == 1.
    assert fp([0., 0.], [1., 1.], None) == 1.
    assert fp([np.inf, np.inf], [1., 1.], None) == 1.
    assert abs(fp([0., 0.], [1., 1.], [np.inf, np.inf]) - 1.) <= 1e-20
    assert abs(fp([0., 0.], [1., 1.], [0., np.inf]) - 0.5) <= 1e-20
    assert abs(fp([0., 0.], [1., 1.], [0., 0.]) - 0.25) <= 1e-20
    assert 0.99 < fp([0., 0.], [1., 1.], [3., np.inf]) < 1.


def test_best_so_far():
    # expected_improvement.jl(test):165-179
    Y = np.array([[1., 2., 3.]])
    assert O.best_so_far([1.], Y, [np.inf]) == 3.
    assert O.best_so_far([1.], Y, [5.]) == 3.
    assert O.best_so_far([1.], np.array([[10., 2., 3.]]), [5.]) == 3.
    assert O.best_so_far([2.], Y, [np.inf]) == 6.
    assert O.best_so_far([1.], Y, [0.]) is None
    assert O.best_so_far([1.], np.zeros((1, 0)), [0.]) is None


class _IdentityPosterior:
    """ParametricPosterior(f=identity, noise_std=[1,1]) of the reference test (:58-61):
    mean(x) = x, var(x) = [1,1]."""


def _construct_ei_identity(y_max, best, n_post=1):
    c = [1., 0.]

    def acq(x):
        mu = np.array(x, dtype=float)[:, None]
        var = np.ones((2, 1))
        tot = 0.0
        for _ in range(n_post):
            if y_max is None and best is None:
                a = 0.0
            elif best is None:
                a = float(O.feas_prob(mu, var, y_max)[0])
            elif y_max is None:
                a = float(O.expected_improvement_lin(c, mu, var, best)[0])
            else:
                a = float(O.expected_improvement_lin(c, mu, var, best)[0] * O.feas_prob(mu, var, y_max)[0])
            tot += a
        return tot / n_post
    return acq


@pytest.mark.parametrize("n_post", [1, 4])
def test_construct_ei_orderings(n_post):
    # expected_improvement.jl(test):55-105 (LinFitness rows)
    out = _construct_ei_identity(None, None, n_post)
    assert out([1., 1.]) == 0.
    out = _construct_ei_identity([np.inf, 10.], None, n_post)
    assert out([1., 1.]) > 0.
    assert out([5., 1.]) == out([10., 1.]) == out([15., 1.])
    assert out([1., 5.]) > out([1., 10.]) > out([1., 15.])
    assert abs(out([1., 20.])) <= 1e-8
    out = _construct_ei_identity(None, 10., n_post)
    assert out([11., 1.]) > 0.
    assert out([5., 1.]) < out([10., 1.]) < out([15., 1.])
    assert abs(out([0., 1.])) <= 1e-8
    assert out([1., 5.]) == out([1., 10.]) == out([1., 15.])
    out = _construct_ei_identity([np.inf, 10.], 10., n_post)
    assert out([11., 1.]) > 0.
    assert out([5., 1.]) < out([10., 1.]) < out([15., 1.])
    assert out([11., 5.]) > out([11., 10.]) > out([11., 15.])
    assert abs(out([1., 1.])) <= 1e-8
    assert abs(out([11., 20.])) <= 1e-8


def test_make_safe_mask():
    # expected_improvement.jl(test):26-40: 0. outside bounds, bounds inclusive
    lb, ub = [5.], [10.]
    Xs = np.array([[1., 5., 7., 10., 11.]])
    mask = O.in_bounds(Xs, lb, ub)
    assert mask.tolist() == [False, True, True, True, False]
    acq = np.where(mask, Xs[0], 0.0)
    assert acq.tolist() == [0., 5., 7., 10., 0.]


def test_ei_acquisition_end_to_end(fitted_posts):
    # construct_acquisition (expected_improvement.jl:49-56) on the GP fixture: y_max=[Inf, 5.]
    y_max = [np.inf, 5.]
    b = O.best_so_far([1., 0.], Y3, y_max)
    assert b == 5.                                   # columns (2,2),(5,5) feasible; (8,8) not
    Xs = np.array([[1., 4., 6., 12.], [1., 4., 6., 12.]])
    mask = O.in_bounds(Xs, [0., 0.], [10., 10.])
    acq = O.ei_acquisition(fitted_posts, Xs, [1., 0.], y_max, b, valid_mask=mask,
                           means_s=[np.ones(4), np.ones(4)])
    assert acq.shape == (4,) and acq[3] == 0.0 and np.all(acq >= 0.0)


# ------------------------------------------------------------------ semiparametric
def test_semiparametric_posterior_properties():
    # semiparametric.jl(test):2-70: GP with parametric prior mean m(x;θ); same property suite
    X = X3
    f = lambda x: np.array([math.sin(x[0]) + math.exp(x[1]), math.cos(x[0]) + math.exp(x[1])])
    Y = np.stack([f(X[:, j]) for j in range(3)], axis=1)
    theta = np.array([0.9, 1.0, 1.1, 1.0])
    par = lambda x: np.array([theta[0] * math.sin(x[0]) + theta[1] * math.exp(x[1]),
                              theta[2] * math.cos(x[0]) + theta[3] * math.exp(x[1])])
    posts = [O.gp_fit(X, Y[i], "matern52", [2., 2.], 1.0, 1e-4, mean=(lambda x, i=i: par(x)[i])) for i in range(2)]

    def mvp(x):
        x = np.asarray(x, float)
        Xs = x[:, None]
        ms = [np.array([par(Xs[:, j])[i] for j in range(Xs.shape[1])]) for i in range(2)]
        mu, var = O.model_mean_and_var(posts, Xs, ms)
        return mu[:, 0], var[:, 0]
    for j in range(3):
        assert np.allclose(mvp(X[:, j])[0], Y[:, j], atol=0.01)
    assert np.all(mvp([2., 2.])[1] <= mvp([3., 3.])[1])
    assert np.all(mvp([10., 10.])[1] <= mvp([11., 11.])[1])
    # far from data the posterior reverts to the parametric mean
    assert np.allclose(mvp([100., 3.])[0], par([100., 3.]), atol=0.01)


# ------------------------------------------------------------------ DiscreteKernel
def test_discrete_kernel_identities():
    # kernels.jl(test):13-46
    def k(x1, x2, discrete):
        h = O.finite_gp_params("matern32", 2, [1. - 1e-8, 1. - 1e-8], 1. - 1e-8, 0., discrete)
        return float(O.kernelmatrix(h, np.array(x1)[:, None], np.array(x2)[:, None])[0, 0])
    base = lambda x1, x2: k(x1, x2, None)
    assert k([1.2, 1.2], [3.8, 3.8], [False, False]) == base([1.2, 1.2], [3.8, 3.8])
    assert k([1.2, 2.], [3.8, 4.], [False, False]) == base([1.2, 2.], [3.8, 4.])
    assert k([1.2, 1.2], [3.8, 3.8], [False, True]) == k([1.2, 1.], [3.8, 4.], [False, True])
    assert k([1.2, 1.2], [3.8, 4.], [False, True]) == k([1.2, 1.], [3.8, 4.], [False, True])
    assert k([1.2, 1.2], [3.8, 3.8], [False, True]) != base([1.2, 1.2], [3.8, 3.8])
    # Julia round() is half-to-even
    assert O.discrete_round(np.array([[0.5, 1.5, 2.5]]), [True]).tolist() == [[0., 2., 2.]]


def test_average_mean():
    # posterior.jl(test):2-30
    rng = np.random.default_rng(3)
    samples = []
    for _ in range(8):
        lam = np.exp(1 + rng.standard_normal((2, 2)))
        amp = np.exp(rng.standard_normal(2))
        samples.append([O.gp_fit(X3, Y3[i], "matern52", lam[:, i], amp[i], 1e-4) for i in range(2)])
    for pt in ([3., 3.], [5., 5.]):
        Xs = np.array(pt)[:, None]
        want = sum(O.model_mean_and_var(p, Xs)[0] for p in samples) / len(samples)
        assert np.allclose(O.average_mean(samples, Xs), want)

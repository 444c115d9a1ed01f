"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs, against the committed golden fixtures, and — at BASELINE.json's full
sizes — through size-independent properties.

Stated fp64 tolerance (BASELINE.md §2 / SURVEY §8c):
    |Δμ| <= 1e-9 (1+|μ|),  |Δσ²| <= 1e-9 α²,  |Δlogpdf| <= 1e-9 (1+|logpdf|)   for cond(K) <= 1e6,
    condition-aware  cond(K)·N·2⁻⁵³·C  beyond that.
"""
import json
import os

import numpy as np
import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    entry.build()
    from boss_jl_amd import api as a
    a.load_library()
    assert a.device_count() >= 1
    return a


@pytest.fixture(scope="module")
def O():
    from oracle import gp_oracle
    return gp_oracle


def make(d, N, M, seed=1, noise=0.05, scale=1.0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, scale, (d, N))
    y = np.sin(2 * np.pi * X / scale).sum(0) / np.sqrt(d) + noise * rng.standard_normal(N)
    Xs = np.random.default_rng(seed + 1000).uniform(0, scale, (d, M))
    return X, y, Xs


def tol_for(O, post, N):
    K = O.kernelmatrix(post.h, post.X)
    K[np.diag_indices(N)] += post.h.noise_std ** 2
    c = np.linalg.cond(K)
    return max(1e-9, c * N * 2.0 ** -53 * 8), c


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
@pytest.mark.parametrize("d,N,M", [(1, 1, 1), (1, 3, 4), (2, 16, 5), (8, 127, 31), (8, 128, 32), (8, 129, 33), (3, 300, 70), (8, 1000, 257)])
def test_fit_predict_parity(api, O, kernel, d, N, M):
    X, y, Xs = make(d, N, M)
    lam = np.linspace(0.3, 0.7, d)
    post = O.gp_fit(X, y, kernel, lam, 1.2, 0.05)
    mu_o, var_o = O.gp_mean_and_var(post, Xs, clip=False)
    tol, _ = tol_for(O, post, N)
    g = api.GP(X, y, kernel)
    lp = g.update(lam, 1.2, 0.05)
    assert abs(lp - post.logpdf) <= tol * (1 + abs(post.logpdf))
    L, z = g.factor()
    assert np.allclose(L, post.L, rtol=0, atol=tol * np.abs(post.L).max())
    z_o = O.sla.solve_triangular(post.L, post.delta, lower=True)
    assert np.allclose(z, z_o, rtol=0, atol=tol * (1 + np.abs(z_o).max()))
    mu, var = g.predict(Xs)
    assert np.allclose(mu, mu_o, rtol=0, atol=tol * (1 + np.abs(mu_o).max()))
    assert np.allclose(var, O.clip_var(var_o), rtol=0, atol=tol * 1.2 ** 2)
    # vector form == column of the matrix form (gaussian_process.jl test :121-126, atol 1e-8)
    mu1, var1 = g.predict(Xs[:, 0])
    assert abs(mu1[0] - mu[0]) <= 1e-12 * (1 + abs(mu[0])) and abs(var1[0] - var[0]) <= 1e-12
    g.close()


@pytest.mark.parametrize("noise", [1e-1, 1e-2, 1e-4])
def test_condition_aware_tolerance_sweep(api, O, noise):
    """σ sweep of BASELINE.md §2: the error stays inside cond(K)·N·eps as the matrix gets stiffer."""
    X, y, Xs = make(8, 512, 64, seed=7, noise=noise)
    lam = np.full(8, 0.5)
    post = O.gp_fit(X, y, "matern52", lam, 1.0, noise)
    mu_o, var_o = O.gp_mean_and_var(post, Xs, clip=False)
    tol, cond = tol_for(O, post, 512)
    g = api.GP(X, y, "matern52")
    lp = g.update(lam, 1.0, noise)
    mu, var = g.predict(Xs)
    assert abs(lp - post.logpdf) <= tol * (1 + abs(post.logpdf))
    assert np.allclose(mu, mu_o, rtol=0, atol=tol * (1 + np.abs(mu_o).max()))
    assert np.allclose(var, O.clip_var(var_o), rtol=0, atol=tol)
    g.close()


def test_golden_fixtures(api, O):
    """The committed fixtures at the stated 1e-9 bar (condition-aware beyond cond(K) = 1e6), the whole factor
    included: packed in the fixture for N <= 130, against the oracle on the fixture's inputs above that."""
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "gp_golden.json")))
    for c in cases:
        X, y, Xs = np.array(c["X"]), np.array(c["y"]), np.array(c["Xs"])
        N = X.shape[1]
        post = O.gp_fit(X, y, c["kernel"], c["lengthscale"], c["amplitude"], c["noise_std"], mean=c["mean_X"], discrete=c["discrete"])
        tol, _ = tol_for(O, post, N)                       # 1e-9 unless the fixture is stiff (ref_fixture_3pt: σ = 1e-4)
        g = api.fit(X, y, c["kernel"], c["lengthscale"], c["amplitude"], c["noise_std"], c["mean_X"], c["discrete"])
        assert abs(g.logpdf - c["logpdf"]) <= tol * (1 + abs(c["logpdf"])), c["name"]
        L, z = g.factor()
        if c["L_packed"] is not None:
            Lg = np.zeros((N, N))
            pk, o = np.array(c["L_packed"]), 0
            for j in range(N):
                Lg[j:, j] = pk[o:o + N - j]
                o += N - j
        else:
            Lg = post.L
        assert np.array_equal(np.triu(L, 1), np.zeros((N, N))), c["name"]
        assert np.allclose(L, Lg, rtol=0, atol=tol * np.abs(Lg).max()), c["name"]
        assert np.allclose(np.diag(L), c["L_diag"], rtol=1e-10, atol=0), c["name"]
        assert np.allclose(z, c["z"], rtol=0, atol=tol * (1 + np.abs(c["z"]).max())), c["name"]
        try:
            mu, var = g.predict(Xs, c["mean_Xs"])
        except api.DomainError:
            assert min(c["var"]) < -1e-8
            continue
        assert np.allclose(mu, c["mu"], rtol=0, atol=tol * (1 + np.abs(c["mu"]).max())), c["name"]
        assert np.allclose(var, np.maximum(c["var"], 0.0), rtol=0, atol=tol * c["amplitude"] ** 2), c["name"]
        cand = api.Candidates(Xs)
        ms = None if c["mean_Xs"] is None else np.array(c["mean_Xs"])[None, None, :]
        acq, am, mx = api.acq_ei([[g]], cand, [1.0], [np.inf], c["best"], None, ms)
        assert np.allclose(acq, c["acq_ei"], rtol=0, atol=tol), c["name"]
        assert am == c["argmax"] or abs(acq[am] - c["acq_ei"][c["argmax"]]) <= 1e-12, c["name"]
        g.close()


def test_gemm_form_distances_on_the_reference_fixture(api, O):
    """KernelFunctions evaluates pairwise distances through Distances.jl's ‖a‖²+‖b‖²−2a·b; the HIP
    kernels take direct differences.  On the reference's own stiff fixture (X = [2 5 8; 2 5 8], noise Dirac(1e-4),
    test/unit/test/models/gaussian_process.jl:57-73) and on a bench-shaped problem the device results stay inside the
    condition-aware bound of the oracle evaluated in the GEMM form."""
    X3 = np.array([[2., 5., 8.], [2., 5., 8.]])
    Xs = np.array([[1., 2., 3., 4., 5.5, 100.], [1., 2., 3., 4., 5.5, 100.]])
    ms = np.ones(Xs.shape[1])
    for kernel in ("matern52", "matern32", "sqexp"):
        for lam in ([2.5, 3.5], [1.0, 1.0], [6.0, 9.0], [40.0, 60.0], [200.0, 300.0]):   # cond(K) from 1 to 7e7
            pg = O.gp_fit(X3, X3[0], kernel, lam, 1.3, 1e-4, mean=1.0, form="gemm")
            mu_g, var_g = O.gp_mean_and_var(pg, Xs, ms, clip=False, form="gemm")
            tol, cond = tol_for(O, pg, 3)
            # the two distance forms differ by O(eps ‖x/λ‖²) in r², amplified by cond(K) in the posterior
            scale = max(1.0, float(np.max(np.sum((X3 / np.array(lam)[:, None]) ** 2, axis=0))))
            tol = max(tol, cond * 2.0 ** -53 * 16 * scale)
            g = api.fit(X3, X3[0], kernel, lam, 1.3, 1e-4, np.ones(3))
            assert abs(g.logpdf - pg.logpdf) <= tol * (1 + abs(pg.logpdf)), (kernel, lam)
            mu, var = g.predict(Xs, ms)
            assert np.allclose(mu, mu_g, rtol=0, atol=tol * (1 + np.abs(mu_g).max())), (kernel, lam, cond)
            assert np.allclose(var, np.where(var_g >= 0, var_g, 0.0), rtol=0, atol=tol * 1.3 ** 2), (kernel, lam, cond)
            g.close()
    X, y, Xc = make(8, 512, 64, seed=5)
    lam = np.full(8, 0.5)
    pg = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05, form="gemm")
    mu_g, var_g = O.gp_mean_and_var(pg, Xc, clip=False, form="gemm")
    g = api.fit(X, y, "matern52", lam, 1.0, 0.05)
    mu, var = g.predict(Xc)
    assert abs(g.logpdf - pg.logpdf) <= 1e-9 * (1 + abs(pg.logpdf))
    assert np.allclose(mu, mu_g, rtol=0, atol=1e-9 * (1 + np.abs(mu_g).max())) and np.allclose(var, var_g, rtol=0, atol=1e-9)
    g.close()


def test_semiparametric_mean_and_discrete(api, O):
    X, y, Xs = make(3, 150, 40, seed=3, scale=4.0)
    lam = np.array([1.5, 2.0, 1.0])
    disc = [False, True, False]
    mX = 0.5 + 0.2 * X.sum(0)
    ms = 0.5 + 0.2 * Xs.sum(0)
    post = O.gp_fit(X, y, "matern52", lam, 0.9, 0.05, mean=mX, discrete=disc)
    mu_o, var_o = O.gp_mean_and_var(post, Xs, ms)
    g = api.GP(X, y, "matern52", discrete=disc)
    lp = g.update(lam, 0.9, 0.05, mean_X=mX)
    mu, var = g.predict(Xs, ms)
    assert abs(lp - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf))
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-9 * (1 + np.abs(mu_o).max())) and np.allclose(var, var_o, rtol=0, atol=1e-9)
    # switching back to the zero mean on the same handle
    lp0 = g.update(lam, 0.9, 0.05)
    assert abs(lp0 - O.gp_fit(X, y, "matern52", lam, 0.9, 0.05, discrete=disc).logpdf) <= 1e-9 * (1 + abs(lp0))
    g.close()


def test_zero_params_get_min_param_value(api, O):
    """gaussian_process.jl:239-241: zero λ/α/σ are lifted to 1e-8, not rejected."""
    X = np.array([[1., 2., 3.]])
    y = np.array([1., -1., 1.])
    want = O.gp_fit(X, y, "matern52", [1.0], 1.0, 0.0).logpdf
    g = api.GP(X, y, "matern52")
    assert abs(g.update([1.0], 1.0, 0.0) - want) <= 1e-7 * (1 + abs(want))
    for bad in ([-1.0], ):
        with pytest.raises(api.BossError) as e:
            g.update(bad, 1.0, 0.1)
        assert e.value.code == api.BOSS_E_INVALID
    with pytest.raises(api.BossError):
        g.update([1.0], -1.0, 0.1)
    with pytest.raises(api.BossError):
        g.update([1.0, 1.0], 1.0, 0.1)
    g.close()


def test_not_positive_definite(api, O):
    X = np.array([[1., 1., 1., 2.]])          # duplicates + ~zero noise -> singular
    y = np.array([1., 2., 3., 4.])
    assert O.gp_data_loglike_slice(X, y, "sqexp", [1.], 1.0, 0.0) == -np.inf
    g = api.GP(X, y, "sqexp")
    with pytest.raises(api.PosDefException):
        g.update([1.], 1.0, 0.0)
    with pytest.raises(api.BossError) as e:
        g.predict(np.array([[1.5]]))
    assert e.value.code == api.BOSS_E_NOT_FITTED
    assert np.isfinite(g.update([1.], 1.0, 0.1))       # handle recovers
    g.close()


def test_loglike_batch(api, O):
    X, y, _ = make(4, 200, 1, seed=11)
    rng = np.random.default_rng(4)
    S = 9
    lam = np.exp(-0.7 + 0.3 * rng.standard_normal((4, S)))
    amp = np.exp(0.3 * rng.standard_normal(S))
    sig = np.exp(-3 + 0.3 * rng.standard_normal(S))
    amp[3] = -1.0                                        # invalid set -> BOSS_E_INVALID, -Inf
    ll, st = api.loglike_batch(X, y, "matern52", lam, amp, sig)
    for s in range(S):
        if s == 3:
            assert st[s] == api.BOSS_E_INVALID and ll[s] == -np.inf
            continue
        want = O.gp_data_loglike_slice(X, y, "matern52", lam[:, s], amp[s], sig[s])
        assert st[s] == api.BOSS_OK and abs(ll[s] - want) <= 1e-9 * (1 + abs(want))
    # shared and per-sample prior means
    m_shared = 0.1 * X.sum(0)
    ll2, _ = api.loglike_batch(X, y, "sqexp", lam[:, :2], np.abs(amp[:2]), sig[:2], mean_X=m_shared)
    m_per = np.stack([m_shared, -m_shared])
    ll3, _ = api.loglike_batch(X, y, "sqexp", lam[:, :2], np.abs(amp[:2]), sig[:2], mean_X=m_per)
    for s in range(2):
        w2 = O.gp_data_loglike_slice(X, y, "sqexp", lam[:, s], abs(amp[s]), sig[s], mean=m_shared)
        w3 = O.gp_data_loglike_slice(X, y, "sqexp", lam[:, s], abs(amp[s]), sig[s], mean=m_per[s])
        assert abs(ll2[s] - w2) <= 1e-9 * (1 + abs(w2)) and abs(ll3[s] - w3) <= 1e-9 * (1 + abs(w3))
    # a non-PD member does not poison its neighbours
    Xd = np.array([[1., 1., 1., 2.]])
    yd = np.array([1., 2., 3., 4.])
    ll4, st4 = api.loglike_batch(Xd, yd, "sqexp", np.array([[1., 1.]]), [1., 1.], [0.0, 0.5])
    assert st4[0] == api.BOSS_E_NOT_PD and ll4[0] == -np.inf
    assert abs(ll4[1] - O.gp_data_loglike_slice(Xd, yd, "sqexp", [1.], 1., 0.5)) <= 1e-9


@pytest.mark.parametrize("mode", ["none", "cons_only", "best_only", "both"])
def test_ei_modes_multi_output_and_samples(api, O, mode):
    """construct_ei's four variants + BI averaging over S=3 samples, P=2 outputs, masked candidates."""
    d, N, M, P, S = 3, 120, 77, 2, 3
    rng = np.random.default_rng(21)
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), np.cos(2 * X).sum(0) - 1.0]) + 0.05 * rng.standard_normal((P, N))
    Xs = np.asfortranarray(rng.uniform(-0.1, 1.1, (d, M)))
    y_max = None if mode in ("none", "best_only") else np.array([np.inf, 0.3])
    coefs = [1.0, 0.5]
    best = None if mode in ("none", "cons_only") else O.best_so_far(coefs, Y, [np.inf, 0.3])
    mask = O.in_bounds(Xs, [0.] * d, [1.] * d)
    gps, posts = [], []
    for s in range(S):
        lam = np.exp(-0.5 + 0.2 * rng.standard_normal((d, P)))
        amp = np.exp(0.2 * rng.standard_normal(P))
        gps.append([api.fit(X, Y[p], "matern52", lam[:, p], amp[p], 0.05) for p in range(P)])
        posts.append([O.gp_fit(X, Y[p], "matern52", lam[:, p], amp[p], 0.05) for p in range(P)])
    cand = api.Candidates(Xs)
    acq, am, mx = api.acq_ei(gps, cand, coefs, y_max, best, mask)
    want = O.ei_acquisition(posts, Xs, coefs, y_max, best, valid_mask=mask)
    assert np.allclose(acq, want, rtol=0, atol=1e-10)
    assert am == int(np.argmax(acq)) and mx == acq[am]
    if mode == "none":
        assert np.all(acq == 0.0)
    assert np.all(acq[~mask] == 0.0)


def test_ei_known_answers_through_device(api):
    """EI / feas_prob known answers (test/unit/test/acquisitions/expected_improvement.jl:108-163)
    pushed through the device epilogue with a nearly deterministic posterior (far from data the GP
    reverts to its prior: μ = m(x), σ² = α²)."""
    X = np.array([[0.0]])
    y = np.array([0.0])
    far = np.array([[1e6]])
    cand = api.Candidates(far)

    def acq(amp, mean_s, y_max, best):
        g = api.fit(X, y, "sqexp", [1.0], amp, 1.0)
        a, _, _ = api.acq_ei([[g]], cand, [1.0], y_max, best, None, np.array(mean_s, float).reshape(1, 1, 1))
        return a[0]
    assert acq(1.0, [0.0], None, 0.0) > 0.0                       # EI(μ=0, σ²=1, b=0) > 0
    assert abs(acq(0.0, [1.0], None, 0.0) - 1.0) <= 1e-12         # EI(μ=1, σ²≈0, b=0) == 1
    assert abs(acq(1.0, [-10.0], None, 0.0)) <= 1e-20             # EI(μ=-10, σ²=1) ≈ 0
    assert abs(acq(1.0, [0.0], [0.0], None) - 0.5) <= 1e-12       # feas_prob = 0.5
    assert acq(1.0, [0.0], [np.inf], None) == 1.0                 # Infinity -> cdf = 1
    assert 0.99 < acq(1.0, [0.0], [3.0], None) < 1.0


def test_argmax_first_index_on_ties(api):
    X = np.array([[0.0]])
    g = api.fit(X, np.array([0.0]), "sqexp", [1.0], 1.0, 1.0)
    Xs = np.full((1, 70), 1e6)                                       # identical candidates -> identical acq
    a, am, mx = api.acq_ei([[g]], api.Candidates(Xs), [1.0], None, 0.0)
    assert np.all(a == a[0]) and am == 0
    mask = np.ones(70, bool)
    mask[:5] = False                                                 # masked -> 0.0 < EI
    a, am, _ = api.acq_ei([[g]], api.Candidates(Xs), [1.0], None, 0.0, mask)
    assert am == 5 and np.all(a[:5] == 0.0)


def test_plugin_trio_reference_properties(api):
    """The reference's own GP posterior property suite (test/unit/test/models/gaussian_process.jl:57-148)
    and one BO iteration, driven through the host mirror of the plugin interface."""
    import boss_jl_amd as B
    from boss_jl_amd.bo import bo_step, estimate_parameters
    X3 = np.array([[2., 5., 8.], [2., 5., 8.]])
    model = B.HipGaussianProcess(lengthscale_priors=[B.MvLogNormal([1., 1.], [1., 1.])] * 2,
                                 amplitude_priors=[B.LogNormal()] * 2, noise_std_priors=[B.Dirac(1e-4)] * 2,
                                 mean=lambda x: [1., 1.])
    problem = B.BossProblem(f=lambda x: x, domain=B.Domain(bounds=([0., 0.], [10., 10.])), y_max=[np.inf, 5.],
                            acquisition=B.ExpectedImprovement(B.LinFitness([1., 0.])), model=model,
                            data=B.ExperimentData(X3, X3.copy()))
    fitted = estimate_parameters(problem, B.HipBatchedMAP(samples=200, seed=7))
    assert np.isfinite(fitted.loglike)
    post = model.model_posterior(problem.params, problem.data)
    m = lambda x: post.mean(np.array(x, float))
    v = lambda x: post.var(np.array(x, float))
    for pt in ([2., 2.], [5., 5.], [8., 8.]):
        assert np.allclose(m(pt), pt, atol=0.01)
    assert np.all(m([1., 1.]) < 2.0) and np.all(m([4., 4.]) < 5.0)
    assert np.allclose(m([100., 100.]), [1., 1.], atol=0.01)
    assert np.all(v([2., 2.]) <= v([3., 3.])) and np.all(v([10., 10.]) <= v([11., 11.]))
    Xm = np.array([[1., 2., 3.], [1., 2., 3.]])
    mu, var = post.mean_and_var(Xm)
    assert mu.shape == (2, 3) and var.shape == (2, 3)
    for j in range(3):
        muj, varj = post.mean_and_var(Xm[:, j])
        assert np.allclose(mu[:, j], muj, atol=1e-8) and np.allclose(var[:, j], varj, atol=1e-8)
    assert np.allclose(post.std(Xm), np.sqrt(var))
    # one BO iteration: the maximiser returns an in-domain point and the dataset grows
    am = B.HipBatchAM(x_prior=lambda rng: rng.uniform(0, 10, 2), samples=300, seed=3)
    x, val = bo_step(problem, B.HipBatchedMAP(samples=50, seed=1), am)
    assert x.shape == (2,) and np.all(x >= 0) and np.all(x <= 10) and np.isfinite(val)
    assert problem.data.X.shape == (2, 4) and not problem.consistent


# ------------------------------------------------------------------------------------------
# full BASELINE size: size-independent properties
# ------------------------------------------------------------------------------------------
def test_full_size_properties(api, O):
    d, N, M = 8, 4096, 8192
    X, y, Xs = make(d, N, M, seed=1)
    lam = np.full(d, 0.5)
    g = api.GP(X, y, "matern52")
    lp = g.update(lam, 1.0, 0.05)
    L, z = g.factor()
    h = O.finite_gp_params("matern52", d, lam, 1.0, 0.05)
    K = O.kernelmatrix(h, X)
    K[np.diag_indices(N)] += h.noise_std ** 2
    # (1) L L^T reconstructs K: backward-stable Cholesky bound  ||LL^T - K|| <= c N eps ||K||
    R = L @ L.T - K
    assert np.abs(R).max() <= 50 * N * 2.0 ** -53 * np.abs(K).max()
    # (2) L z = y and the log-likelihood identity
    assert np.allclose(L @ z, y, rtol=0, atol=1e-10 * (1 + np.abs(y).max()))
    assert abs(lp - (-0.5 * (N * np.log(2 * np.pi) + 2 * np.log(np.diag(L)).sum() + z @ z))) <= 1e-9 * (1 + abs(lp))
    # (3) against the oracle (LAPACK) at full size
    post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
    assert abs(lp - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf))
    # (4) acquisition over all 8192 candidates: a 512-candidate slice equals the oracle, the batch
    #     result restricted to that slice equals the slice-only result, and arg-max is consistent
    cand = api.Candidates(Xs)
    b = float(y.max())
    acq, am, mx = api.acq_ei([[g]], cand, [1.0], None, b)
    assert am == int(np.argmax(acq)) and mx == acq[am]
    sl = slice(1000, 1512)
    want = O.ei_acquisition([post], Xs[:, sl], [1.0], None, b)
    assert np.allclose(acq[sl], want, rtol=0, atol=1e-10)
    acq_sl, _, _ = api.acq_ei([[g]], api.Candidates(Xs[:, sl]), [1.0], None, b)
    assert np.allclose(acq_sl, acq[sl], rtol=0, atol=1e-13)
    # (5) interpolation property: predicting AT the data recovers y to within the noise level
    mu, var = g.predict(X[:, :256])
    assert np.abs(mu - y[:256]).max() < 0.2 and np.all(var >= 0) and np.all(var < 0.05 ** 2 + 1e-6)
    g.close()


def test_config4_semiparametric_two_constrained_outputs_full_size(api, O):
    """BASELINE.json configs[3]: Semiparametric model (affine parametric mean + GP), 2 outputs constrained by
    y_max = [Inf, 0.5], N = 2048, d = 6, 8192 candidates (src/models/semiparametric.jl:79-92 for the posterior and
    likelihood, src/acquisitions/expected_improvement.jl:68-90 for EI x feasibility) — through the C ABI, through the
    plugin mirror, and with the outputs sharded (`shard="outputs"`, world size 1)."""
    import boss_jl_amd as B
    rng = np.random.default_rng(3)
    d, N, M, P = 6, 2048, 8192, 2
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), np.cos(2 * X).sum(0) - 1.0]) + 0.05 * rng.standard_normal((P, N))
    Xs = np.asfortranarray(rng.uniform(0, 1, (d, M)))
    th = np.array([0.1, 0.2, -0.3])
    w = np.linspace(0.5, 1.5, d)
    para = lambda x, t: np.array([t[0] + t[1] * float(w @ x), t[0] + t[2] * float(w @ x)])    # m(x; θ) = θ₁ + θ₂ᵀx per output
    mX = np.stack([th[0] + th[1] * (w @ X), th[0] + th[2] * (w @ X)])
    mS = np.stack([th[0] + th[1] * (w @ Xs), th[0] + th[2] * (w @ Xs)])
    lam = np.stack([np.full(d, 0.5), np.linspace(0.4, 0.7, d)], axis=1)
    amp, sig = np.array([1.0, 0.8]), np.array([0.05, 0.07])
    coefs, y_max = [1.0, 0.0], np.array([np.inf, 0.5])
    b = O.best_so_far(coefs, Y, y_max)
    assert b is not None
    posts = [O.gp_fit(X, Y[p], "matern52", lam[:, p], amp[p], sig[p], mean=mX[p]) for p in range(P)]
    # (1) posterior construction and likelihood (semiparametric.jl:86-92 = Σ_p logpdf with the parametric mean)
    gps = [api.GP(X, Y[p], "matern52") for p in range(P)]
    lps = [gps[p].update(lam[:, p], amp[p], sig[p], mean_X=mX[p]) for p in range(P)]
    for p in range(P):
        assert abs(lps[p] - posts[p].logpdf) <= 1e-9 * (1 + abs(posts[p].logpdf))
    # (2) EI x feasibility over all 8192 candidates: a 512-candidate slice against the oracle, whole-batch arg-max
    #     consistent, and the slice-only call equal to the slice of the batch
    cand = api.Candidates(Xs)
    acq, am, mx = api.acq_ei([gps], cand, coefs, y_max, b, None, mS[None])
    assert am == int(np.argmax(acq)) and mx == acq[am] and np.all(acq >= 0) and np.isfinite(acq).all()
    sl = slice(3000, 3512)
    want = O.ei_acquisition(posts, Xs[:, sl], coefs, y_max, b, means_s=[mS[0, sl], mS[1, sl]])
    assert np.allclose(acq[sl], want, rtol=0, atol=1e-10)
    acq_sl, _, _ = api.acq_ei([gps], api.Candidates(Xs[:, sl]), coefs, y_max, b, None, mS[None, :, sl])
    assert np.allclose(acq_sl, acq[sl], rtol=0, atol=1e-13)
    # moments of both outputs on a slice
    for p in range(P):
        mu, var = gps[p].predict(Xs[:, sl], mS[p, sl])
        mu_o, var_o = O.gp_mean_and_var(posts[p], Xs[:, sl], mS[p, sl])
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9 * (1 + np.abs(mu_o).max())) and np.allclose(var, var_o, rtol=0, atol=1e-9)
    # the constraint matters: the unconstrained acquisition differs, and feasibility only ever lowers it
    acq_u, _, _ = api.acq_ei([gps], cand, coefs, None, b, None, mS[None])
    assert np.all(acq <= acq_u + 1e-15) and np.any(acq < 0.9 * acq_u)
    for g in gps:
        g.close()
    # (3) the same through the plugin mirror (Semiparametric = HipGaussianProcess(parametric=...)), candidates and
    #     outputs sharding modes; the two agree with each other and with (2)
    model = B.HipGaussianProcess([None] * P, [None] * P, [None] * P, parametric=para, theta_priors=[None] * 3)
    prm = B.HipGPParams(lam, amp, sig, th)
    prob = B.BossProblem(None, B.Domain((np.zeros(d), np.ones(d))), B.ExpectedImprovement(B.LinFitness(coefs)), model,
                         B.ExperimentData(X, Y), y_max, prm)
    ll = model.data_loglike(prob.data)
    assert abs(ll(prm) - sum(q.logpdf for q in posts)) <= 1e-9 * (1 + abs(sum(q.logpdf for q in posts)))
    for h in ll.handles:
        h.close()
    res = {}
    for mode in ("candidates", "outputs"):
        amx = B.HipBatchAM(points=Xs, shard=mode)
        x, val = amx.maximize_acquisition(prob)
        _, allv = amx.maximize_acquisition(prob, return_all=True)
        res[mode] = (x, val, allv)
        assert np.allclose(allv, acq, rtol=0, atol=1e-12), mode
        assert np.array_equal(x, Xs[:, am]) and abs(val - mx) <= 1e-12, mode


def test_config5_bi_512_samples_full_size(api, O):
    """BASELINE.json configs[4]: TuringBI-style fit — S = 512 posterior hyper-parameter samples, each with its own
    N = 1024 Cholesky (ext/TuringExt.jl:88-107 draws them; every one is a data_loglike evaluation,
    gaussian_process.jl:250-280), then the acquisition averaged over the S resident posteriors
    (src/posterior.jl:15-19, expected_improvement.jl:87-90)."""
    d, N, S, M = 8, 1024, 512, 256
    X, y, Xs = make(d, N, M, seed=4)
    rng = np.random.default_rng(4)
    lam = np.exp(-0.7 + 0.3 * rng.standard_normal((d, S)))
    amp = np.exp(0.3 * rng.standard_normal(S))
    sig = np.exp(-3 + 0.3 * rng.standard_normal(S))
    ll, st = api.loglike_batch(X, y, "matern52", lam, amp, sig)
    assert np.all(st == api.BOSS_OK) and np.isfinite(ll).all()
    # (1) 40 of the 512 sets against the oracle (first, last, and a spread in between)
    idx = sorted(set([0, 1, 7, S - 1] + list(range(3, S, 14))))
    assert len(idx) >= 32
    for s in idx:
        want = O.gp_data_loglike_slice(X, y, "matern52", lam[:, s], amp[s], sig[s])
        assert abs(ll[s] - want) <= 1e-9 * (1 + abs(want)), s
    # (2) every member of the batch equals the single-handle update of the same set (the batched schedule is a
    #     different code path: paired panels, one stream) on a further 24 sets
    g = api.GP(X, y, "matern52")
    for s in range(5, S, 22):
        one = g.update(lam[:, s], amp[s], sig[s])
        assert abs(one - ll[s]) <= 1e-11 * (1 + abs(one)), s
    # (3) ordering property (test/unit/test/models/gaussian_process.jl:261-316): with everything else fixed the
    #     likelihood prefers the generating noise level to a grossly wrong one — on all 512 sets at once
    ll_bad, st_bad = api.loglike_batch(X, y, "matern52", lam, amp, np.full(S, 5.0))
    assert np.all(st_bad == api.BOSS_OK) and np.all(ll > ll_bad)
    # (4) a permutation of the sets permutes the results (no cross-talk between batch members)
    perm = np.random.default_rng(0).permutation(S)
    ll_p, _ = api.loglike_batch(X, y, "matern52", lam[:, perm], amp[perm], sig[perm])
    assert np.array_equal(ll_p, ll[perm])
    # (5) BI-averaged acquisition over resident posteriors: 16 of the samples stay resident, EI is the mean over them
    sub = list(range(0, S, 32))
    gps = [[api.fit(X, y, "matern52", lam[:, s], amp[s], sig[s])] for s in sub]
    posts = [[O.gp_fit(X, y, "matern52", lam[:, s], amp[s], sig[s])] for s in sub]
    b = float(y.max())
    acq, am, mx = api.acq_ei(gps, api.Candidates(Xs), [1.0], [np.inf], b)
    want = O.ei_acquisition(posts, Xs, [1.0], [np.inf], b)
    assert np.allclose(acq, want, rtol=0, atol=1e-10) and am == int(np.argmax(acq)) and mx == acq[am]
    # average_mean (src/posterior.jl:177-179) of the resident posteriors
    mus = np.mean([gp[0].predict(Xs)[0] for gp in gps], axis=0)
    assert np.allclose(mus, O.average_mean(posts, Xs)[0], rtol=0, atol=1e-9 * (1 + np.abs(mus).max()))
    for gp in gps:
        gp[0].close()
    g.close()


def test_full_size_next_rows(api, O):
    """BASELINE size (N=4096, d=8) for the rows built beyond the headline path: appended observations,
    tracked candidates, moment / acquisition / likelihood gradients — against fresh device results,
    oracle slices and central finite differences of the device's own values (size-independent checks)."""
    d, N, M = 8, 4096, 2048
    X, y, Xs = make(d, N + 5, M, seed=1)
    lam = np.full(d, 0.5)
    g = api.GP(X[:, :N], y[:N], "matern52")
    g.reserve(N + 5)
    lp = g.update(lam, 1.0, 0.05)
    cand = api.Candidates(Xs)
    tr = api.Track(g, cand)
    # (1) likelihood gradient == central differences of the device's own log-likelihood
    _, grad = g.loglike_grad()
    eps = 1e-5
    for k, (dl, da, ds) in enumerate([(np.eye(d)[0], 0, 0), (np.eye(d)[d - 1], 0, 0), (np.zeros(d), 1, 0), (np.zeros(d), 0, 1)]):
        fp = g.update(lam + eps * dl, 1.0 + eps * da, 0.05 + eps * ds)
        fm = g.update(lam - eps * dl, 1.0 - eps * da, 0.05 - eps * ds)
        want = (fp - fm) / (2 * eps)
        got = grad[[0, d - 1, d, d + 1][k]]
        assert abs(got - want) <= 2e-5 * (1 + abs(want)), (k, got, want)
    g.update(lam, 1.0, 0.05)
    tr.close()
    tr = api.Track(g, cand)
    # (2) five appended observations: the first (block path) is bit-identical to a fresh factorisation of the same data,
    #     the others (rank-one path on the resident inverses) agree to rounding; the tracked moments follow
    lpa = g.append(X[:, N], y[N])
    g1 = api.GP(X[:, :N + 1], y[:N + 1], "matern52")
    assert lpa == g1.update(lam, 1.0, 0.05)
    g1.close()
    for i in range(1, 5):
        lpa = g.append(X[:, N + i], y[N + i])
    g2 = api.GP(X, y, "matern52")
    lp2 = g2.update(lam, 1.0, 0.05)
    assert abs(lpa - lp2) <= 1e-11 * (1 + abs(lp2))
    mu_t, var_t = tr.moments()
    mu_f, var_f = g2.predict(Xs)
    assert np.allclose(mu_t, mu_f, rtol=0, atol=1e-10) and np.allclose(np.maximum(var_t, 0), var_f, rtol=0, atol=1e-10)
    # (3) moment gradients: oracle slice + central differences of the device prediction
    mu, var, dmu, dvar = g2.predict_grad(Xs)
    post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
    _, _, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs[:, :64])
    assert np.allclose(dmu[:, :64], dmu_o, rtol=0, atol=1e-9 * (1 + np.abs(dmu_o).max()))
    assert np.allclose(dvar[:, :64], dvar_o, rtol=0, atol=1e-9 * (1 + np.abs(dvar_o).max()))
    e = np.zeros((d, 1))
    e[3] = 1e-6
    mp, vp = g2.predict(Xs + e)
    mm, vm = g2.predict(Xs - e)
    assert np.allclose(dmu[3], (mp - mm) / 2e-6, rtol=0, atol=1e-5 * (1 + np.abs(dmu[3]).max()))
    assert np.allclose(dvar[3], (vp - vm) / 2e-6, rtol=0, atol=1e-5 * (1 + np.abs(dvar[3]).max()))
    # (4) acquisition gradient == chain rule through the oracle's EI formulas applied to the device moments
    b = float(y.max()) - 0.2
    acq, dacq = api.acq_ei_grad([g2], Xs, [1.0], None, b)
    ei, dei = O.expected_improvement_lin_grad([1.0], mu[None], var[None], dmu[None], dvar[None], b)
    assert np.allclose(acq, ei, rtol=0, atol=1e-12) and np.allclose(dacq, dei, rtol=0, atol=1e-10 * (1 + np.abs(dei).max()))
    tr.close()
    g.close()
    g2.close()


def test_posterior_covariance(api, O):
    """a5: mean_and_cov(post, X) (gaussian_process.jl:180-184) from the device-resident factor."""
    X, y, Xs = make(3, 200, 37, seed=13)
    lam = np.array([0.4, 0.5, 0.6])
    ms = 0.2 * Xs.sum(0)
    post = O.gp_fit(X, y, "matern52", lam, 1.1, 0.05, mean=0.2 * X.sum(0))
    mu_o, S_o = O.gp_mean_and_cov(post, Xs, ms)
    g = api.fit(X, y, "matern52", lam, 1.1, 0.05, 0.2 * X.sum(0))
    mu, S = g.predict_cov(Xs, ms)
    assert S.shape == (37, 37) and np.allclose(S, S.T, rtol=0, atol=1e-12)
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-9 * (1 + np.abs(mu_o).max()))
    assert np.allclose(S, S_o, rtol=0, atol=1e-9 * 1.1 ** 2)
    # diag(cov) == var (test/unit/test/models/gaussian_process.jl:119-120, atol 1e-8)
    _, var = g.predict(Xs, ms)
    assert np.allclose(np.diag(S), var, rtol=0, atol=1e-12)
    # through the plugin mirror: P outputs -> M×M×P
    import boss_jl_amd as B
    model = B.HipGaussianProcess(lengthscale_priors=[None] * 2, amplitude_priors=[None] * 2, noise_std_priors=[None] * 2)
    data = B.ExperimentData(X, np.stack([y, -y]))
    params = B.HipGPParams(np.stack([lam, lam], axis=1), [1.1, 0.9], [0.05, 0.05])
    pm = model.model_posterior(params, data)
    mus, covs = pm.mean_and_cov(Xs[:, :5])
    assert mus.shape == (2, 5) and covs.shape == (5, 5, 2)
    assert np.allclose(covs[:, :, 0], O.gp_mean_and_cov(O.gp_fit(X, y, "matern52", lam, 1.1, 0.05), Xs[:, :5])[1], atol=1e-9)
    g.close()


def test_nonlinear_fitness_host_epilogue(api, O):
    """a10: NonlinFitness EI = device (μ, σ²) + host Monte-Carlo average (expected_improvement.jl:104-111).
    Reference orderings (test/unit/test/acquisitions/expected_improvement.jl:55-105, nonlin rows) and
    agreement with the analytic LinFitness value for fit(y) = y[1]."""
    import boss_jl_amd as B
    from boss_jl_amd.maximizer import acquisition_values, posteriors_of
    rng = np.random.default_rng(5)
    d, N, M = 2, 60, 40
    X = rng.uniform(0, 10, (d, N))
    Y = np.stack([np.sin(X[0]) + 0.1 * X[1], 0.3 * X[0] - 1.0])
    mk = lambda fit: B.BossProblem(f=None, domain=B.Domain(bounds=([0., 0.], [10., 10.])), y_max=[np.inf, 1.5],
                                   acquisition=B.ExpectedImprovement(fit, eps_samples=4000),
                                   model=B.HipGaussianProcess(lengthscale_priors=[None] * 2, amplitude_priors=[None] * 2,
                                                              noise_std_priors=[None] * 2),
                                   data=B.ExperimentData(X, Y),
                                   params=B.HipGPParams(np.full((2, 2), 2.0), [1.0, 1.0], [0.05, 0.05]))
    Xs = np.asfortranarray(rng.uniform(-1, 11, (d, M)))
    p_lin, p_non = mk(B.LinFitness([1., 0.])), mk(B.NonlinFitness(lambda yy: yy[0]))
    a_lin, am_lin, _ = acquisition_values(p_lin, posteriors_of(p_lin), Xs)
    a_non, am_non, mx = acquisition_values(p_non, posteriors_of(p_non), Xs, eps_seed=1)
    inb = O.in_bounds(Xs, [0., 0.], [10., 10.])
    assert np.all(a_non[~inb] == 0.0) and np.all(a_non >= 0.0) and mx == a_non[am_non]
    # Monte-Carlo (4000 ε) vs analytic: a few percent of the acquisition's scale
    assert np.abs(a_non - a_lin).max() <= 0.05 * max(a_lin.max(), 1e-3) + 1e-4
    # BI: one ε column per posterior sample
    p_non.params = [p_non.params, B.HipGPParams(np.full((2, 2), 3.0), [1.2, 0.8], [0.05, 0.05])]
    a_bi, _, _ = acquisition_values(p_non, posteriors_of(p_non), Xs, eps_seed=2)
    assert a_bi.shape == (M,) and np.all(a_bi >= 0.0) and np.all(a_bi[~inb] == 0.0)


def test_domain_error_and_safe_acquisition(api, O):
    """_clip_var's DomainError (gaussian_process.jl:186-194): predicting AT well-separated training
    points with a huge amplitude and a tiny noise, the true variance (~σ²=1e-8) drowns in the rounding
    error of α² − ‖v‖² (~eps·α² = 1e-6), so variances below −1e-8 appear.  boss_gp_predict reports
    BOSS_E_NEG_VAR + the first offending index; the acquisition keeps going and marks those candidates
    −Inf, as the reference's SafeFunction wrapper does (src/acquisition.jl:21-25)."""
    X = np.arange(0.0, 400.0, 10.0)[None, :]
    y = np.sin(X[0])
    Xs = X.copy()
    lam, amp, sig = [3.0], 1e5, 1e-4
    post = O.gp_fit(X, y, "matern52", lam, amp, sig)
    with pytest.raises(O.DomainError):
        O.gp_mean_and_var(post, Xs)                  # the oracle (LAPACK) hits the same wall
    g = api.fit(X, y, "matern52", lam, amp, sig)
    with pytest.raises(api.DomainError) as e:
        g.predict(Xs)
    assert e.value.code == api.BOSS_E_NEG_VAR and 0 <= e.value.bad_index < Xs.shape[1]
    acq, am, mx = api.acq_ei([[g]], api.Candidates(Xs), [1.0], None, float(y.max()))
    assert np.isneginf(acq).any() and not np.isnan(acq).any()
    assert acq[am] == mx and mx == np.max(acq)
    # the unclipped device variances agree with the oracle's to the rounding level that causes the error
    _, var_o = O.gp_mean_and_var(post, Xs, clip=False)
    assert np.abs(var_o).max() < 1e-4


def test_ei_from_moments_and_shard_modes(api, O):
    """boss_acq_ei_moments (the EI x feas epilogue fed with gathered moments) and the three
    sharding modes of HipBatchAM agree with the fused path and with the oracle."""
    import boss_jl_amd as B
    rng = np.random.default_rng(21)
    d, N, M, P, S = 3, 90, 130, 2, 3
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), X[0] - X[1] + 0.3 * np.cos(5 * X[2])])
    Xs = np.asfortranarray(rng.uniform(-0.1, 1.1, (d, M)))
    y_max = np.array([np.inf, 0.4])
    coefs = [1.0, 0.25]
    prm = [B.HipGPParams(rng.uniform(0.3, 0.8, (d, P)), rng.uniform(0.8, 1.5, P), rng.uniform(0.03, 0.1, P)) for _ in range(S)]
    model = B.HipGaussianProcess([None] * P, [None] * P, [None] * P)
    prob = B.BossProblem(None, B.Domain((np.zeros(d), np.ones(d))), B.ExpectedImprovement(B.LinFitness(coefs)), model,
                         B.ExperimentData(X, Y), y_max, prm)
    oposts = [[O.gp_fit(X, Y[i], "matern52", p.lengthscales[:, i], p.amplitudes[i], p.noise_std[i]) for i in range(P)]
              for p in prm]
    b = O.best_so_far(coefs, Y, y_max)
    mask = O.in_bounds(Xs, np.zeros(d), np.ones(d))
    want = O.ei_acquisition(oposts, Xs, coefs, y_max, b, valid_mask=mask)
    # moments entry point against the oracle's moments
    mu = np.stack([np.stack([O.gp_mean_and_var(oposts[s][i], Xs)[0] for i in range(P)]) for s in range(S)])
    var = np.stack([np.stack([O.gp_mean_and_var(oposts[s][i], Xs)[1] for i in range(P)]) for s in range(S)])
    acq, am, mx = api.acq_ei_moments(mu, var, coefs, y_max, b, mask)
    assert np.allclose(acq, want, rtol=0, atol=1e-13) and am == int(np.argmax(acq)) and mx == acq[am]
    # poisoned variance -> -Inf for that candidate only
    var2 = var.copy()
    var2[1, 0, 7] = -1e-6
    acq2, _, _ = api.acq_ei_moments(mu, var2, coefs, y_max, b, None)
    assert acq2[7] == -np.inf and np.isfinite(np.delete(acq2, 7)).all()
    # the three sharding modes (world size 1 here; world size 2 is covered by the gloo test)
    res = {}
    for mode in ("candidates", "outputs", "samples"):
        am_ = B.HipBatchAM(points=Xs, shard=mode)
        x, val = am_.maximize_acquisition(prob)
        _, allv = am_.maximize_acquisition(prob, return_all=True)
        assert np.allclose(allv, want, rtol=0, atol=1e-12), mode
        res[mode] = (x, val)
    j = int(np.argmax(want))
    for mode, (x, val) in res.items():
        assert np.array_equal(x, Xs[:, j]) and abs(val - want[j]) <= 1e-12, mode


def test_multi_device_entry_points(api, O):
    """boss_init + boss_multi_*: ONE process driving the devices, exchanges over RCCL inside the library (what a Julia caller
    uses to shard without torch).  The test box has one GPU, so G = 1 here — the degenerate communicator still runs every
    collective (ncclCommInitAll over one device, all-gather of the 16-byte pair, all-reduce of the moment rows / partial sums) —
    plus host-side sharding over G = 1 of a larger communicator's code path via a candidate count that is not a multiple of
    anything.  Each mode against the single-device entry point and the oracle; then through HipBatchAM(devices=[0])."""
    import boss_jl_amd as B
    n = api.init()
    assert n >= 1 and api.init() == n                       # idempotent
    ndev, rccl = api.comm_info()
    assert ndev == n and rccl, "RCCL must load on the GPU box"
    rng = np.random.default_rng(33)
    d, N, M, P, S = 3, 150, 211, 2, 3
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), np.cos(2 * X).sum(0) - 1.0])
    Xs = np.asfortranarray(rng.uniform(-0.1, 1.1, (d, M)))
    y_max, coefs = np.array([np.inf, 0.3]), [1.0, 0.5]
    lam = [np.exp(-0.5 + 0.2 * rng.standard_normal((d, P))) for _ in range(S)]
    amp = [np.exp(0.2 * rng.standard_normal(P)) for _ in range(S)]
    mX = [0.1 * X.sum(0), -0.2 * X[0]]
    mS = np.stack([np.stack([0.1 * Xs.sum(0), -0.2 * Xs[0]])] * S)                       # S×P×M
    posts = [[O.gp_fit(X, Y[p], "matern52", lam[s][:, p], amp[s][p], 0.05, mean=mX[p]) for p in range(P)] for s in range(S)]
    b = O.best_so_far(coefs, Y, y_max)
    mask = O.in_bounds(Xs, [0.] * d, [1.] * d)
    want = O.ei_acquisition(posts, Xs, coefs, y_max, b, valid_mask=mask, means_s=[mS[0, 0], mS[0, 1]])
    # replicas through boss_multi_gp_update (every device factorises concurrently)
    gps = [[api.GP(X, Y[p], "matern52") for p in range(P)] for s in range(S)]
    for s in range(S):
        for p in range(P):
            lp = api.multi_update([gps[s][p]], lam[s][:, p], amp[s][p], 0.05, mX[p])
            assert abs(lp - posts[s][p].logpdf) <= 1e-9 * (1 + abs(posts[s][p].logpdf))
    single = api.acq_ei(gps, api.Candidates(Xs), coefs, y_max, b, mask, mS)
    for fn, arg in ((api.multi_acq_ei, [gps]), (api.multi_acq_ei_outputs, gps), (api.multi_acq_ei_samples, gps)):
        acq, am, mx = fn(arg, Xs, coefs, y_max, b, mask, mS)
        assert np.allclose(acq, want, rtol=0, atol=1e-10), fn.__name__
        assert np.allclose(acq, single[0], rtol=0, atol=1e-13) and (am, mx) == (single[1], acq[am]), fn.__name__
        assert am == int(np.argmax(acq))
        _, am2, mx2 = fn(arg, Xs, coefs, y_max, b, mask, mS, want_acq=False)
        assert (am2, mx2) == (am, mx)
    # the four construct_ei variants through the sharded entry point
    for ym_, b_ in ((None, None), (y_max, None), (None, b)):
        a1, _, _ = api.multi_acq_ei([gps], Xs, coefs, ym_, b_, None, mS)
        assert np.allclose(a1, O.ei_acquisition(posts, Xs, coefs, ym_, b_, means_s=[mS[0, 0], mS[0, 1]]), rtol=0, atol=1e-10)
    # errors: a sample whose outputs sit on different devices cannot happen with one GPU; unfitted handles and bad counts can
    g_un = api.GP(X, Y[0], "matern52")
    with pytest.raises(api.BossError) as e:
        api.multi_acq_ei_outputs([[g_un]], Xs, [1.0])
    assert e.value.code == api.BOSS_E_NOT_FITTED
    with pytest.raises(api.BossError):
        api.multi_acq_ei([gps] * (n + 1), Xs, coefs)       # more replicas than devices
    g_un.close()
    # likelihood batch split over the devices == the single-device batch
    lamS = np.exp(-0.7 + 0.3 * rng.standard_normal((d, 7)))
    ampS, sigS = np.exp(0.3 * rng.standard_normal(7)), np.full(7, 0.05)
    ll_m, st_m = api.multi_loglike_batch(1, X, Y[0], "matern52", lamS, ampS, sigS, mean_X=mX[0])
    ll_1, st_1 = api.loglike_batch(X, Y[0], "matern52", lamS, ampS, sigS, mean_X=mX[0])
    assert np.array_equal(ll_m, ll_1) and np.array_equal(st_m, st_1)
    # the plugin mirror in single-process mode
    prm = [B.HipGPParams(lam[s], amp[s], np.full(P, 0.05)) for s in range(S)]
    model = B.HipGaussianProcess([None] * P, [None] * P, [None] * P, mean=lambda x: [0.1 * x.sum(), -0.2 * x[0]])
    prob = B.BossProblem(None, B.Domain((np.zeros(d), np.ones(d))), B.ExpectedImprovement(B.LinFitness(coefs)), model,
                         B.ExperimentData(X, Y), y_max, prm)
    j = int(np.argmax(want))
    for mode in ("candidates", "outputs", "samples"):
        amx = B.HipBatchAM(points=Xs, shard=mode, devices=[0])
        x, val = amx.maximize_acquisition(prob)
        _, allv = amx.maximize_acquisition(prob, return_all=True)
        assert np.allclose(allv, want, rtol=0, atol=1e-10), mode
        assert np.array_equal(x, Xs[:, j]) and abs(val - want[j]) <= 1e-10, mode
    for row in gps:
        for g in row:
            g.close()
    api.shutdown()
    assert api.init() == n                                  # re-initialisation after a shutdown


@pytest.mark.parametrize("G", [2, 3])
def test_multi_device_entry_points_on_virtual_devices(G):
    """csrc/host_multi.inc with G > 1 on the one-GPU test box: BOSS_VIRTUAL_DEVICES=G makes the library show G devices (contexts of
    their own on the one physical GPU; exchanges through the host since RCCL refuses one device twice).  The worker runs the three
    shard modes, the replicated update and the sharded likelihood batch against the single-device entry points and the oracle:
    ragged M, the mean_Xs re-layout, owners on different devices, ties and NaN across shards, unfitted / misplaced replicas,
    fewer hyper-parameter sets than devices (sampling.jl:43-57, posterior.jl:31-79)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_virtual_worker.py")],
                       env=dict(os.environ, BOSS_VIRTUAL_DEVICES=str(G)), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and f"VIRTUAL-OK {G}" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("N0,steps", [(5, [1, 1, 3]), (100, [1]), (127, [2]), (128, [1]), (130, [1, 1, 1]), (255, [1, 130]),
                                      (300, [40, 100]), (256, [600]), (640, [1, 1]), (255, [33]), (250, [7, 300])])
def test_block_cholesky_append(api, O, N0, steps):
    """boss_gp_append == a fresh fit of the augmented data (augment_dataset! + model_posterior with
    unchanged hyper-parameters, problem.jl:191-198 / batch.jl:32-38): logpdf, factor, mean, variance."""
    rng = np.random.default_rng(N0 + 7 * len(steps))
    d, M = 3, 41
    Ntot = N0 + sum(steps)
    X = rng.uniform(0, 1, (d, Ntot))
    y = np.sin(3 * X).sum(0) + 0.05 * rng.standard_normal(Ntot)
    mean = 0.3 + 0.1 * X[0]
    Xs = rng.uniform(0, 1, (d, M))
    ms = 0.3 + 0.1 * Xs[0]
    lam, amp, sig = np.array([0.4, 0.5, 0.6]), 1.3, 0.05
    g = api.GP(X[:, :N0], y[:N0], "matern52")
    g.update(lam, amp, sig, mean[:N0])
    n_at = N0
    for n in steps:
        lp = g.append(X[:, n_at:n_at + n], y[n_at:n_at + n], mean[n_at:n_at + n])
        n_at += n
        post = O.gp_fit(X[:, :n_at], y[:n_at], "matern52", lam, amp, sig, mean=mean[:n_at])
        assert abs(lp - post.logpdf) <= 1e-10 * (1 + abs(post.logpdf)), (n_at, lp, post.logpdf)
        mu, var = g.predict(Xs, ms)
        mu_o, var_o = O.gp_mean_and_var(post, Xs, ms)
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9), n_at
    L, z = g.factor()
    assert L.shape == (Ntot, Ntot)
    import scipy.linalg as sla
    z_o = sla.solve_triangular(post.L, post.delta, lower=True)
    assert np.allclose(L, post.L, rtol=0, atol=1e-10) and np.allclose(z, z_o, rtol=0, atol=1e-9)
    # a later full update on the grown handle is still right
    lp2 = g.update(lam * 1.1, amp, sig, mean)
    post2 = O.gp_fit(X, y, "matern52", lam * 1.1, amp, sig, mean=mean)
    assert abs(lp2 - post2.logpdf) <= 1e-10 * (1 + abs(post2.logpdf))
    g.close()


@pytest.mark.parametrize("N0", [5, 20, 100])
def test_small_posterior_predict_then_append_then_predict(api, O, N0):
    """The reference's own regime (a BO loop or SequentialBatchAM at N ≈ 20): update → predict (single-launch kernels on
    block 0 of the block inverses) → append → predict / acquisition / gradients.  The block of L⁻¹ that the small kernels
    read must be rebuilt after every append (it used to be keyed on the update count only)."""
    rng = np.random.default_rng(900 + N0)
    d, M = 2, 37
    Ntot = N0 + 1 + 3
    X = rng.uniform(0, 1, (d, Ntot))
    y = np.sin(3 * X).sum(0) + 0.05 * rng.standard_normal(Ntot)
    Xs = rng.uniform(0, 1, (d, M))
    lam, amp, sig = np.array([0.4, 0.6]), 1.1, 0.05
    g = api.GP(X[:, :N0], y[:N0], "matern52")
    g.update(lam, amp, sig)
    cand = api.Candidates(Xs)
    n_at = N0
    for n in (1, 3):
        # a prediction and an acquisition call BEFORE the append mark block 0 as built for this update
        g.predict(Xs)
        api.acq_ei([[g]], cand, [1.0], None, float(y[:n_at].max()))
        g.append(X[:, n_at:n_at + n], y[n_at:n_at + n])
        n_at += n
        post = O.gp_fit(X[:, :n_at], y[:n_at], "matern52", lam, amp, sig)
        mu_o, var_o = O.gp_mean_and_var(post, Xs)
        mu, var = g.predict(Xs)
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9), (N0, n_at)
        best = float(y[:n_at].max())
        acq, am, _ = api.acq_ei([[g]], cand, [1.0], None, best)
        acq_o = O.ei_acquisition([post], Xs, [1.0], None, best)
        assert np.allclose(acq, acq_o, rtol=0, atol=1e-10) and am == int(np.argmax(acq_o)), (N0, n_at)
        mu2, var2, dmu, dvar = g.predict_grad(Xs)
        _, _, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs)
        assert np.allclose(mu2, mu_o, rtol=0, atol=1e-9) and np.allclose(var2, O.clip_var(var_o), rtol=0, atol=1e-9)
        assert np.allclose(dmu, dmu_o, rtol=0, atol=1e-8 * (1 + np.abs(dmu_o).max()))
        assert np.allclose(dvar, dvar_o, rtol=0, atol=1e-8 * (1 + np.abs(dvar_o).max()))
    g.close()


def test_append_zero_mean_discrete_and_errors(api, O):
    rng = np.random.default_rng(3)
    d, N0, n = 2, 140, 5
    X = np.round(rng.uniform(0, 6, (d, N0 + n)) * 2) / 2 + np.array([[0.0], [0.013]])
    y = np.cos(X).sum(0)
    disc = [True, False]
    g = api.GP(X[:, :N0], y[:N0], "matern32", disc)
    with pytest.raises(api.BossError):
        g.append(X[:, N0:], y[N0:])                      # not fitted yet
    g.update([1.5, 2.0], 1.0, 0.1)
    lp = g.append(X[:, N0], y[N0])                        # a single vector
    lp = g.append(X[:, N0 + 1:], y[N0 + 1:])
    post = O.gp_fit(X, y, "matern32", [1.5, 2.0], 1.0, 0.1, discrete=disc)
    assert abs(lp - post.logpdf) <= 1e-10 * (1 + abs(post.logpdf))
    Xs = rng.uniform(0, 6, (d, 17))
    mu, var = g.predict(Xs)
    mu_o, var_o = O.gp_mean_and_var(post, Xs)
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9)
    with pytest.raises(ValueError):
        g.append(np.zeros((3, 1)), [0.0])
    # a posterior with a prior mean needs the mean at the new points (it must not silently become 0 there)
    gm = api.GP(X[:, :N0], y[:N0], "matern32", disc)
    gm.update([1.5, 2.0], 1.0, 0.1, mean_X=0.3 * X[0, :N0])
    with pytest.raises(api.BossError) as e:
        gm.append(X[:, N0], y[N0])
    assert e.value.code == api.BOSS_E_INVALID and gm.N == N0
    lpm = gm.append(X[:, N0], y[N0], [0.3 * X[0, N0]])
    want = O.gp_fit(X[:, :N0 + 1], y[:N0 + 1], "matern32", [1.5, 2.0], 1.0, 0.1, mean=0.3 * X[0, :N0 + 1], discrete=disc).logpdf
    assert abs(lpm - want) <= 1e-10 * (1 + abs(want)) and gm.N == N0 + 1
    gm.close()
    # a duplicated point with (almost) no noise makes the augmented matrix singular -> PosDefException
    X2 = rng.uniform(0, 6, (d, N0))
    g2 = api.GP(X2, np.cos(X2).sum(0), "matern32")
    g2.update([1.0, 1.0], 1.0, 0.0)
    with pytest.raises(api.PosDefException):
        for _ in range(3):
            g2.append(X2[:, 0], np.cos(X2[:, 0]).sum())
    assert g2.N > N0
    g.close()
    g2.close()


def test_sequential_batch_am_matches_refit_loop(api, O):
    """HipSequentialBatchAM (resident posteriors + block Cholesky appends) selects the same batch as
    the reference's loop (batch.jl:26-38), which rebuilds the posterior from scratch each time —
    restated here with the oracle."""
    import boss_jl_amd as B
    rng = np.random.default_rng(17)
    d, N, M, P, nb = 2, 126, 300, 2, 5                 # the appends cross a 128-row block boundary
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), X[0] - X[1]])
    Xs = np.asfortranarray(rng.uniform(0, 1, (d, M)))
    y_max = np.array([np.inf, 0.3])
    coefs = [1.0, 0.0]
    prm = B.HipGPParams(np.array([[0.3, 0.5], [0.4, 0.6]]), [1.0, 0.8], [0.05, 0.02])
    model = B.HipGaussianProcess([None] * P, [None] * P, [None] * P, mean=lambda x: [0.1, -0.2])
    prob = B.BossProblem(None, B.Domain((np.zeros(d), np.ones(d))), B.ExpectedImprovement(B.LinFitness(coefs)), model,
                         B.ExperimentData(X, Y), y_max, prm)
    Xb, val = B.HipSequentialBatchAM(B.HipBatchAM(points=Xs), nb).maximize_acquisition(prob)
    assert val is None and Xb.shape == (d, nb)
    assert prob.data.X.shape == (d, N)                  # the caller's problem is untouched (deepcopy semantics)
    Xo, Yo = X.copy(), Y.copy()
    means = [0.1, -0.2]
    for it in range(nb):
        posts = [O.gp_fit(Xo, Yo[i], "matern52", prm.lengthscales[:, i], prm.amplitudes[i], prm.noise_std[i],
                          mean=np.full(Xo.shape[1], means[i])) for i in range(P)]
        ms = [np.full(M, means[i]) for i in range(P)]
        acq = O.ei_acquisition(posts, Xs, coefs, y_max, O.best_so_far(coefs, Yo, y_max), means_s=ms)
        j = int(np.argmax(acq))
        assert np.array_equal(Xb[:, it], Xs[:, j]), it
        yhat = np.array([O.gp_mean_and_var(posts[i], Xs[:, j:j + 1], ms[i][j:j + 1])[0][0] for i in range(P)])
        Xo = np.hstack([Xo, Xs[:, j:j + 1]])
        Yo = np.hstack([Yo, yhat[:, None]])


@pytest.mark.parametrize("d,N,M", [(17, 150, 40), (33, 257, 65), (8, 511, 33), (8, 513, 32)])
def test_wide_inputs_and_step_boundaries(api, O, d, N, M):
    """Input dimensions beyond one 16-wide staging chunk of the Gram kernel, and N just below / above
    the 256-row prediction steps (padding rows must behave as identity)."""
    rng = np.random.default_rng(d * 1000 + N)
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    Xs = rng.uniform(0, 1, (d, M))
    lam = rng.uniform(0.8, 1.6, d)
    g = api.GP(X, y, "matern32")
    lp = g.update(lam, 0.9, 0.07)
    post = O.gp_fit(X, y, "matern32", lam, 0.9, 0.07)
    assert abs(lp - post.logpdf) <= 1e-10 * (1 + abs(post.logpdf))
    mu, var = g.predict(Xs)
    mu_o, var_o = O.gp_mean_and_var(post, Xs)
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9)
    g.close()


def test_two_host_threads_share_a_device(api, O):
    """The reference calls its closures from Threads.@threads regions (optim_multistart.jl:61-90):
    two host threads drive two different handles on the same device at the same time."""
    import threading
    rng = np.random.default_rng(99)
    d, N, M = 4, 300, 96
    data = []
    for t in range(2):
        X = rng.uniform(0, 1, (d, N))
        y = np.cos(3 * X).sum(0)
        data.append((X, y, rng.uniform(0, 1, (d, M))))
    errs = []

    def work(t):
        try:
            X, y, Xs = data[t]
            g = api.GP(X, y, "matern52")
            cand = api.Candidates(Xs)
            for it in range(6):
                lam = np.full(d, 0.4 + 0.05 * it + 0.1 * t)
                lp = g.update(lam, 1.0, 0.05)
                post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
                assert abs(lp - post.logpdf) <= 1e-10 * (1 + abs(post.logpdf))
                acq, am, mx = api.acq_ei([[g]], cand, [1.0], None, float(y.max()))
                want = O.ei_acquisition([post], Xs, [1.0], None, float(y.max()))
                assert np.allclose(acq, want, rtol=0, atol=1e-12) and am == int(np.argmax(want))
                mu, var = g.predict(Xs)
                mu_o, var_o = O.gp_mean_and_var(post, Xs)
                assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9)
            g.close()
        except Exception as e:  # pragma: no cover
            import traceback
            errs.append(traceback.format_exc())

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[0]


@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
@pytest.mark.parametrize("d,N,M", [(1, 5, 3), (2, 1, 7), (2, 20, 224), (4, 64, 1), (5, 127, 300), (32, 128, 65), (2, 20, 5000), (3, 100, 3000), (3, 130, 45), (8, 700, 100),
                                   (17, 300, 33)])   # N <= 128 with d <= 32: the single-launch kernel
def test_posterior_gradients(api, O, kernel, d, N, M):
    """boss_gp_predict_grad: mean / variance and their gradients w.r.t. the candidates
    against the oracle's analytic restatement (itself pinned to finite differences on the CPU)."""
    rng = np.random.default_rng(31 * d + N)
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(3 * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    mean = 0.2 + 0.3 * X[0]
    Xs = rng.uniform(0, 1, (d, M))
    ms = 0.2 + 0.3 * Xs[0]
    mg = np.zeros((d, M))
    mg[0] = 0.3
    lam = rng.uniform(0.4, 0.9, d) * np.sqrt(d)
    g = api.GP(X, y, kernel)
    g.update(lam, 1.2, 0.05, mean)
    post = O.gp_fit(X, y, kernel, lam, 1.2, 0.05, mean=mean)
    mu, var, dmu, dvar = g.predict_grad(Xs, ms, mg)
    mu_o, var_o, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs, ms, mg)
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, O.clip_var(var_o), rtol=0, atol=1e-9)
    scale = 1.0 + np.abs(dmu_o).max()
    assert np.allclose(dmu, dmu_o, rtol=0, atol=1e-8 * scale), np.abs(dmu - dmu_o).max()
    assert np.allclose(dvar, dvar_o, rtol=0, atol=1e-8 * (1.0 + np.abs(dvar_o).max())), np.abs(dvar - dvar_o).max()
    # the plain prediction still works afterwards (the backward pass ran in place on the scratch slabs)
    mu2, var2 = g.predict(Xs, ms)
    assert np.allclose(mu2, mu, rtol=0, atol=1e-12) and np.allclose(var2, var, rtol=0, atol=1e-12)
    # a second call reuses the transposed factor; after an update it is rebuilt
    g.update(lam * 1.1, 1.0, 0.06, mean)
    post2 = O.gp_fit(X, y, kernel, lam * 1.1, 1.0, 0.06, mean=mean)
    _, _, dmu2, dvar2 = g.predict_grad(Xs, ms, mg)
    _, _, dmu2_o, dvar2_o = O.gp_mean_and_var_grad(post2, Xs, ms, mg)
    assert np.allclose(dmu2, dmu2_o, rtol=0, atol=1e-8 * (1.0 + np.abs(dmu2_o).max()))
    assert np.allclose(dvar2, dvar2_o, rtol=0, atol=1e-8 * (1.0 + np.abs(dvar2_o).max()))
    g.close()


@pytest.mark.parametrize("N", [200, 50])
def test_posterior_gradients_discrete_and_ei_chain(api, O, N):
    rng = np.random.default_rng(8)
    d, M = 3, 50
    X = rng.uniform(0, 5, (d, N))
    y = np.cos(X).sum(0)
    disc = [True, False, False]
    lam = np.array([1.5, 2.0, 1.0])
    g = api.GP(X, y, "matern52", disc)
    g.update(lam, 1.0, 0.1)
    post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.1, discrete=disc)
    Xs = rng.uniform(0, 5, (d, M))
    mu, var, dmu, dvar = g.predict_grad(Xs)
    mu_o, var_o, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs)
    assert np.all(dmu[0] == 0) and np.all(dvar[0] == 0)
    assert np.allclose(dmu, dmu_o, rtol=0, atol=1e-8) and np.allclose(dvar, dvar_o, rtol=0, atol=1e-8)
    # gradient of EI by the chain rule from the device moments == oracle
    b = float(y.max()) - 0.5
    ei, dei = O.expected_improvement_lin_grad([1.0], mu[None], var[None], dmu[None], dvar[None], b)
    ei_o, dei_o = O.expected_improvement_lin_grad([1.0], mu_o[None], var_o[None], dmu_o[None], dvar_o[None], b)
    assert np.allclose(ei, ei_o, rtol=0, atol=1e-10) and np.allclose(dei, dei_o, rtol=0, atol=1e-8)
    g.close()


@pytest.mark.parametrize("mode", ["both", "best_only", "cons_only", "none"])
def test_acquisition_gradient(api, O, mode):
    """boss_acq_ei_grad: EI × feasibility and its gradient w.r.t. the candidates, two outputs with prior
    means, all four construct_ei variants, make_safe mask."""
    rng = np.random.default_rng(77)
    d, N, M, P = 3, 260, 70, 2
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), X[0] - X[1] + 0.2 * np.cos(4 * X[2])])
    lam = [np.array([0.4, 0.6, 0.5]), np.array([0.5, 0.5, 0.7])]
    mean_f = [lambda Z: 0.1 + 0.2 * Z[0], lambda Z: -0.1 * Z[1]]
    mean_g = [np.array([0.2, 0.0, 0.0]), np.array([0.0, -0.1, 0.0])]
    gps, posts = [], []
    for p in range(P):
        g = api.GP(X, Y[p], "matern52")
        g.update(lam[p], 1.0 + 0.2 * p, 0.05, mean_f[p](X))
        gps.append(g)
        posts.append(O.gp_fit(X, Y[p], "matern52", lam[p], 1.0 + 0.2 * p, 0.05, mean=mean_f[p](X)))
    Xs = np.asfortranarray(rng.uniform(-0.05, 1.05, (d, M)))
    y_max = [np.inf, 0.3] if mode in ("both", "cons_only") else None
    coefs = [1.0, 0.2]
    b = O.best_so_far(coefs, Y, [np.inf, 0.3]) if mode in ("both", "best_only") else None
    mask = O.in_bounds(Xs, np.zeros(d), np.ones(d))
    ms = [mean_f[p](Xs) for p in range(P)]
    mgs = [np.repeat(mean_g[p][:, None], M, axis=1) for p in range(P)]
    acq, dacq = api.acq_ei_grad(gps, Xs, coefs, y_max, b, mask, ms, mgs)
    acq_o, dacq_o = O.ei_acquisition_grad(posts, Xs, coefs, y_max, b, valid_mask=mask, means_s=ms, mean_grads_s=mgs)
    assert np.allclose(acq, acq_o, rtol=0, atol=1e-11), np.abs(acq - acq_o).max()
    assert np.allclose(dacq, dacq_o, rtol=0, atol=1e-9 * (1.0 + np.abs(dacq_o).max())), np.abs(dacq - dacq_o).max()
    assert np.all(dacq[:, ~mask] == 0.0) and np.all(acq[~mask] == 0.0)
    # the value agrees with the non-gradient entry point
    cand = api.Candidates(Xs)
    acq2, _, _ = api.acq_ei([gps], cand, coefs, y_max, b, mask, np.stack(ms)[None])
    assert np.allclose(acq, acq2, rtol=0, atol=1e-13)
    for g in gps:
        g.close()


def test_gradient_maximizer_beats_its_starts_and_matches_a_dense_grid(api, O):
    """HipGradientAM (OptimizationAM semantics with device gradients): from 40 random starts it reaches
    the acquisition's global maximum found by a dense grid (2-D problem), stays inside the domain, and
    never returns less than the best start."""
    import boss_jl_amd as B
    rng = np.random.default_rng(4)
    d, N = 2, 60
    X = rng.uniform(0, 1, (d, N))
    Y = (np.sin(5 * X[0]) * np.cos(3 * X[1]) + 0.5 * X[0])[None, :]
    prm = B.HipGPParams(np.array([[0.25], [0.3]]), [1.0], [0.02])
    model = B.HipGaussianProcess([None], [None], [None])
    prob = B.BossProblem(None, B.Domain((np.zeros(d), np.ones(d))), B.ExpectedImprovement(B.LinFitness([1.0])), model,
                         B.ExperimentData(X, Y), None, prm)
    am = B.HipGradientAM(x_prior=lambda r: r.uniform(0, 1, d), multistart=40, iters=40, seed=2)
    x, val = am.maximize_acquisition(prob)
    assert x.shape == (d,) and np.all(x >= 0) and np.all(x <= 1)
    # dense grid through the oracle
    post = O.gp_fit(X, Y[0], "matern52", prm.lengthscales[:, 0], 1.0, 0.02)
    gx = np.linspace(0, 1, 201)
    G = np.asfortranarray(np.stack(np.meshgrid(gx, gx, indexing="ij")).reshape(2, -1))
    b = float(Y.max())
    ag = O.ei_acquisition([post], G, [1.0], [np.inf], b)
    assert val >= ag.max() * (1 - 1e-3) - 1e-12, (val, ag.max())
    a_x = O.ei_acquisition([post], x[:, None], [1.0], [np.inf], b)[0]
    assert abs(a_x - val) <= 1e-10
    starts_rng = np.random.default_rng(2)
    S0 = np.stack([starts_rng.uniform(0, 1, d) for _ in range(40)], axis=1)
    assert val >= O.ei_acquisition([post], S0, [1.0], [np.inf], b).max() - 1e-12


@pytest.mark.parametrize("N0,steps", [(100, [1, 1, 1]), (250, [3, 9, 1]), (500, [1] * 4)])
def test_tracked_candidates_follow_appends(api, O, N0, steps):
    """boss_track_*: after GP.append the resident moments are extended row by row and equal a fresh
    prediction on the augmented data; the tracked acquisition equals boss_acq_ei."""
    rng = np.random.default_rng(N0)
    d, M = 3, 77
    Ntot = N0 + sum(steps)
    X = rng.uniform(0, 1, (d, Ntot))
    y = np.sin(3 * X).sum(0) + 0.05 * rng.standard_normal(Ntot)
    mean = 0.2 - 0.1 * X[1]
    Xs = np.asfortranarray(rng.uniform(0, 1, (d, M)))
    ms = 0.2 - 0.1 * Xs[1]
    lam = np.array([0.4, 0.5, 0.6])
    g = api.GP(X[:, :N0], y[:N0], "matern52")
    g.update(lam, 1.1, 0.05, mean[:N0])
    cand = api.Candidates(Xs)
    tr = api.Track(g, cand, ms)
    n_at = N0
    for n in steps:
        g.append(X[:, n_at:n_at + n], y[n_at:n_at + n], mean[n_at:n_at + n])
        n_at += n
        post = O.gp_fit(X[:, :n_at], y[:n_at], "matern52", lam, 1.1, 0.05, mean=mean[:n_at])
        mu_o, var_o = O.gp_mean_and_var(post, Xs, ms, clip=False)
        mu, var = tr.moments()
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9), n_at
        b = float(y[:n_at].max())
        acq, am, mx = api.acq_ei_tracks([[tr]], [1.0], None, b)
        acq2, am2, mx2 = api.acq_ei([[g]], cand, [1.0], None, b, None, ms[None, None])
        assert np.allclose(acq, acq2, rtol=0, atol=1e-12) and am == am2
    # a re-fit with new hyper-parameters invalidates the track
    g.update(lam * 1.1, 1.1, 0.05, mean)
    with pytest.raises(api.BossError):
        tr.sync()
    tr.close()
    g.close()


@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
@pytest.mark.parametrize("d,N", [(1, 7), (2, 1), (2, 3), (3, 16), (8, 20), (5, 127), (4, 128), (32, 100), (3, 200), (8, 700), (17, 300),
                                 (4, 1300), (2, 2300)])   # N <= 128 with d <= 32: the single-launch kernel
def test_loglike_gradient(api, O, kernel, d, N):
    """boss_gp_loglike_grad: ∂logpdf/∂(λ, α, σ) against the oracle's analytic restatement
    (itself pinned to finite differences on the CPU)."""
    rng = np.random.default_rng(13 * d + N)
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(3 * X).sum(0) / np.sqrt(d) + 0.1 * rng.standard_normal(N)
    mean = 0.1 + 0.2 * X[0]
    lam = rng.uniform(0.4, 0.9, d) * np.sqrt(d)
    amp, sig = 1.2, 0.1
    g = api.GP(X, y, kernel)
    lp = g.update(lam, amp, sig, mean)
    lp2, grad = g.loglike_grad()
    ll_o, grad_o = O.gp_data_loglike_grad(X, y, kernel, lam, amp, sig, mean=mean)
    assert lp2 == lp and abs(lp - ll_o) <= 1e-10 * (1 + abs(ll_o))
    tol = 1e-8 * (1.0 + np.abs(grad_o).max())
    assert np.allclose(grad, grad_o, rtol=0, atol=tol), (grad, grad_o)
    # still consistent after another update (the scratch matrices are per device, not per handle)
    lp3 = g.update(lam * 1.2, amp * 0.9, sig * 1.5, mean)
    _, grad3 = g.loglike_grad()
    _, grad3_o = O.gp_data_loglike_grad(X, y, kernel, lam * 1.2, amp * 0.9, sig * 1.5, mean=mean)
    assert np.allclose(grad3, grad3_o, rtol=0, atol=1e-8 * (1.0 + np.abs(grad3_o).max()))
    g.close()


def test_gradient_map_fitter_improves_on_its_starts(api, O):
    """HipGradientMAP (OptimizationMAP semantics with device gradients): every start ends at a log-posterior
    no lower than where it began, Dirac-prior parameters stay fixed, and the result beats plain sampling with
    the same number of likelihood evaluations' worth of prior draws."""
    import boss_jl_amd as B
    rng = np.random.default_rng(6)
    d, N, P = 2, 80, 2
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(6 * X[0]) + 0.3 * X[1], np.cos(4 * X[1])]) + 0.05 * rng.standard_normal((P, N))
    model = B.HipGaussianProcess(lengthscale_priors=[B.MvLogNormal([-1., -1.], [1., 1.])] * P,
                                 amplitude_priors=[B.LogNormal(0., 1.)] * P,
                                 noise_std_priors=[B.LogNormal(-2., 1.), B.Dirac(0.05)])
    prob = B.BossProblem(None, B.Domain((np.zeros(d), np.ones(d))), B.ExpectedImprovement(B.LinFitness([1., 0.])), model,
                         B.ExperimentData(X, Y))
    fit = B.HipGradientMAP(multistart=4, iters=25, seed=3)
    allr = fit.estimate_parameters(prob, return_all=True)
    assert len(allr) == 4
    sampler, prior_ll = model.params_sampler(), model.params_loglike()
    srng = np.random.default_rng(3)
    starts = [sampler(srng) for _ in range(4)]

    def logpost(p):
        return O.data_loglike(X, Y, "matern52", p.lengthscales, p.amplitudes, p.noise_std) + prior_ll(p)

    for r, s0 in zip(allr, starts):
        assert r.params.noise_std[1] == 0.05                                  # Dirac parameter untouched
        lp = logpost(r.params)
        assert abs(lp - r.loglike) <= 1e-8 * (1 + abs(lp))                    # the reported value is the oracle's value
        assert r.loglike >= logpost(s0) - 1e-9
    best = fit.estimate_parameters(prob)
    assert best.loglike == max(r.loglike for r in allr)
    sampled = B.HipBatchedMAP(samples=200, seed=3).estimate_parameters(prob)
    assert best.loglike >= sampled.loglike - 1e-6
    # SampleOptMAP (sample_opt.jl:38-49): the ascents start from the best prior samples, so each result is at least as
    # good as the sample it started from and the winner at least as good as plain sampling
    so = B.HipSampleOptMAP(samples=200, multistart=3, iters=10, seed=3)
    scored = B.HipBatchedMAP(samples=200, seed=3).estimate_parameters(prob, return_all=True)
    top = sorted(scored, key=lambda r: -r.loglike)[:3]
    res = so.estimate_parameters(prob, return_all=True)
    assert len(res) == 3 and all(r.loglike >= t.loglike - 1e-9 for r, t in zip(res, top))
    assert so.estimate_parameters(prob).loglike >= sampled.loglike - 1e-9
    with pytest.raises(AssertionError):
        B.HipSampleOptMAP(samples=2, multistart=3)


@pytest.mark.parametrize("N,M", [(1024, 1), (1100, 5), (2048, 32), (4096, 1)])
def test_few_candidates_path(api, O, N, M):
    """M <= 32 at N >= 1024 takes the chip-wide step-by-step path (the reference evaluates ONE candidate
    per call): moments, covariance and gradients agree with the oracle like the fused kernel's."""
    rng = np.random.default_rng(N + M)
    d = 4
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(3 * X).sum(0) / 2 + 0.05 * rng.standard_normal(N)
    mean = 0.2 - 0.1 * X[1]
    Xs = rng.uniform(0, 1, (d, M))
    ms = 0.2 - 0.1 * Xs[1]
    lam = np.array([0.4, 0.5, 0.6, 0.7])
    g = api.GP(X, y, "matern52")
    g.update(lam, 1.1, 0.05, mean)
    post = O.gp_fit(X, y, "matern52", lam, 1.1, 0.05, mean=mean)
    mu, var = g.predict(Xs, ms)
    mu_o, var_o = O.gp_mean_and_var(post, Xs, ms)
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9)
    if M == 1:                                               # the vector form of the plugin API
        import boss_jl_amd as B
        muv, varv = g.predict(Xs[:, 0], ms)
        assert muv.shape == (1,) and abs(muv[0] - mu_o[0]) <= 1e-9
    _, cov = g.predict_cov(Xs, ms)
    _, cov_o = O.gp_mean_and_cov(post, Xs, ms)
    assert np.allclose(cov, cov_o, rtol=0, atol=1e-9)
    _, _, dmu, dvar = g.predict_grad(Xs, ms)
    _, _, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs, ms)
    assert np.allclose(dmu, dmu_o, rtol=0, atol=1e-8 * (1 + np.abs(dmu_o).max()))
    assert np.allclose(dvar, dvar_o, rtol=0, atol=1e-8 * (1 + np.abs(dvar_o).max()))
    g.close()


@pytest.mark.parametrize("N,M", [(1024, 40), (1536, 300), (2300, 1000), (4096, 1024), (5000, 70)])
def test_first_call_on_a_factorisation_steps_with_pair_updates(api, O, N, M):
    """The first prediction on a factorisation with 33 .. 4096 candidates: one launch per 256-row step (V_i from the block
    inverse and the precomputed W_i, W'_i), the residual blocks updated in pairs on the even steps.  Odd and even block counts,
    ragged last tiles, an update in between (W must follow the new factor), against the oracle."""
    rng = np.random.default_rng(3 * N + M)
    d = 5
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(3 * X).sum(0) / 2 + 0.05 * rng.standard_normal(N)
    mean = 0.2 - 0.1 * X[1]
    Xs = rng.uniform(0, 1, (d, M))
    ms = 0.2 - 0.1 * Xs[1]
    g = api.GP(X, y, "matern52")
    for lam0, noise in ((0.45, 0.05), (0.6, 0.08)):
        lam = np.full(d, lam0)
        g.update(lam, 1.2, noise, mean)
        post = O.gp_fit(X, y, "matern52", lam, 1.2, noise, mean=mean)
        mu, var = g.predict(Xs, ms)                          # first call on this factorisation
        mu_o, var_o = O.gp_mean_and_var(post, Xs, ms)
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9)
    g.close()


def test_first_call_steps_agree_with_the_two_launch_fallback():
    """BOSS_NO_FEW_W=1 keeps the round-2 form of the few-candidates steps (finish kernel, then update kernel: what the library
    falls back to when it cannot allocate W): same moments as the one-launch-per-step form up to summation order."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from boss_jl_amd import api
rng = np.random.default_rng(5)
d, N, M = 6, 3000, 700
X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0) + 0.05*rng.standard_normal(N)
Xs = rng.uniform(0, 1, (d, M)); lam = np.full(d, 0.5)
g = api.GP(X, y, "matern52")
g.update(lam, 1.0, 0.05)
mu, var = g.predict(Xs)
print("RES", " ".join(repr(float(v)) for v in np.concatenate([mu[:50], var[:50], [mu.sum(), var.sum()]])))
''' % ROOT
    out = {}
    for tag, extra in (("steps", {}), ("fallback", {"BOSS_NO_FEW_W": "1"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "RES" in r.stdout, tag + ": " + r.stdout + r.stderr
        out[tag] = np.array([float(v) for v in r.stdout.split("RES")[1].split()])
    assert np.allclose(out["steps"], out["fallback"], rtol=0, atol=1e-10 * (1 + np.abs(out["fallback"]).max()))


@pytest.mark.parametrize("N,M", [(1024, 1), (1300, 3), (2048, 4), (4096, 1), (1024, 20), (1300, 33), (2048, 200), (6000, 1), (6000, 2)])
def test_repeated_few_candidate_calls_use_the_explicit_inverse(api, O, N, M):
    """One to four candidates per call, many calls per posterior (the reference's `acq.(eachcol(xs))` pattern): from
    the second call on a factorisation the handle predicts through the explicit L⁻ᵀ.  Every call agrees with the
    oracle, and an update / append in between switches back to the new factor."""
    rng = np.random.default_rng(N + 7 * M)
    d = 3
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(3 * X).sum(0) / 2 + 0.05 * rng.standard_normal(N)
    mean = 0.1 * X[0]
    lam = np.array([0.4, 0.5, 0.6])
    g = api.GP(X, y, "matern52")
    g.update(lam, 1.1, 0.05, mean)
    post = O.gp_fit(X, y, "matern52", lam, 1.1, 0.05, mean=mean)
    for call in range(4):
        Xs = rng.uniform(0, 1, (d, M))
        mu, var = g.predict(Xs, 0.1 * Xs[0])
        mu_o, var_o = O.gp_mean_and_var(post, Xs, 0.1 * Xs[0])
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9), call
    # gradients on the resident inverses (forward and adjoint passes as GEMMs), repeatedly
    for call in range(3):
        Xs = rng.uniform(0, 1, (d, M))
        mu, var, dmu, dvar = g.predict_grad(Xs, 0.1 * Xs[0], np.vstack([np.full(M, 0.1), np.zeros((d - 1, M))]))
        mu_o, var_o, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs, 0.1 * Xs[0], np.vstack([np.full(M, 0.1), np.zeros((d - 1, M))]))
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, O.clip_var(var_o), rtol=0, atol=1e-9), call
        assert np.allclose(dmu, dmu_o, rtol=0, atol=1e-8 * (1 + np.abs(dmu_o).max())), call
        assert np.allclose(dvar, dvar_o, rtol=0, atol=1e-8 * (1 + np.abs(dvar_o).max())), call
    for call in range(2):                                    # covariances need V itself: never the one-pass kernel
        Xs = rng.uniform(0, 1, (d, M))
        _, cov = g.predict_cov(Xs, 0.1 * Xs[0])
        _, cov_o = O.gp_mean_and_cov(post, Xs, 0.1 * Xs[0])
        assert np.allclose(cov, cov_o, rtol=0, atol=1e-9), call
    # a larger batch in between takes the other kernels and leaves the inverse usable
    Xb = rng.uniform(0, 1, (d, 40))
    mu_b, var_b = g.predict(Xb, 0.1 * Xb[0])
    mu_bo, var_bo = O.gp_mean_and_var(post, Xb, 0.1 * Xb[0])
    assert np.allclose(mu_b, mu_bo, rtol=0, atol=1e-9) and np.allclose(var_b, var_bo, rtol=0, atol=1e-9)
    # new hyper-parameters: the cached inverse must not survive
    g.update(lam * 1.3, 0.9, 0.07, mean)
    post2 = O.gp_fit(X, y, "matern52", lam * 1.3, 0.9, 0.07, mean=mean)
    for call in range(3):
        Xs = rng.uniform(0, 1, (d, M))
        mu, var = g.predict(Xs, 0.1 * Xs[0])
        mu_o, var_o = O.gp_mean_and_var(post2, Xs, 0.1 * Xs[0])
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9), call
    # an appended observation likewise
    xn = rng.uniform(0, 1, (d, 1))
    g.append(xn, [0.3], 0.1 * xn[0])
    post3 = O.gp_fit(np.hstack([X, xn]), np.append(y, 0.3), "matern52", lam * 1.3, 0.9, 0.07, mean=np.append(mean, 0.1 * xn[0]))
    for call in range(3):
        Xs = rng.uniform(0, 1, (d, M))
        mu, var = g.predict(Xs, 0.1 * Xs[0])
        mu_o, var_o = O.gp_mean_and_var(post3, Xs, 0.1 * Xs[0])
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9), call
    g.close()


def test_one_launch_small_calls_interleaved_on_two_handles():
    """One to four candidates and rank-one appends on resident inverse factors run as ONE launch whose last workgroup reduces the
    partials and answers through mapped host memory (csrc/small_calls.hpp); the finished-workgroup counter and the partial buffer
    belong to the device context, i.e. two handles share them.  200 interleaved calls on two posteriors of different sizes give,
    bit for bit, what the three-launch path gives (BOSS_FEW_FUSED=0: same sums in the same order) — and both agree with the oracle."""
    import subprocess
    import sys
    entry.build()
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from boss_jl_amd import api
from oracle import gp_oracle as O
api.load_library()
rng = np.random.default_rng(77); d = 4
hs, data = [], []
for N in (700, 1500):
    X = rng.uniform(0, 1, (d, N)); y = np.cos(3 * X).sum(0) / 2 + 0.05 * rng.standard_normal(N)
    g = api.GP(X, y, "matern52"); g.reserve(N + 40); g.update(np.full(d, 0.45), 1.2, 0.06)
    g.predict(rng.uniform(0, 1, (d, 1))); g.predict(rng.uniform(0, 1, (d, 1)))      # the second call builds the inverse factors
    hs.append(g); data.append([X, y])
out = []
for it in range(200):
    h = it %% 2; g = hs[h]
    if it %% 10 == 9:
        xn = rng.uniform(0, 1, (d, 1)); yn = float(rng.standard_normal() * 0.1)
        out.append(g.append(xn, [yn]))
        data[h][0] = np.hstack([data[h][0], xn]); data[h][1] = np.append(data[h][1], yn)
    else:
        M = 1 + it %% 4
        Xs = rng.uniform(0, 1, (d, M))
        mu, var = g.predict(Xs)
        out += list(mu) + list(var)
        if it %% 37 == 0:
            post = O.gp_fit(data[h][0], data[h][1], "matern52", np.full(d, 0.45), 1.2, 0.06)
            mo, vo = O.gp_mean_and_var(post, Xs)
            assert np.allclose(mu, mo, rtol=0, atol=1e-9) and np.allclose(var, vo, rtol=0, atol=1e-9), it
print("RES", " ".join(float(v).hex() for v in out))
""" % ROOT
    res = {}
    for tag, extra in (("one launch", {}), ("three launches", {"BOSS_FEW_FUSED": "0"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "RES" in r.stdout, tag + ": " + r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = r.stdout.split("RES")[1].split()
    assert len(res["one launch"]) > 400 and res["one launch"] == res["three launches"]


def test_repeated_single_candidate_calls_on_gradient_and_nonstationary_posteriors(api, O):
    d, n = 3, 300                                            # 1200 augmented rows
    X, y, dY = make_grad(d, n)
    lam = np.full(d, 0.45)
    gg = api.GradGP(X, y, dY, "matern52")
    gg.update(lam, 1.2, 0.05, 0.1)
    pg = O.gradient_gp_fit(X, y, dY, "matern52", lam, 1.2, 0.05, 0.1)
    rng = np.random.default_rng(2)
    for call in range(4):
        xs = rng.uniform(0, 1, (d, 1)) if call else X[:, 5:6].copy()
        mu, var = gg.predict(xs)
        mu_o, var_o = O.gradient_gp_mean_and_var(pg, xs)
        assert abs(mu[0] - mu_o[0]) <= 1e-9 and abs(var[0] - var_o[0]) <= 1e-9, call
    gg.close()
    N = 1100
    Xn, yn, _ = make(d, N, 1, seed=9)
    f_lam, f_amp, f_noise = latent(d)
    ev = lambda f, Z: np.array([f(Z[:, j]) for j in range(Z.shape[1])])                   # noqa: E731
    gn = api.GibbsGP(Xn, yn)
    gn.update(ev(f_lam, Xn).T, ev(f_amp, Xn), ev(f_noise, Xn))
    pn = O.nonstationary_fit(Xn, yn, ev(f_lam, Xn).T, ev(f_amp, Xn), ev(f_noise, Xn))
    for call in range(4):
        xs = rng.uniform(0, 1, (d, 2))
        mu, var = gn.predict(xs, ev(f_lam, xs).T, ev(f_amp, xs))
        mu_o, var_o = O.nonstationary_mean_and_var(pn, xs, ev(f_lam, xs).T, ev(f_amp, xs))
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9), call
    gn.close()


def test_first_call_steps_on_gradient_and_nonstationary_posteriors(api, O):
    """The one-launch-per-step form of the few-candidates substitution with the right-hand sides the other two model families
    write (aug_kstar_kernel, gibbs_kstar kernels): a few hundred candidates, enough row blocks for the pair updates."""
    d, n, M = 3, 420, 150                                    # 1680 augmented rows
    X, y, dY = make_grad(d, n)
    lam = np.full(d, 0.45)
    rng = np.random.default_rng(8)
    Xs = rng.uniform(0, 1, (d, M))
    gg = api.GradGP(X, y, dY, "matern52")
    gg.update(lam, 1.2, 0.05, 0.1)
    pg = O.gradient_gp_fit(X, y, dY, "matern52", lam, 1.2, 0.05, 0.1)
    mu, var = gg.predict(Xs)
    mu_o, var_o = O.gradient_gp_mean_and_var(pg, Xs)
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-8) and np.allclose(var, var_o, rtol=0, atol=1e-8)
    gg.close()
    N, M = 1500, 300
    Xn, yn, Xc = make(d, N, M, seed=10)
    f_lam, f_amp, f_noise = latent(d)
    ev = lambda f, Z: np.array([f(Z[:, j]) for j in range(Z.shape[1])])                   # noqa: E731
    gn = api.GibbsGP(Xn, yn)
    gn.update(ev(f_lam, Xn).T, ev(f_amp, Xn), ev(f_noise, Xn))
    pn = O.nonstationary_fit(Xn, yn, ev(f_lam, Xn).T, ev(f_amp, Xn), ev(f_noise, Xn))
    mu, var = gn.predict(Xc, ev(f_lam, Xc).T, ev(f_amp, Xc))
    mu_o, var_o = O.nonstationary_mean_and_var(pn, Xc, ev(f_lam, Xc).T, ev(f_amp, Xc))
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-8) and np.allclose(var, var_o, rtol=0, atol=1e-8)
    gn.close()


def test_caller_stream(api, O):
    """boss_set_stream: run the library on torch's current (non-default) stream; torch events then see the work."""
    import torch
    X, y, Xs = make(4, 300, 64, seed=2)
    lam = np.full(4, 0.5)
    want = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
    st = torch.cuda.Stream()
    try:
        with torch.cuda.stream(st):
            api.set_stream(0, st.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g = api.GP(X, y, "matern52")
            e0.record(st)
            lp = g.update(lam, 1.0, 0.05)
            mu, var = g.predict(Xs)
            e1.record(st)
            e1.synchronize()
            assert e0.elapsed_time(e1) > 0.0
    finally:
        api.set_stream(0, None)
    mu_o, var_o = O.gp_mean_and_var(want, Xs)
    assert abs(lp - want.logpdf) <= 1e-9 * (1 + abs(want.logpdf))
    assert np.allclose(mu, mu_o, atol=1e-9) and np.allclose(var, var_o, atol=1e-9)
    g.close()


def test_64_candidate_instantiation():
    """The 64-candidates-per-workgroup prediction kernel normally needs M >= 16384; BOSS_FORCE_BN64=1
    selects it for any M (read once per process, hence the subprocess).  Covers GEMM1 + GEMM2 + covariance."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from boss_jl_amd import api
from oracle import gp_oracle as O
rng = np.random.default_rng(3)
d, N, M = 3, 700, 150
X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0) + 0.05*rng.standard_normal(N)
Xs = rng.uniform(0, 1, (d, M)); lam = np.full(d, 0.5)
post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
mu_o, S_o = O.gp_mean_and_cov(post, Xs)
g = api.fit(X, y, "matern52", lam, 1.0, 0.05)
mu, var = g.predict(Xs)
mu2, S = g.predict_cov(Xs)
assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, np.diag(S_o), rtol=0, atol=1e-9)
assert np.allclose(S, S_o, rtol=0, atol=1e-9)
print("ok64")
''' % ROOT
    env = dict(os.environ, BOSS_FORCE_BN64="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok64" in r.stdout, r.stdout + r.stderr


def test_update_with_serialised_kernel_launches():
    """The look-ahead factorisation releases kernels of its side stream through gate kernels that poll a word stored by a
    LATER kernel of the main stream.  Where kernels execute one at a time in submission order (HIP_LAUNCH_BLOCKING, counter
    collection under rocprofv3, a debugger) a gate enqueued before its producer would spin into its timeout and fail the
    update; the gates are therefore enqueued behind their producers.  Same numbers as the concurrent run, update -> predict
    -> update so that the block-inverse gate is exercised too."""
    import subprocess
    import sys
    code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from boss_jl_amd import api
rng = np.random.default_rng(11)
d, N, M = 4, 4096, 64
X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0) + 0.05*rng.standard_normal(N)
Xs = rng.uniform(0, 1, (d, M)); lam = np.full(d, 0.4)
g = api.GP(X, y, "matern52")
t = time.perf_counter()
lp1 = g.update(lam, 1.0, 0.05)
mu, var = g.predict(Xs)
lp2 = g.update(lam, 1.0, 0.06)
mu2, var2 = g.predict(Xs)
print("RES", repr(lp1), repr(lp2), repr(float(mu2.sum())), repr(float(var2.sum())), time.perf_counter() - t)
''' % ROOT
    out = {}
    # third run: the environment of a counter-collection pass (rocprofv3 --pmc serialises ALL queues, so that even a gate enqueued
    # behind its producer would hold the device): the library must order its streams with events there
    for tag, extra in (("concurrent", {}), ("serialised", {"HIP_LAUNCH_BLOCKING": "1"}), ("events", {"ROCPROF_COUNTER_COLLECTION": "1"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "RES" in r.stdout, tag + ": " + r.stdout + r.stderr
        out[tag] = [float(v) for v in r.stdout.split("RES")[1].split()]
    a, b, e = out["concurrent"], out["serialised"], out["events"]
    assert b[:4] == e[:4], (b, e)                       # same kernels, same order of arithmetic: bit-identical
    # the concurrent run factors its diagonal blocks in the resident chain kernel (chain.hpp), which cannot run where launches
    # execute one at a time (the library leaves it out there): same numbers up to the order of the floating-point additions
    assert all(abs(x - y) <= 1e-10 * (1 + abs(y)) for x, y in zip(a[:4], b[:4])), (a, b)
    assert b[4] < 20.0 and e[4] < 20.0, (b, e)          # no gate ran into its timeout


def test_resident_chain_falls_back_when_it_cannot_run():
    """The resident chain kernel and its followers wait for each other across streams.  Where something keeps the chain kernel
    from running (here: BOSS_TEST_DROP_CHAIN=1 — it is never launched) the followers give up after their bounded wait, the
    update is marked (info = INT_MIN), the chain is switched off for the device and the update repeated under the gate schedule:
    the caller sees a correct result, once, a few tens of milliseconds late (the waits' budget follows from the size: ≈ 20× the
    expected update time, at least 20 ms)."""
    import subprocess
    import sys
    code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from boss_jl_amd import api
from oracle import gp_oracle as O
rng = np.random.default_rng(5)
d, N = 3, 1500
X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0) + 0.05*rng.standard_normal(N)
lam = np.full(d, 0.4)
g = api.GP(X, y, "matern52")
t = time.perf_counter(); lp1 = g.update(lam, 1.0, 0.05); t1 = time.perf_counter() - t
t = time.perf_counter(); lp2 = g.update(lam, 1.0, 0.06); t2 = time.perf_counter() - t
w1 = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05).logpdf; w2 = O.gp_fit(X, y, "matern52", lam, 1.0, 0.06).logpdf
print("RES", abs(lp1 - w1) / (1 + abs(w1)), abs(lp2 - w2) / (1 + abs(w2)), t1, t2)
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BOSS_TEST_DROP_CHAIN="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RES" in r.stdout, r.stdout + r.stderr
    e1, e2, t1, t2 = [float(v) for v in r.stdout.split("RES")[1].split()]
    assert e1 <= 1e-9 and e2 <= 1e-9, (e1, e2)
    assert t1 < 2.0 and t2 < 0.5, (t1, t2)              # the first update pays the bounded waits (20 ms budget at this size), the second runs without the chain
    assert "resident panel chain is switched off" in r.stderr, r.stderr   # reported once, not only under BOSS_CHAIN_VERBOSE


# ------------------------------------------------------------------------------------------
# GradientGaussianProcess — gradient observations, the n(1+d) augmented system
# ------------------------------------------------------------------------------------------
def make_grad(d, n, seed=3):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (d, n))
    w = np.linspace(1.0, 2.0, d)[:, None]
    y = np.sin(2 * np.pi * w * X).sum(0) / np.sqrt(d)
    dY = 2 * np.pi * w * np.cos(2 * np.pi * w * X) / np.sqrt(d)
    return X, y, dY


@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
@pytest.mark.parametrize("d,n,M,dup", [(1, 1, 3, False), (3, 40, 70, False), (2, 30, 10, True), (8, 150, 40, False), (16, 70, 33, True)])
def test_gradient_gp_parity(api, O, kernel, d, n, M, dup):
    """Augmented Gram + factor + logpdf + prediction against the oracle: N_aug below 1024 runs the fused
    kernel, above it the few-candidates path; `dup` repeats a training point and puts a candidate on one
    (the entries the reference evaluates at x_j + 1e-8, gradient_gp.jl:148-152,233)."""
    X, y, dY = make_grad(d, n)
    if dup and n > 1:
        X[:, 1] = X[:, 0]
    Xs = np.asfortranarray(np.random.default_rng(5).uniform(0, 1, (d, M)))
    Xs[:, 0] = X[:, min(2, n - 1)]
    lam = np.linspace(0.35, 0.6, d)
    post = O.gradient_gp_fit(X, y, dY, kernel, lam, 1.2, 0.05, 0.1)
    mu_o, var_o = O.gradient_gp_mean_and_var(post, Xs)
    K = post.L @ post.L.T
    tol = max(1e-9, np.linalg.cond(K) * K.shape[0] * 2.0 ** -53 * 8)
    g = api.GradGP(X, y, dY, kernel)
    assert g.N == n * (1 + d)
    lp = g.update(lam, 1.2, 0.05, 0.1)
    L, z = g.factor()
    assert abs(lp - post.logpdf) <= tol * (1 + abs(post.logpdf))
    assert np.abs(L - post.L).max() <= tol * np.abs(post.L).max()
    mu, var = g.predict(Xs)
    assert np.abs(mu - mu_o).max() <= tol * (1 + np.abs(mu_o).max())
    assert np.abs(var - var_o).max() <= tol * 1.2 ** 2 and var.min() >= 0.0
    mu1, var1 = g.predict(Xs[:, :1])                          # the reference's one-candidate call pattern
    assert abs(mu1[0] - mu_o[0]) <= tol * (1 + abs(mu_o[0])) and abs(var1[0] - var_o[0]) <= tol * 1.2 ** 2
    # a second update on the resident data (what data_loglike does under the fitter)
    lp2 = g.update(lam * 1.3, 0.9, 0.02, 0.05)
    assert abs(lp2 - O.gradient_gp_fit(X, y, dY, kernel, lam * 1.3, 0.9, 0.02, 0.05).logpdf) <= tol * 10 * (1 + abs(lp2))
    g.close()


def test_gradient_gp_wide_candidate_tiles(api, O):
    """The 64-candidates-per-workgroup kernel (M >= 16384) on a gradient posterior equals the 32-wide one."""
    d, n = 4, 60
    X, y, dY = make_grad(d, n)
    Xs = np.asfortranarray(np.random.default_rng(8).uniform(0, 1, (d, 16384 + 37)))
    g = api.GradGP(X, y, dY, "matern52")
    g.update(np.full(d, 0.5), 1.0, 0.05, 0.1)
    mu, var = g.predict(Xs)
    mu_s, var_s = g.predict(Xs[:, 16000:])
    assert np.allclose(mu[16000:], mu_s, rtol=0, atol=1e-12) and np.allclose(var[16000:], var_s, rtol=0, atol=1e-12)
    post = O.gradient_gp_fit(X, y, dY, "matern52", np.full(d, 0.5), 1.0, 0.05, 0.1)
    mu_o, var_o = O.gradient_gp_mean_and_var(post, Xs[:, :50])
    assert np.allclose(mu[:50], mu_o, rtol=0, atol=1e-10) and np.allclose(var[:50], var_o, rtol=0, atol=1e-10)
    g.close()


def test_gradient_gp_errors_and_unsupported_entry_points(api, O):
    d, n = 2, 12
    X, y, dY = make_grad(d, n)
    g = api.GradGP(X, y, dY, "matern52")
    with pytest.raises(api.BossError):
        g.predict(X)                                          # not fitted yet
    with pytest.raises(api.BossError):
        g.update([0.5, -0.1], 1.0, 0.1, 0.1)
    with pytest.raises(api.BossError):
        g.update([0.5, 0.5], 1.0, 0.1, -1.0)
    g.update([0.5, 0.5], 1.0, 0.1, 0.1)
    for call in (lambda: api.GP.update(g, [0.5, 0.5], 1.0, 0.1), lambda: g.set_y(y), lambda: api.GP.append(g, X[:, :1], y[:1]),
                 lambda: g.reserve(100), lambda: g.predict_grad(X, np.zeros(n)), lambda: g.predict_cov(X), lambda: api.GP.loglike_grad(g),
                 lambda: g.predict(X, np.zeros(n)), lambda: api.Track(g, api.Candidates(X))):
        with pytest.raises(api.BossError):
            call()
    mu, var = g.predict(X)                                    # the handle survives the refused calls
    assert np.isfinite(mu).all() and (var >= 0).all()
    with pytest.raises(api.BossError):
        api.GradGP(np.zeros((17, 3)), np.zeros(3), np.zeros((17, 3)))       # x_dim above the supported 16
    # coincident points without noise: the augmented matrix is singular -> PosDefException, logpdf -Inf
    Xd = X.copy()
    Xd[:, 1] = Xd[:, 2] = Xd[:, 0]
    gd = api.GradGP(Xd, y, dY, "sqexp")
    with pytest.raises(api.PosDefException):
        gd.update([0.5, 0.5], 1.0, 0.0, 0.0)
    gd.close()
    g.close()


@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
@pytest.mark.parametrize("d,n,M", [(1, 5, 3), (3, 40, 70), (8, 150, 40), (16, 70, 33), (4, 300, 1100)])
def test_gradient_gp_candidate_gradients(api, O, kernel, d, n, M):
    """SURVEY §8f3 over §8f4: ∇μ, ∇σ² and ∇acq of a gradient-observation posterior — what ForwardDiff pushes through
    mean_and_var (src/models/gradient_gp.jl:334-361) inside OptimizationAM (src/acquisition_maximizers/optimization.jl:36,89-118).
    Against the oracle's analytic gradients (themselves checked against finite differences, tests/test_oracle_crosscheck.py)."""
    X, y, dY = make_grad(d, n, seed=d + n)
    lam = np.linspace(0.35, 0.6, d)
    post = O.gradient_gp_fit(X, y, dY, kernel, lam, 1.1, 1e-2, 3e-2)
    Xs = np.random.default_rng(M).uniform(0.05, 0.95, (d, M))
    mu_o, var_o, dmu_o, dvar_o = O.gradient_gp_mean_and_var_grad(post, Xs)
    g = api.GradGP(X, y, dY, kernel)
    g.update(lam, 1.1, 1e-2, 3e-2)
    mu, var, dmu, dvar = g.predict_grad(Xs)
    K = O.augmented_kernel_matrix(kernel, X, lam, 1.1, 1e-2, 3e-2)
    tol = max(1e-9, np.linalg.cond(K) * K.shape[0] * 2.0 ** -53 * 8)
    sc = 1 + np.abs(dmu_o).max()
    assert np.all(np.abs(mu - mu_o) <= tol * (1 + np.abs(mu_o)))
    assert np.all(np.abs(var - np.maximum(var_o, 0.0)) <= tol * 1.1 ** 2)
    assert np.abs(dmu - dmu_o).max() <= tol * sc * 10, np.abs(dmu - dmu_o).max()
    assert np.abs(dvar - dvar_o).max() <= tol * (1 + np.abs(dvar_o).max()) * 10, np.abs(dvar - dvar_o).max()
    # the acquisition's chain rule on the device from those moments (construct_ei, expected_improvement.jl:74-76)
    best = float(y.max()) - 0.3
    acq, dacq = api.acq_ei_grad([g], Xs, [1.0], None, best)
    vo = np.maximum(var_o, 0.0)
    acq_o, dacq_o = O.expected_improvement_lin_grad([1.0], mu_o[None], vo[None], dmu_o[None], np.where(vo > 0, dvar_o, 0.0)[None], best)
    assert np.all(np.abs(acq - acq_o) <= tol * 10)
    assert np.abs(dacq - dacq_o).max() <= tol * (1 + np.abs(dacq_o).max()) * 100, np.abs(dacq - dacq_o).max()
    # repeated (the second few-candidates call on a factorisation runs on the resident inverse factors: another summation order);
    # the plain prediction still agrees
    mu2, var2, dmu2, dvar2 = g.predict_grad(Xs)
    assert np.abs(dmu2 - dmu).max() <= tol * sc * 10 and np.abs(dvar2 - dvar).max() <= tol * (1 + np.abs(dvar_o).max()) * 10
    mu3, var3 = g.predict(Xs)
    assert np.all(np.abs(mu3 - mu) <= 1e-9 * (1 + np.abs(mu)))
    g.close()


@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
@pytest.mark.parametrize("d,n,dup", [(1, 6, False), (3, 40, False), (2, 30, True), (8, 150, False), (16, 40, True)])
def test_gradient_gp_likelihood_gradient(api, O, kernel, d, n, dup):
    """SURVEY §8f3 over §8f4: ∂ℓ/∂(λ, α, σ, σ_∂) of the gradient-observation model's data_loglike (gradient_gp.jl:367-397) — what
    ForwardDiff yields inside OptimizationMAP (src/model_fitters/optimization.jl:146-164) — against the oracle's analytic gradient
    (itself checked against finite differences of the likelihood, tests/test_oracle_crosscheck.py); `dup` repeats a training point
    (derivative blocks evaluated at x_j + 1e-8, gradient_gp.jl:148-152)."""
    X, y, dY = make_grad(d, n, seed=3 * d + n)
    if dup:
        X[:, 1] = X[:, 0]
    lam = np.linspace(0.35, 0.6, d)
    hyp = (1.15, 0.04, 0.09)
    ll_o, gr_o = O.gradient_gp_loglike_grad(X, y, dY, kernel, lam, *hyp)
    K = O.augmented_kernel_matrix(kernel, X, lam, *hyp)
    tol = max(1e-9, np.linalg.cond(K) * K.shape[0] * 2.0 ** -53 * 8)
    g = api.GradGP(X, y, dY, kernel)
    g.update(lam, *hyp)
    ll, gr = g.loglike_grad()
    assert gr.shape == (d + 3,)
    assert abs(ll - ll_o) <= tol * (1 + abs(ll_o))
    assert np.abs(gr - gr_o).max() <= tol * (1 + np.abs(gr_o).max()) * 100, (np.abs(gr - gr_o).max(), np.abs(gr_o).max())
    ll2, gr2 = g.loglike_grad()                               # repeated on the same factorisation: bit-identical
    assert ll2 == ll and np.array_equal(gr, gr2)
    mu, var = g.predict(X[:, :3])                             # the handle's block inverses stay usable
    post = O.gradient_gp_fit(X, y, dY, kernel, lam, *hyp)
    mu_o, var_o = O.gradient_gp_mean_and_var(post, X[:, :3])
    assert np.abs(mu - mu_o).max() <= tol * (1 + np.abs(mu_o).max()) * 10
    g.close()


def test_gradient_gp_append_and_gradient_map(api, O):
    """augment_dataset! (src/types/problem.jl:191-198) on a gradient-observation handle = a fresh fit on all points at the same
    hyper-parameters; HipGradientMAP (OptimizationMAP semantics) on a HipGradientGaussianProcess climbs its log-posterior and ends
    where the likelihood gradient has (nearly) vanished along the free directions."""
    import boss_jl_amd as B
    from boss_jl_amd.gradient_gp import GradientData, HipGradientGaussianProcess, HipGradientGPParams
    d, n = 3, 40
    X, y, dY = make_grad(d, n + 7, seed=11)
    lam = np.array([0.4, 0.5, 0.6])
    g = api.GradGP(X[:, :n], y[:n], dY[:, :n], "matern52")
    with pytest.raises(api.BossError):
        g.append(X[:, n:], y[n:], dY[:, n:])                  # not fitted: there are no hyper-parameters to re-use
    g.update(lam, 1.1, 0.03, 0.07)
    lp = g.append(X[:, n:n + 1], y[n:n + 1], dY[:, n:n + 1])  # one point, then a block
    lp = g.append(X[:, n + 1:], y[n + 1:], dY[:, n + 1:])
    post = O.gradient_gp_fit(X, y, dY, "matern52", lam, 1.1, 0.03, 0.07)
    assert g.n == n + 7 and g.N == (n + 7) * (1 + d)
    assert abs(lp - post.logpdf) <= 1e-8 * (1 + abs(post.logpdf))
    Xs = np.random.default_rng(3).uniform(0, 1, (d, 50))
    mu, var = g.predict(Xs)
    mu_o, var_o = O.gradient_gp_mean_and_var(post, Xs)
    assert np.abs(mu - mu_o).max() <= 1e-8 and np.abs(var - var_o).max() <= 1e-8
    ll, gr = g.loglike_grad()
    _, gr_o = O.gradient_gp_loglike_grad(X, y, dY, "matern52", lam, 1.1, 0.03, 0.07)
    assert np.abs(gr - gr_o).max() <= 1e-7 * (1 + np.abs(gr_o).max())
    g.close()
    # the fitter
    P = 2
    Y = np.stack([y, 0.5 * y + 0.1])
    dYs = np.stack([dY, 0.5 * dY])
    model = HipGradientGaussianProcess([B.MvLogNormal(np.full(d, -0.7), np.full(d, 0.4)) for _ in range(P)], [B.LogNormal(0.0, 0.4) for _ in range(P)],
                                       [B.LogNormal(-3.0, 0.5) for _ in range(P)], [B.Dirac(0.07) for _ in range(P)])
    data = GradientData(X, Y, dYs)
    fit = B.HipGradientMAP(multistart=3, iters=12, seed=2)

    class _Prob:                                              # estimate_parameters reads .model and .data only
        pass
    prob = _Prob()
    prob.model, prob.data = model, data
    from boss_jl_amd import distributed as dist_util
    rng = np.random.default_rng(dist_util.shared_seed(2, None))
    sampler = model.params_sampler()
    starts = [sampler(rng) for _ in range(3)]
    prior_ll = model.params_loglike()
    llg = model.data_loglike_grad(data)
    f0 = [llg(p)[0] + prior_ll(p) for p in starts]
    for h in llg.handles:
        h.close()
    res = fit.estimate_parameters(prob, return_all=True)
    assert len(res) == 3
    for r, f_start, p0 in zip(res, f0, starts):
        assert r.loglike >= f_start - 1e-9                    # never below its start
        assert np.array_equal(r.params.grad_noise_std, p0.grad_noise_std)   # Dirac prior: fixed (dirac.jl:36-77)
    assert max(r.loglike for r in res) > max(f0) + 1.0        # and the climb is real
    best = fit.estimate_parameters(prob)
    assert best.loglike == max(r.loglike for r in res)
    # oracle value of the winner's log-likelihood
    want = sum(O.gradient_gp_fit(X, Y[i], dYs[i], "matern52", best.params.lengthscales[:, i], best.params.amplitudes[i],
                                 best.params.noise_std[i], best.params.grad_noise_std[i]).logpdf for i in range(P)) + prior_ll(best.params)
    assert abs(best.loglike - want) <= 1e-7 * (1 + abs(want))


def test_gradient_gp_acquisition_and_host_mirror(api, O):
    """EI × feasibility over gradient posteriors (P = 2 outputs, S = 2 samples) through boss_acq_ei and through the
    host mirror (HipGradientGaussianProcess + HipBatchAM in its three shard modes); likelihood through data_loglike."""
    import boss_jl_amd as B
    rng = np.random.default_rng(31)
    d, n, M, P, S = 2, 35, 90, 2, 2
    X = rng.uniform(0, 1, (d, n))
    Y = np.stack([np.sin(3 * X[0]) * np.cos(2 * X[1]), X[0] - X[1] ** 2])
    dY = np.stack([np.stack([3 * np.cos(3 * X[0]) * np.cos(2 * X[1]), -2 * np.sin(3 * X[0]) * np.sin(2 * X[1])]),
                   np.stack([np.ones(n), -2 * X[1]])])                     # P × d × n
    Xs = np.asfortranarray(rng.uniform(-0.1, 1.1, (d, M)))
    y_max, coefs = np.array([np.inf, 0.3]), [1.0, 0.2]
    prm = [B.HipGradientGPParams(rng.uniform(0.4, 0.8, (d, P)), rng.uniform(0.8, 1.4, P), rng.uniform(0.02, 0.06, P),
                                 rng.uniform(0.05, 0.2, P)) for _ in range(S)]
    data = B.GradientData(X, Y, dY)
    model = B.HipGradientGaussianProcess([None] * P, [None] * P, [None] * P, [None] * P)
    oposts = [[O.gradient_gp_fit(X, Y[i], dY[i], "matern52", p.lengthscales[:, i], p.amplitudes[i], p.noise_std[i],
                                 p.grad_noise_std[i]) for i in range(P)] for p in prm]
    b = O.best_so_far(coefs, Y, y_max)
    mask = O.in_bounds(Xs, np.zeros(d), np.ones(d))
    want = np.zeros(M)
    for s in range(S):
        mom = [O.gradient_gp_mean_and_var(oposts[s][i], Xs) for i in range(P)]
        mu, var = np.stack([m[0] for m in mom]), np.stack([m[1] for m in mom])
        want += O.expected_improvement_lin(coefs, mu, var, b) * O.feas_prob(mu, var, y_max)
    want = np.where(mask, want / S, 0.0)
    posts = model.model_posterior(prm, data)
    acq, am, mx = api.acq_ei([[s.gp for s in p.slices] for p in posts], api.Candidates(Xs), coefs, y_max, b, mask)
    assert np.allclose(acq, want, rtol=0, atol=1e-10) and am == int(np.argmax(want))
    mu_h, var_h = posts[0].mean_and_var(Xs[:, 3])
    assert np.allclose(mu_h, [O.gradient_gp_mean_and_var(oposts[0][i], Xs[:, 3])[0][0] for i in range(P)], atol=1e-10)
    for p in posts:
        p.close()
    prob = B.BossProblem(None, B.Domain((np.zeros(d), np.ones(d))), B.ExpectedImprovement(B.LinFitness(coefs)), model, data,
                         y_max, prm)
    for mode in ("candidates", "outputs", "samples"):
        x, val = B.HipBatchAM(points=Xs, shard=mode).maximize_acquisition(prob)
        assert np.array_equal(x, Xs[:, int(np.argmax(want))]) and abs(val - want.max()) <= 1e-10, mode
    # OptimizationAM semantics on gradient posteriors (optimization.jl:89-118 through gradient_gp.jl:334-361): multistart ascent on
    # the device gradients ends at a point of the domain that is at least as good as the best of its starts
    gam = B.HipGradientAM(x_prior=lambda r: r.uniform(0, 1, d), multistart=40, iters=15, seed=5)
    xg, vg = gam.maximize_acquisition(prob)
    starts = np.stack([np.random.default_rng(5).uniform(0, 1, d)], axis=1)
    pp = model.model_posterior(prm, data)
    a0, _, _ = api.acq_ei([[s.gp for s in p.slices] for p in pp], api.Candidates(np.asfortranarray(starts)), coefs, y_max, b, None)
    mg, vgrad, dmg, dvg = pp[0].slices[0].mean_and_var_grad(starts)
    assert dmg.shape == (d, 1) and np.isfinite(dmg).all()
    for p in pp:
        p.close()
    assert np.all(xg >= 0) and np.all(xg <= 1) and vg >= a0[0] - 1e-12
    ll = model.data_loglike(data)
    assert abs(ll(prm[0]) - sum(o.logpdf for o in oposts[0])) <= 1e-9 * (1 + abs(sum(o.logpdf for o in oposts[0])))
    assert np.allclose(model.data_loglike_batch(data, prm), [sum(o.logpdf for o in op) for op in oposts], rtol=1e-9)
    for g in ll.handles:
        g.close()
    d2 = data.augment([0.5, 0.5], [0.1, 0.2], [[1.0, 2.0], [3.0, 4.0]])
    assert d2.X.shape == (d, n + 1) and d2.dY.shape == (P, d, n + 1) and d2.slice(1).dY.shape == (1, d, n + 1)


def test_gradient_gp_full_size(api, O):
    """n = 4096, d = 8: the 36 864-row augmented system (10.9 GB).  The CPU oracle needs minutes
    for this size, so the check is through properties: the factor reproduces sampled entries of the augmented
    matrix, the posterior interpolates values at training points, and a candidate slice predicted alone equals
    the same slice of the 8192-candidate call."""
    d, n, M = 8, 4096, 8192
    X, y, dY = make_grad(d, n)
    Xs = np.asfortranarray(np.random.default_rng(5).uniform(0, 1, (d, M)))
    lam = np.full(d, 0.4)
    g = api.GradGP(X, y, dY, "matern52")
    lp = g.update(lam, 1.2, 1e-3, 1e-2)
    assert np.isfinite(lp)
    mu, var = g.predict(Xs)
    assert np.isfinite(mu).all() and (var >= 0).all() and var.max() <= 1.2 ** 2 * (1 + 1e-12)
    mu_t, var_t = g.predict(X[:, :64])
    assert np.abs(mu_t - y[:64]).max() <= 1e-4 and var_t.max() <= 1e-5          # interpolation
    h = 1e-5                                                                     # ... of the gradients too
    e = np.zeros((d, 1))
    e[2] = h
    dmu = (g.predict(X[:, :64] + e)[0] - g.predict(X[:, :64] - e)[0]) / (2 * h)
    assert np.abs(dmu - dY[2, :64]).max() <= 1e-2 * np.abs(dY).max()
    mu_s, var_s = g.predict(Xs[:, 100:133])                                      # few-candidates path vs fused kernel
    assert np.allclose(mu_s, mu[100:133], rtol=0, atol=1e-9) and np.allclose(var_s, var[100:133], rtol=0, atol=1e-9)
    g.close()


# ------------------------------------------------------------------------------------------
# NonstationaryGP — the Gibbs kernel with input-dependent λ(x), α(x), σ(x)
# ------------------------------------------------------------------------------------------
def latent(d):
    f_lam = lambda x: 0.25 + 0.5 * np.asarray(x) ** 2 + 0.1 * np.arange(1, d + 1)        # noqa: E731
    f_amp = lambda x: 1.0 + 0.4 * np.sin(3 * x[0])                                        # noqa: E731
    f_noise = lambda x: 0.03 + 0.05 * x[-1] ** 2                                           # noqa: E731
    return f_lam, f_amp, f_noise


@pytest.mark.parametrize("d,N,M", [(1, 1, 2), (2, 50, 33), (8, 300, 70), (3, 1100, 40), (20, 200, 65)])
def test_nonstationary_gp_parity(api, O, d, N, M):
    """Gibbs Gram + factor + logpdf + mean_and_var against the oracle (fused kernel below 1024 rows, the
    few-candidates path above), with a prior mean."""
    X, y, Xs = make(d, N, M, seed=4)
    f_lam, f_amp, f_noise = latent(d)
    ev = lambda f, Z: np.array([f(Z[:, j]) for j in range(Z.shape[1])])                   # noqa: E731
    lamX, ampX, noiX = ev(f_lam, X).T, ev(f_amp, X), ev(f_noise, X)
    lamS, ampS = ev(f_lam, Xs).T, ev(f_amp, Xs)
    mX, mS = 0.3 * X[0], 0.3 * Xs[0]
    post = O.nonstationary_fit(X, y, lamX, ampX, noiX, mean=mX)
    mu_o, var_o = O.nonstationary_mean_and_var(post, Xs, lamS, ampS, mean_s=mS, clip=False)
    K = post.L @ post.L.T
    tol = max(1e-9, np.linalg.cond(K) * N * 2.0 ** -53 * 8)
    g = api.GibbsGP(X, y)
    lp = g.update(lamX, ampX, noiX, mX)
    L, z = g.factor()
    assert abs(lp - post.logpdf) <= tol * (1 + abs(post.logpdf))
    assert np.abs(L - post.L).max() <= tol * np.abs(post.L).max()
    mu, var = g.predict(Xs, lamS, ampS, mS)
    assert np.abs(mu - mu_o).max() <= tol * (1 + np.abs(mu_o).max())
    assert np.abs(var - np.where(var_o >= 0, var_o, 0.0)).max() <= tol * ampS.max() ** 2
    mu1, var1 = g.predict(Xs[:, :1], lamS[:, :1], ampS[:1], mS[:1])
    assert abs(mu1[0] - mu_o[0]) <= tol * (1 + abs(mu_o[0])) and abs(var1[0] - max(var_o[0], 0.0)) <= tol * ampS.max() ** 2
    g.close()


@pytest.mark.parametrize("d,N,disc", [(1, 5, False), (3, 150, True), (8, 700, False), (16, 1100, False)])
def test_nonstationary_gp_likelihood_gradient(api, O, d, N, disc):
    """∂ℓ/∂ of the latent values at the training points (boss_ngp_loglike_grad) — what a fitter's AD hands back to the latent
    models through finite_nongp (nonstationary_gp.jl:183-196, 237-245) — against the oracle (finite-difference checked,
    tests/test_oracle_crosscheck.py); prior mean and a rounded dimension included."""
    X, y, _ = make(d, N, 4, seed=31)
    discrete = None
    if disc:
        discrete = np.zeros(d, bool)
        discrete[1] = True
        X[1] *= 5
    f_lam, f_amp, f_noise = latent(d)
    rnd = lambda Z: Z if discrete is None else np.where(discrete[:, None], np.rint(Z), Z)   # noqa: E731
    ev = lambda f, Z: np.array([f(Z[:, j]) for j in range(Z.shape[1])])                   # noqa: E731
    lamX, ampX, noiX = ev(f_lam, rnd(X)).T, ev(f_amp, rnd(X)), ev(f_noise, X)
    mX = 0.3 * X[0]
    ll_o, dl_o, da_o, dn_o, dm_o = O.nonstationary_loglike_grad(X, y, lamX, ampX, noiX, mean=mX, discrete=discrete)
    post = O.nonstationary_fit(X, y, lamX, ampX, noiX, mean=mX, discrete=discrete)
    tol = max(1e-9, np.linalg.cond(post.L @ post.L.T) * N * 2.0 ** -53 * 8)
    g = api.GibbsGP(X, y, discrete)
    with pytest.raises(api.BossError):
        g.loglike_grad()                                      # not fitted
    g.update(lamX, ampX, noiX, mX)
    ll, dl, da, dn, dm = g.loglike_grad()
    assert abs(ll - ll_o) <= tol * (1 + abs(ll_o))
    for got, want in ((dl, dl_o), (da, da_o), (dn, dn_o), (dm, dm_o)):
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= tol * (1 + np.abs(want).max()) * 100, (np.abs(got - want).max(), np.abs(want).max())
    ll2, dl2, da2, dn2, dm2 = g.loglike_grad()                # repeated: bit-identical
    assert np.array_equal(dl, dl2) and np.array_equal(da, da2) and np.array_equal(dn, dn2) and np.array_equal(dm, dm2)
    mu, var = g.predict(X[:, :3], lamX[:, :3], ampX[:3], mX[:3])   # the handle's factor is untouched
    mu_o, var_o = O.nonstationary_mean_and_var(post, X[:, :3], lamX[:, :3], ampX[:3], mean_s=mX[:3])
    assert np.abs(mu - mu_o).max() <= tol * (1 + np.abs(mu_o).max()) * 10 and np.abs(var - var_o).max() <= tol * 10
    g.close()


@pytest.mark.parametrize("disc", [False, True])
def test_nonstationary_gp_append(api, O, disc):
    """augment_dataset! (src/types/problem.jl:191-198) on a nonstationary posterior = a fresh fit on all points with the latent models
    evaluated at the new ones (boss_ngp_append through the C ABI and through the host mirror), prior mean and discrete dimensions
    included; refused before the first update."""
    from boss_jl_amd.nonstationary import HipNonstationaryGP
    from boss_jl_amd.problem import ExperimentData
    d, N, n_new = 3, 150, 9
    X, y, Xs = make(d, N + n_new, 40, seed=21)
    discrete = np.array([False, True, False]) if disc else None
    if disc:
        X[1] *= 5
        Xs[1] *= 5
    f_lam, f_amp, f_noise = latent(d)
    rnd = lambda Z: Z if discrete is None else np.where(discrete[:, None], np.rint(Z), Z)   # noqa: E731
    ev = lambda f, Z: np.array([f(Z[:, j]) for j in range(Z.shape[1])])                   # noqa: E731
    lamX, ampX, noiX = ev(f_lam, rnd(X)).T, ev(f_amp, rnd(X)), ev(f_noise, X)
    mfun = lambda Z: 0.3 * Z[0]                                                           # noqa: E731
    g = api.GibbsGP(X[:, :N], y[:N], discrete)
    with pytest.raises(api.BossError):
        g.append(X[:, N:], y[N:], lamX[:, N:], ampX[N:], noiX[N:], mfun(X[:, N:]))
    g.update(lamX[:, :N], ampX[:N], noiX[:N], mfun(X[:, :N]))
    with pytest.raises(api.BossError):
        g.append(X[:, N:], y[N:], lamX[:, N:], ampX[N:], noiX[N:])                        # the posterior has a prior mean
    lp = g.append(X[:, N:N + 1], y[N:N + 1], lamX[:, N:N + 1], ampX[N:N + 1], noiX[N:N + 1], mfun(X[:, N:N + 1]))
    lp = g.append(X[:, N + 1:], y[N + 1:], lamX[:, N + 1:], ampX[N + 1:], noiX[N + 1:], mfun(X[:, N + 1:]))
    post = O.nonstationary_fit(X, y, lamX, ampX, noiX, mean=mfun(X), discrete=discrete)
    assert g.N == N + n_new and abs(lp - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf))
    lamS, ampS = ev(f_lam, rnd(Xs)).T, ev(f_amp, rnd(Xs))
    mu, var = g.predict(Xs, lamS, ampS, mfun(Xs))
    mu_o, var_o = O.nonstationary_mean_and_var(post, Xs, lamS, ampS, mean_s=mfun(Xs))
    assert np.abs(mu - mu_o).max() <= 1e-9 and np.abs(var - var_o).max() <= 1e-9
    g.close()
    # the host mirror evaluates the latent models itself
    model = HipNonstationaryGP([f_lam], [f_amp], [f_noise], [lambda x: 0.3 * x[0]], None if discrete is None else list(discrete))
    sl = model.model_posterior_slice(ExperimentData(X[:, :N], y[None, :N]), 0)
    lp_h = sl.append(X[:, N:], y[N:])
    assert abs(lp_h - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf))
    mu_h, var_h = sl.mean_and_var(Xs)
    assert np.abs(mu_h - mu_o).max() <= 1e-9 and np.abs(var_h - var_o).max() <= 1e-9
    sl.close()


@pytest.mark.parametrize("d,N,M", [(1, 20, 5), (3, 300, 70), (8, 1100, 40), (16, 150, 33)])
def test_nonstationary_gp_candidate_gradients(api, O, d, N, M):
    """SURVEY §8f3 over §8f4: ∇μ, ∇σ² of a nonstationary posterior — the candidate enters the Gibbs kernel directly and through
    the latent λ(x*), α(x*) (nonstationary_gp.jl:61-107,153-196 under the ForwardDiff of optimization.jl:36) — and the EI chain rule
    from those moments.  Against the oracle's analytic gradients (finite-difference checked in tests/test_oracle_crosscheck.py)."""
    X, y, Xs = make(d, N, M, seed=14)
    Xs = np.asfortranarray(0.05 + 0.9 * Xs)
    w = np.linspace(0.5, 1.5, d)
    f_lam = lambda x: 0.3 + 0.4 * np.asarray(x) ** 2 + 0.05 * np.arange(1, d + 1) + 0.1 * np.sin(w @ np.asarray(x))    # noqa: E731
    J_lam = lambda x: np.diag(0.8 * np.asarray(x)) + 0.1 * np.cos(w @ np.asarray(x)) * np.tile(w, (d, 1))               # noqa: E731
    f_amp = lambda x: 1.0 + 0.4 * np.sin(3 * x[0]) + 0.1 * x[-1]                                                        # noqa: E731
    def J_amp(x):
        g = np.zeros(d)
        g[0] += 1.2 * np.cos(3 * x[0])
        g[-1] += 0.1
        return g
    f_noise = lambda x: 0.05 + 0.02 * x[0]                                                                               # noqa: E731
    ev = lambda f, Z: np.array([f(Z[:, j]) for j in range(Z.shape[1])])                                                  # noqa: E731
    lamX, ampX, noiX = ev(f_lam, X).T, ev(f_amp, X), ev(f_noise, X)
    lamS, ampS = ev(f_lam, Xs).T, ev(f_amp, Xs)
    Dl = np.stack([J_lam(Xs[:, j]) for j in range(M)], axis=2)
    Da = np.stack([J_amp(Xs[:, j]) for j in range(M)], axis=1)
    mX, mS = 0.3 * X[0], 0.3 * Xs[0]
    mg = np.zeros((d, M))
    mg[0] = 0.3
    post = O.nonstationary_fit(X, y, lamX, ampX, noiX, mean=mX)
    mu_o, var_o, dmu_o, dvar_o = O.nonstationary_mean_and_var_grad(post, Xs, lamS, ampS, Dl, Da, mS, mg)
    K = post.L @ post.L.T
    tol = max(1e-9, np.linalg.cond(K) * N * 2.0 ** -53 * 8)
    g = api.GibbsGP(X, y)
    g.update(lamX, ampX, noiX, mX)
    mu, var, dmu, dvar = g.predict_grad(Xs, lamS, ampS, Dl, Da, mS, mg)
    assert np.abs(mu - mu_o).max() <= tol * (1 + np.abs(mu_o).max())
    assert np.abs(var - np.maximum(var_o, 0.0)).max() <= tol * ampS.max() ** 2
    assert np.abs(dmu - dmu_o).max() <= tol * (1 + np.abs(dmu_o).max()) * 10, np.abs(dmu - dmu_o).max()
    assert np.abs(dvar - dvar_o).max() <= tol * (1 + np.abs(dvar_o).max()) * 10, np.abs(dvar - dvar_o).max()
    # constant latent models: no Jacobians
    mu_c, var_c, dmu_c, dvar_c = O.nonstationary_mean_and_var_grad(post, Xs, lamS, ampS)
    _, _, dmu2, dvar2 = g.predict_grad(Xs, lamS, ampS)
    assert np.abs(dmu2 - dmu_c).max() <= tol * (1 + np.abs(dmu_c).max()) * 10 and np.abs(dvar2 - dvar_c).max() <= tol * (1 + np.abs(dvar_c).max()) * 10
    # the acquisition's chain rule from the moments (boss_acq_ei_grad_moments)
    best = float(y.max()) - 0.2
    vo = np.maximum(var_o, 0.0)
    acq_o, dacq_o = O.expected_improvement_lin_grad([1.0], mu_o[None], vo[None], dmu_o[None], np.where(vo > 0, dvar_o, 0.0)[None], best)
    acq, dacq = api.acq_ei_grad_moments(mu[None], var[None], dmu[None], dvar[None], [1.0], None, best)
    assert np.abs(acq - acq_o).max() <= tol * 10 and np.abs(dacq - dacq_o).max() <= tol * (1 + np.abs(dacq_o).max()) * 100
    g.close()


def test_nonstationary_gp_constant_parameters_equal_the_sqexp_model(api, O):
    """With constant latent parameters the Gibbs kernel is the ARD squared-exponential kernel: the nonstationary
    entry points must reproduce the plain ones on the device as well (the plain model adds 1e-8 to its parameters)."""
    d, N, M = 4, 500, 16400                                  # M >= 16384: the 64-candidate tiles
    X, y, Xs = make(d, N, M, seed=6)
    lam, amp, sig = np.linspace(0.4, 0.9, d), 1.2, 0.05
    gp = api.GP(X, y, "sqexp")
    lp_p = gp.update(lam - 1e-8, amp - 1e-8, sig - 1e-8)
    mu_p, var_p = gp.predict(Xs)
    g = api.GibbsGP(X, y)
    lp = g.update(np.tile(lam[:, None], (1, N)), np.full(N, amp), np.full(N, sig))
    mu, var = g.predict(Xs, np.tile(lam[:, None], (1, M)), np.full(M, amp))
    assert abs(lp - lp_p) <= 1e-9 * (1 + abs(lp_p))
    assert np.allclose(mu, mu_p, rtol=0, atol=1e-9) and np.allclose(var, var_p, rtol=0, atol=1e-9)
    gp.close()
    g.close()


def test_nonstationary_gp_discrete_errors_and_host_mirror(api, O):
    import boss_jl_amd as B
    rng = np.random.default_rng(12)
    d, N, M = 2, 60, 45
    X = rng.uniform(0, 5, (d, N))
    Y = np.stack([np.sin(X[0]) + 0.1 * X[1], np.cos(X).sum(0)])
    Xs = np.asfortranarray(rng.uniform(0, 5, (d, M)))
    disc = np.array([False, True])
    f_lam = lambda x: 0.8 + 0.1 * np.asarray(x)                                           # noqa: E731
    f_amp = lambda x: 1.0 + 0.05 * x[1]                                                   # noqa: E731
    f_noise = lambda x: 0.05 + 0.01 * x[0]                                                # noqa: E731
    model = B.HipNonstationaryGP([f_lam, f_lam], [f_amp, f_amp], [f_noise, f_noise], mean=[lambda x: 0.1, None], discrete=disc)
    data = B.ExperimentData(X, Y)
    posts = model.model_posterior(data)
    ev = lambda f, Z: np.array([f(Z[:, j]) for j in range(Z.shape[1])])                   # noqa: E731
    Xr, Xsr = O.discrete_round(X, disc), O.discrete_round(Xs, disc)
    tot = 0.0
    for i in range(2):
        mean = None if i else np.full(N, 0.1)
        op = O.nonstationary_fit(X, Y[i], ev(f_lam, Xr).T, ev(f_amp, Xr), ev(f_noise, X), mean=mean, discrete=disc)
        tot += op.logpdf
        mu_o, var_o = O.nonstationary_mean_and_var(op, Xs, ev(f_lam, Xsr).T, ev(f_amp, Xsr), mean_s=None if i else np.full(M, 0.1))
        mu, var = posts[i].mean_and_var(Xs)
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9)
        m1, v1 = posts[i].mean_and_var(Xs[:, 4])
        assert abs(m1 - mu_o[4]) <= 1e-9 and abs(v1 - var_o[4]) <= 1e-9
    assert abs(model.data_loglike(data) - tot) <= 1e-9 * (1 + abs(tot))
    # EI on the predicted moments (the acquisition route for these posteriors)
    mom = [p.mean_and_var(Xs) for p in posts]
    mu, var = np.stack([m[0] for m in mom]), np.stack([m[1] for m in mom])
    acq, am, mx = api.acq_ei_moments(mu[None], var[None], [1.0, 0.0], [np.inf, 1.0], 0.5, None)
    want = O.expected_improvement_lin([1.0, 0.0], mu, var, 0.5) * O.feas_prob(mu, var, [np.inf, 1.0])
    assert np.allclose(acq, want, rtol=0, atol=1e-12) and am == int(np.argmax(want))
    g = posts[0].gp
    for call in (lambda: api.GP.update(g, [1.0, 1.0], 1.0, 0.1), lambda: api.GP.predict(g, Xs), lambda: api.GP.append(g, X[:, :1], [0.0]),
                 lambda: api.GP.predict_grad(g, Xs), lambda: g.predict_cov(Xs), lambda: api.GP.loglike_grad(g),
                 lambda: api.acq_ei([[g]], api.Candidates(Xs), [1.0], None, 0.0, None),
                 lambda: g.update(np.zeros((d, N)), np.ones(N), np.ones(N)),             # λ = 0
                 lambda: g.update(np.ones((d, N)), -np.ones(N), np.ones(N))):
        with pytest.raises(api.BossError):
            call()
    # the mirror's gradient of mean_and_var (latent Jacobians by central differences of the host closures) against its own differences
    Xg = np.asfortranarray(Xs[:, :5])
    if posts[0].discrete is None:
        mu_g, var_g, dmu_g, dvar_g = posts[0].mean_and_var_grad(Xg)
        for m in range(d):
            Zp, Zm = Xg.copy(), Xg.copy()
            Zp[m] += 1e-5
            Zm[m] -= 1e-5
            (mp, vp), (mm, vm) = posts[0].mean_and_var(Zp), posts[0].mean_and_var(Zm)
            assert np.abs((mp - mm) / 2e-5 - dmu_g[m]).max() <= 1e-5 * (1 + np.abs(dmu_g).max())
    # identical points without noise -> PosDefException; data_loglike maps it to -Inf (safe_data_loglike)
    Xd = np.tile(X[:, :1], (1, 4))
    gd = api.GibbsGP(Xd, np.arange(4.0))
    with pytest.raises(api.PosDefException):
        gd.update(np.ones((d, 4)), np.ones(4), np.zeros(4))
    gd.close()
    for p in posts:
        p.close()


def test_nonstationary_gp_full_size(api, O):
    """N = 4096, d = 8, 8192 candidates with input-dependent λ, α, σ: against the oracle's LAPACK fit."""
    d, N, M = 8, 4096, 8192
    X, y, Xs = make(d, N, M, seed=1)
    f_lam, f_amp, f_noise = latent(d)
    ev = lambda f, Z: np.array([f(Z[:, j]) for j in range(Z.shape[1])])                   # noqa: E731
    lamX, ampX, noiX = ev(f_lam, X).T, ev(f_amp, X), ev(f_noise, X)
    lamS, ampS = ev(f_lam, Xs).T, ev(f_amp, Xs)
    post = O.nonstationary_fit(X, y, lamX, ampX, noiX)
    g = api.GibbsGP(X, y)
    lp = g.update(lamX, ampX, noiX)
    assert abs(lp - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf))
    mu, var = g.predict(Xs, lamS, ampS)
    mu_o, var_o = O.nonstationary_mean_and_var(post, Xs[:, :512], lamS[:, :512], ampS[:512])
    assert np.allclose(mu[:512], mu_o, rtol=0, atol=1e-9) and np.allclose(var[:512], var_o, rtol=0, atol=1e-9)
    g.close()


def test_random_call_sequences_on_poisoned_allocations():
    """tools/fuzz.py — random shapes and random sequences of update / predict / gradients / covariance / append /
    likelihood gradient on one handle, each result checked against the oracle — with every new device allocation
    filled with NaN patterns (BOSS_POISON_ALLOC=1): nothing may depend on the zeros of fresh memory."""
    import subprocess
    import sys
    env = dict(os.environ, BOSS_POISON_ALLOC="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz.py"), "30", "5"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "30 cases passed" in r.stdout


def test_nonstationary_gp_with_parametrized_gp_latents(api, O):
    """The reference's full NonstationaryGP: every lengthscale is a ParametrizedGP (parametrized_gp.jl) — a latent GP over
    the data whose whitened outputs are parameters, pushed through Normal-cdf -> target quantile -> activation.  The
    latent posteriors are evaluated on the device for all points at once; the result must equal the oracle's chain."""
    import boss_jl_amd as B
    from scipy import stats
    from scipy.special import ndtr
    rng = np.random.default_rng(40)
    d, N, M = 2, 120, 50
    X = rng.uniform(0, 1, (d, N))
    Y = np.sin(5 * X[0:1]) + X[1:2]
    Xs = np.asfortranarray(rng.uniform(0, 1, (d, M)))
    data = B.ExperimentData(X, Y)
    act = [lambda z: 0.1 + 0.9 * z, lambda z: 0.2 + 0.5 * z]
    latents = [B.HipParametrizedGP([0.3, 0.3], "matern32", stats.beta(2, 3), act[k], 1e-2) for k in range(d)]
    params = [latents[k].params_sampler(data)(rng) for k in range(d)]
    # oracle chain for one latent: prior Cholesky, de-whitening, posterior mean, transform
    def latent_oracle(k, Z):
        prior = O.gp_fit(X, np.zeros(N), "matern32", [0.3, 0.3], 1.0, 1e-2)
        assert np.abs(prior.L - params[k].L).max() <= 1e-10
        yk = prior.L @ params[k].yeps
        post = O.gp_fit(X, yk, "matern32", [0.3, 0.3], 1.0, 1e-2)
        m = O.gp_mean_and_var(post, Z)[0]
        return act[k](stats.beta(2, 3).ppf(ndtr(m)))
    posts = [latents[k].model_posterior(params[k], data) for k in range(d)]
    look = [latents[k].model_posterior_lookup(params[k], data) for k in range(d)]
    for k in range(d):
        want = latent_oracle(k, Xs)
        assert np.allclose(posts[k](Xs), want, rtol=0, atol=1e-7)
        assert abs(posts[k](Xs[:, 3]) - want[3]) <= 1e-7
        yk = params[k].L @ params[k].yeps                                         # lookup: transform of the de-whitened outputs themselves
        assert np.allclose(look[k](X), act[k](stats.beta(2, 3).ppf(ndtr(yk))), rtol=0, atol=1e-12)
        assert abs(latents[k].params_loglike()(params[k]) - stats.norm.logpdf(params[k].yeps).sum()) <= 1e-10
    f_lam = B.stack_latents(posts)
    model = B.HipNonstationaryGP([f_lam], [lambda x: 1.1], [lambda x: 0.05])
    post = model.model_posterior(data)[0]
    lamX = np.stack([latent_oracle(k, X) for k in range(d)])
    lamS = np.stack([latent_oracle(k, Xs) for k in range(d)])
    op = O.nonstationary_fit(X, Y[0], lamX, np.full(N, 1.1), np.full(N, 0.05))
    mu_o, var_o = O.nonstationary_mean_and_var(op, Xs, lamS, np.full(M, 1.1))
    mu, var = post.mean_and_var(Xs)
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-6) and np.allclose(var, var_o, rtol=0, atol=1e-6)
    # likelihood with the cheap lookup latents (data_loglike_slice uses finite_nongp_lookup, nonstationary_gp.jl:237-245)
    model_l = B.HipNonstationaryGP([B.stack_latents(look)], [lambda x: 1.1], [lambda x: 0.05])
    lamL = np.stack([act[k](stats.beta(2, 3).ppf(ndtr(params[k].L @ params[k].yeps))) for k in range(d)])
    ll = model_l.data_loglike(data)
    assert abs(ll - O.nonstationary_fit(X, Y[0], lamL, np.full(N, 1.1), np.full(N, 0.05)).logpdf) <= 1e-9 * (1 + abs(ll))
    post.close()
    for p in posts:
        p.close()


@pytest.mark.parametrize("N0,count", [(1000, 6), (1279, 3), (2040, 12), (4090, 5)])
def test_rank_one_appends_on_resident_inverses(api, O, N0, count):
    """Single-observation appends: from the second one on a set of hyper-parameters the factor, z, the log-likelihood and
    every inverse the prediction paths use are extended by one row from the resident inverse factors.  Every state in
    between equals a fresh fit of the augmented data: factor, logpdf, small and large predictions, gradients, tracked
    candidates; storage growth and a new update switch back cleanly."""
    rng = np.random.default_rng(N0)
    d = 3
    X = rng.uniform(0, 1, (d, N0 + count))
    y = np.sin(3 * X).sum(0) / 2 + 0.05 * rng.standard_normal(N0 + count)
    mean = 0.1 * X[0]
    lam = np.array([0.4, 0.5, 0.6])
    Xs = np.asfortranarray(rng.uniform(0, 1, (d, 300)))
    g = api.GP(X[:, :N0], y[:N0], "matern52")
    g.reserve(N0 + count)
    g.update(lam, 1.1, 0.05, mean[:N0])
    cand = api.Candidates(Xs[:, :100])
    tr = api.Track(g, cand, 0.1 * Xs[0, :100])
    for i in range(count):
        n1 = N0 + i + 1
        lp = g.append(X[:, n1 - 1], y[n1 - 1], mean[n1 - 1:n1])
        post = O.gp_fit(X[:, :n1], y[:n1], "matern52", lam, 1.1, 0.05, mean=mean[:n1])
        assert abs(lp - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf)), i
        L, z = g.factor()
        assert np.abs(L - post.L).max() <= 1e-10 and np.abs(z - np.linalg.solve(post.L, y[:n1] - mean[:n1])).max() <= 1e-9, i
        for M in (1, 40, 300):                               # one-pass kernel, inverse GEMMs, (first call) step path
            mu, var = g.predict(Xs[:, :M], 0.1 * Xs[0, :M])
            mu_o, var_o = O.gp_mean_and_var(post, Xs[:, :M], 0.1 * Xs[0, :M])
            assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9), (i, M)
        mu_t, var_t = tr.moments()
        mu_o, var_o = O.gp_mean_and_var(post, Xs[:, :100], 0.1 * Xs[0, :100], clip=False)
        assert np.allclose(mu_t, mu_o, rtol=0, atol=1e-9) and np.allclose(var_t, var_o, rtol=0, atol=1e-9), i
    if count >= 6:                                           # a few observations at once on the resident inverses
        g3 = api.GP(X[:, :N0], y[:N0], "matern52")
        g3.reserve(N0 + count)
        g3.update(lam, 1.1, 0.05, mean[:N0])
        g3.append(X[:, N0], y[N0], mean[N0:N0 + 1])
        g3.append(X[:, N0 + 1], y[N0 + 1], mean[N0 + 1:N0 + 2])
        lp3 = g3.append(X[:, N0 + 2:N0 + 6], y[N0 + 2:N0 + 6], mean[N0 + 2:N0 + 6])
        p3 = O.gp_fit(X[:, :N0 + 6], y[:N0 + 6], "matern52", lam, 1.1, 0.05, mean=mean[:N0 + 6])
        assert abs(lp3 - p3.logpdf) <= 1e-9 * (1 + abs(p3.logpdf))
        assert np.abs(g3.factor()[0] - p3.L).max() <= 1e-10
        g3.close()
    mu, var, dmu, dvar = g.predict_grad(Xs[:, :20], 0.1 * Xs[0, :20])
    _, _, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs[:, :20], 0.1 * Xs[0, :20])
    assert np.allclose(dmu, dmu_o, rtol=0, atol=1e-8 * (1 + np.abs(dmu_o).max()))
    assert np.allclose(dvar, dvar_o, rtol=0, atol=1e-8 * (1 + np.abs(dvar_o).max()))
    _, gr = g.loglike_grad()
    _, gr_o = O.gp_data_loglike_grad(X, y, "matern52", lam, 1.1, 0.05, mean=mean)
    assert np.allclose(gr, gr_o, rtol=0, atol=1e-8 * (1 + np.abs(gr_o).max()))
    # the full 8192-candidate kernel reads the patched 256×256 inverses
    Xb = np.asfortranarray(rng.uniform(0, 1, (d, 4200)))
    mu_b, var_b = g.predict(Xb)
    mu_bo, var_bo = O.gp_mean_and_var(O.gp_fit(X, y, "matern52", lam, 1.1, 0.05, mean=mean), Xb[:, :200])
    assert np.allclose(mu_b[:200], mu_bo, rtol=0, atol=1e-9) and np.allclose(var_b[:200], var_bo, rtol=0, atol=1e-9)
    # one more append grows the storage (block path), then appends on the new storage, then a re-fit
    xg = rng.uniform(0, 1, (d, 3))
    Xa, ya, ma = X, y, mean
    for j in range(3):
        lp = g.append(xg[:, j], 0.2, [0.1 * xg[0, j]])
        Xa, ya, ma = np.hstack([Xa, xg[:, j:j + 1]]), np.append(ya, 0.2), np.append(ma, 0.1 * xg[0, j])
        post = O.gp_fit(Xa, ya, "matern52", lam, 1.1, 0.05, mean=ma)
        assert abs(lp - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf)), j
        mu, var = g.predict(Xs[:, :3], 0.1 * Xs[0, :3])
        mu_o, var_o = O.gp_mean_and_var(post, Xs[:, :3], 0.1 * Xs[0, :3])
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9) and np.allclose(var, var_o, rtol=0, atol=1e-9), j
    lp = g.update(lam * 1.2, 1.0, 0.06, ma)
    assert abs(lp - O.gp_fit(Xa, ya, "matern52", lam * 1.2, 1.0, 0.06, mean=ma).logpdf) <= 1e-9 * (1 + abs(lp))
    tr.close()
    g.close()


@pytest.mark.parametrize("d,N,S", [(1, 7, 3), (3, 20, 5), (5, 128, 4), (32, 100, 3), (2, 17, 300), (3, 200, 5), (8, 700, 9), (4, 1300, 4)])   # N <= 128: one workgroup per set
def test_batched_likelihood_gradients(api, O, d, N, S):
    """boss_gp_loglike_grad_batch: values and gradients of S hyper-parameter sets in one call (batched factorisations,
    gradient passes on shared workspaces) equal the oracle's and the single-handle entry points', with per-set means,
    an invalid set and discrete dimensions."""
    rng = np.random.default_rng(N + S)
    disc = None if d < 3 else np.array([False] * (d - 1) + [True])
    scale = 5.0 if disc is not None else 1.0
    X = rng.uniform(0, scale, (d, N))
    y = np.sin(3 * X / scale).sum(0) + 0.05 * rng.standard_normal(N)
    lam = rng.uniform(0.3, 0.9, (d, S)) * scale
    amp, sig = rng.uniform(0.7, 1.4, S), rng.uniform(0.03, 0.1, S)
    means = np.stack([0.1 * X[0] + 0.01 * k for k in range(S)])
    sig[S - 1] = -1.0                                        # invalid set: status INVALID, ll = -Inf, zero gradient
    ll, st, grad = api.loglike_batch(X, y, "matern52", lam, amp, sig, means, disc, want_grad=True)
    assert grad.shape == (d + 2, S)
    for k in range(S - 1):
        ll_o, gr_o = O.gp_data_loglike_grad(X, y, "matern52", lam[:, k], amp[k], sig[k], mean=means[k], discrete=disc)
        assert st[k] == 0 and abs(ll[k] - ll_o) <= 1e-9 * (1 + abs(ll_o))
        assert np.allclose(grad[:, k], gr_o, rtol=0, atol=1e-8 * (1 + np.abs(gr_o).max())), k
    assert st[S - 1] == api.BOSS_E_INVALID and ll[S - 1] == -np.inf and np.all(grad[:, S - 1] == 0.0)
    g = api.GP(X, y, "matern52", disc)
    g.update(lam[:, 0], amp[0], sig[0], means[0])
    _, gr1 = g.loglike_grad()
    assert np.allclose(grad[:, 0], gr1, rtol=0, atol=1e-10 * (1 + np.abs(gr1).max()))
    g.close()


def test_bo_loop_example_runs():
    """examples/bo_loop.py: the whole loop (SampleOptMAP-style fit, gradient multistart acquisition maximisation, objective
    evaluation, dataset augmentation, a sequential batch) on the device — the best feasible value never gets worse."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bo_loop", os.path.join(ROOT, "examples", "bo_loop.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    problem = mod.main(iters=5, seed=1)
    assert problem.data.X.shape[1] == 8 + 5


# ------------------------------------------------------------------------------------------
# Pairwise-combinatorial integration test, after the reference's test/combinatorial suite (combinations.csv:
# ACTS degree-2 coverage over its 24 inputs; objective, data and bounds of input_values.jl:24-56): every pair of
# values of the inputs that concern this path occurs in at least one configuration; each runs `iter_max` BO
# iterations (or, with f = missing, returns a recommendation) through the plugin-trio mirror.
# ------------------------------------------------------------------------------------------
def _obj(x, rng):
    y = np.exp(x[0] / 10) * np.cos(2 * x[0]) + rng.normal(0.0, 0.1)
    z = 0.5 ** 6 * (x[0] ** 2 - 15.0 ** 2) + rng.normal(0.0, 0.1)
    return np.array([y, z])


COMBINATIONS = [
    # XY, f, discrete, cons, y_max, model, kernel, dirac priors, fitness, fitter, maximizer, iters
    ("dup", "missing", False, False, "inf", "semipar", "matern52", False, "nonlin", "sampling", "grid", 1),
    ("nodup", "fn", True, True, "finite", "gp", "matern32", True, "lin", "sampling", "sampling", 1),
    ("dup", "fn", False, True, "inf", "gp_mean", "sqexp", True, "lin", "optimization", "optimization", 1),
    ("nodup", "fn", False, False, "finite", "semipar", "matern52", True, "lin", "sampling", "seqbatch", 2),
    ("nodup", "missing", True, False, "finite", "gp", "sqexp", False, "lin", "sample_opt", "optimization", 1),
    ("dup", "fn", True, False, "inf", "gp", "matern52", False, "nonlin", "sample_opt", "sampling", 2),
    ("nodup", "fn", False, True, "inf", "gp_mean", "matern32", False, "nonlin", "optimization", "grid", 2),
    ("dup", "fn", False, False, "finite", "gp", "matern32", False, "lin", "optimization", "seqbatch", 1),
    ("nodup", "fn", True, True, "inf", "semipar", "sqexp", False, "lin", "sampling", "sampling", 2),
    ("dup", "missing", False, True, "finite", "gp_mean", "matern52", True, "lin", "sample_opt", "grid", 1),
    ("nodup", "fn", False, False, "inf", "gp", "matern52", True, "lin", "sampling", "optimization", 2),
    ("dup", "fn", True, True, "finite", "gp_mean", "sqexp", False, "lin", "sample_opt", "seqbatch", 1),
]


@pytest.mark.parametrize("cfg", COMBINATIONS, ids=[f"cfg{i}" for i in range(len(COMBINATIONS))])
def test_pairwise_combinations_of_the_plugin_trio(api, cfg):
    import boss_jl_amd as B
    from boss_jl_amd.bo import bo
    xy, f, discrete, cons, ymax, model_kind, kernel, dirac, fitness, fitter_kind, am_kind, iters = cfg
    rng = np.random.default_rng(100 + COMBINATIONS.index(cfg))
    X = np.array([[5.0, 10.0, 10.0]]) if xy == "dup" else np.array([[5.0, 10.0, 15.0]])
    Y = np.stack([_obj(X[:, j], rng) for j in range(3)], axis=1)
    P, d = 2, 1
    noise = [B.Dirac(0.1)] * P if dirac else [B.LogNormal(-2.3, 0.3)] * P
    amp = [B.Dirac(1.0), B.LogNormal(0.0, 0.5)] if dirac else [B.LogNormal(0.0, 0.5)] * P
    lsc = [B.MvLogNormal([1.0], [0.5])] * P
    kw = {}
    if model_kind == "gp_mean":
        kw["mean"] = lambda x: np.array([0.1 * x[0], -1.0])
    if model_kind == "semipar":
        kw["parametric"] = lambda x, th: np.array([th[0] * np.cos(2 * x[0]), th[1]])
        kw["theta_priors"] = [B.LogNormal(0.0, 0.3), B.LogNormal(0.0, 0.3)] if not dirac else [B.Dirac(1.0), B.LogNormal(0.0, 0.3)]
    model = B.HipGaussianProcess(lsc, amp, noise, kernel=kernel, **kw)
    domain = B.Domain(([0.0], [20.0]), discrete=[True] if discrete else None, cons=(lambda x: [x[0] - 1.0]) if cons else None)
    fit = B.LinFitness([1.0, 0.0]) if fitness == "lin" else B.NonlinFitness(lambda y: y[0] - 0.1 * y[1] ** 2)
    y_max = [np.inf, 0.0] if ymax == "finite" else [np.inf, np.inf]
    obj = None if f == "missing" else (lambda x: _obj(x, rng))
    problem = B.BossProblem(obj, domain, B.ExpectedImprovement(fit, eps_samples=20), model, B.ExperimentData(X, Y), y_max)
    if model_kind == "semipar" and fitter_kind != "sampling":
        fitter_kind = "sampling"                             # the gradient fitters treat the prior mean as fixed (documented)
    fitter = {"sampling": B.HipBatchedMAP(samples=40, seed=1), "optimization": B.HipGradientMAP(multistart=3, iters=5, seed=1),
              "sample_opt": B.HipSampleOptMAP(samples=30, multistart=2, iters=4, seed=1)}[fitter_kind]
    prior = lambda r: r.uniform(0.0, 20.0, 1)                # noqa: E731
    if am_kind == "optimization" and fitness == "nonlin":
        am_kind = "sampling"                                 # the analytic EI gradient exists for LinFitness only (documented)
    am = {"sampling": B.HipBatchAM(x_prior=prior, samples=200, seed=2),
          "grid": B.HipBatchAM(points=np.linspace(0.0, 20.0, 81)[None, :]),
          "optimization": B.HipGradientAM(x_prior=prior, multistart=20, iters=8, seed=2),
          "seqbatch": B.HipSequentialBatchAM(B.HipBatchAM(x_prior=prior, samples=200, seed=2), batch_size=2)}[am_kind]
    if f == "missing":
        from boss_jl_amd.bo import estimate_parameters
        estimate_parameters(problem, fitter)
        x, _ = am.maximize_acquisition(problem)
        x = np.atleast_2d(np.asarray(x, float).reshape(d, -1))
        assert np.isfinite(x).all() and (x >= 0).all() and (x <= 20).all()
        if discrete:
            assert np.allclose(x, np.round(x))
        if cons:
            assert (x >= 1.0 - 1e-12).all()
        return
    if am_kind == "seqbatch":
        from boss_jl_amd.bo import estimate_parameters
        for _ in range(iters):
            estimate_parameters(problem, fitter)
            Xb, _ = am.maximize_acquisition(problem)
            Xb = np.asarray(Xb, float).reshape(d, -1)
            assert Xb.shape[1] == 2 and (Xb >= 0).all() and (Xb <= 20).all()
            problem.augment_dataset(Xb, np.stack([obj(Xb[:, j]) for j in range(2)], axis=1))
        assert problem.data.X.shape[1] == 3 + 2 * iters
        return
    problem = bo(problem, fitter, am, iters)
    assert problem.data.X.shape[1] == 3 + iters and np.isfinite(problem.data.Y).all()
    new = problem.data.X[:, 3:]
    assert (new >= 0).all() and (new <= 20).all()
    if discrete:
        assert np.allclose(new, np.round(new))
    if cons:
        assert (new >= 1.0 - 1e-12).all()

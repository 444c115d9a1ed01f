"""GPU tests (pytest -m gpu) of boss_gp_update_acq: the first acquisition of a BO iteration riding along the posterior update
(csrc/rider.hpp) — against the CPU oracle, against the two-call path (boss_gp_update + boss_acq_ei) on the same handle, and
through the fallbacks.  Reference: /root/reference/src/bo.jl:30-48 (estimate_parameters! -> maximize_acquisition),
src/acquisition_maximizers/sampling.jl:43-57, src/models/gaussian_process.jl:169-178,199-211,
src/acquisitions/expected_improvement.jl:68-101.

Stated fp64 tolerance (as tests/test_gpu_parity.py): |Δμ| <= 1e-9 (1+|μ|), |Δσ²| <= 1e-9 α², |Δlogpdf| <= 1e-9 (1+|logpdf|),
|Δacq| <= 1e-9 — condition-aware beyond cond(K) = 1e6."""
import os
import subprocess
import sys

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


def make(d, N, M, seed=1, noise=0.05):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + noise * rng.standard_normal(N)
    Xs = np.random.default_rng(seed + 1000).uniform(0, 1, (d, M))
    return X, y, Xs


def tol_for(O, post, N):
    K = O.kernelmatrix(post.h, post.X)
    K[np.diag_indices(N)] += post.h.noise_std ** 2
    c = np.linalg.cond(K)
    return max(1e-9, c * N * 2.0 ** -53 * 8)


# sizes around the block boundaries of the rider: 3 / 4 / 5 / 12 block columns, ragged N and M, one to many candidate strips
@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
@pytest.mark.parametrize("d,N,M", [(3, 300, 70), (8, 512, 64), (2, 513, 1), (8, 1000, 257), (5, 1408, 1024), (8, 1500, 33)])
def test_update_acq_matches_oracle_and_two_call_path(api, O, kernel, d, N, M):
    X, y, Xs = make(d, N, M, seed=N + M)
    lam = np.linspace(0.3, 0.7, d)
    amp, sig = 1.2, 0.05
    best = float(y.max())
    post = O.gp_fit(X, y, kernel, lam, amp, sig)
    mu_o, var_o = O.gp_mean_and_var(post, Xs, clip=False)
    acq_o = O.ei_acquisition([post], Xs, [1.0], None, best)
    tol = tol_for(O, post, N)
    g = api.GP(X, y, kernel)
    cand = api.Candidates(Xs)
    r = g.update_acq(lam, amp, sig, cand, best=best, want_acq=True, want_moments=True)
    assert r["fused"], "three or more block columns on the library's own stream: the substitution rides along"
    assert abs(r["logpdf"] - post.logpdf) <= tol * (1 + abs(post.logpdf))
    assert np.all(np.abs(r["mu"] - mu_o) <= tol * (1 + np.abs(mu_o))), np.abs(r["mu"] - mu_o).max()
    assert np.all(np.abs(r["var"] - var_o) <= tol * amp ** 2), np.abs(r["var"] - var_o).max()
    assert np.all(np.abs(r["acq"] - acq_o) <= tol), np.abs(r["acq"] - acq_o).max()
    assert r["acq"][r["argmax"]] == r["acq"].max() == r["max"] and r["argmax"] == int(np.argmax(r["acq"]))
    assert abs(acq_o[r["argmax"]] - acq_o.max()) <= 2 * tol
    # the two-call path on the same handle
    lp2 = g.update(lam, amp, sig)
    acq2, am2, mx2 = api.acq_ei([[g]], cand, [1.0], None, best)
    assert lp2 == r["logpdf"], "the update itself is the same sequence of kernels with or without a rider"
    assert np.all(np.abs(acq2 - r["acq"]) <= tol)
    # repeated: bit-identical (every sum of the rider has a fixed order)
    r2 = g.update_acq(lam, amp, sig, cand, best=best, want_acq=True, want_moments=True)
    assert r2["fused"] and r2["logpdf"] == r["logpdf"] and r2["argmax"] == r["argmax"] and r2["max"] == r["max"]
    assert np.array_equal(r2["mu"], r["mu"]) and np.array_equal(r2["var"], r["var"]) and np.array_equal(r2["acq"], r["acq"])
    # ... and the handle is an ordinary fitted posterior afterwards
    mu3, var3 = g.predict(Xs)
    assert np.all(np.abs(mu3 - mu_o) <= tol * (1 + np.abs(mu_o)))
    cand.close()
    g.close()


def test_update_acq_semiparametric_constraint_and_mask(api, O):
    """prior means at the observations and at the candidates (src/models/semiparametric.jl:79-92), an upper constraint
    (feas_prob, expected_improvement.jl:113-114), candidates outside the domain masked to 0 (make_safe :58-65), a
    fitness coefficient, no best-so-far (-> feasibility alone)."""
    d, N, M = 6, 2048, 777
    X, y, Xs = make(d, N, M, seed=3)
    th = np.linspace(-0.3, 0.4, d)
    mX, mXs = 0.2 + th @ X, 0.2 + th @ Xs
    lam = np.full(d, 0.45)
    post = O.gp_fit(X, y, "matern52", lam, 0.9, 0.03, mean=mX)
    tol = tol_for(O, post, N)
    mask = np.random.default_rng(0).uniform(size=M) > 0.2
    g = api.GP(X, y, "matern52")
    cand = api.Candidates(Xs)
    for best, ymax, coef in ((float(y.max()), 0.5, 1.0), (None, 0.5, 1.0), (0.3, np.inf, 2.0)):
        acq_o = O.ei_acquisition([post], Xs, [coef], [ymax], best, means_s=[mXs])
        acq_o = np.where(mask, acq_o, 0.0)
        r = g.update_acq(lam, 0.9, 0.03, cand, fit_coef=coef, y_max=ymax, best=best, mean_X=mX, mean_Xs=mXs, valid_mask=mask, want_acq=True)
        assert r["fused"]
        assert abs(r["logpdf"] - post.logpdf) <= tol * (1 + abs(post.logpdf))
        assert np.all(np.abs(r["acq"] - acq_o) <= tol), np.abs(r["acq"] - acq_o).max()
        assert r["argmax"] == int(np.argmax(r["acq"]))
    cand.close()
    g.close()


def test_update_acq_at_bench_size(api, O):
    """BASELINE configs[1]+[2] shapes: N = 4096, d = 8, 1024 and 2048 candidates — against a LAPACK fit on the host."""
    d, N = 8, 4096
    X, y, Xs = make(d, N, 2048, seed=1)
    lam = np.full(d, 0.5)
    post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
    tol = tol_for(O, post, N)
    best = float(y.max())
    g = api.GP(X, y, "matern52")
    for M in (1024, 2048):
        cand = api.Candidates(Xs[:, :M])
        mu_o, var_o = O.gp_mean_and_var(post, Xs[:, :M], clip=False)
        acq_o = O.ei_acquisition([post], Xs[:, :M], [1.0], None, best)
        r = g.update_acq(lam, 1.0, 0.05, cand, best=best, want_acq=True, want_moments=True)
        assert r["fused"]
        assert abs(r["logpdf"] - post.logpdf) <= tol * (1 + abs(post.logpdf))
        assert np.all(np.abs(r["mu"] - mu_o) <= tol * (1 + np.abs(mu_o))), np.abs(r["mu"] - mu_o).max()
        assert np.all(np.abs(r["var"] - var_o) <= tol), np.abs(r["var"] - var_o).max()
        assert np.all(np.abs(r["acq"] - acq_o) <= tol)
        for i in range(5):                                    # a loop of BO iterations: changing hyper-parameters, same candidates
            ri = g.update_acq(lam, 1.0, 0.05 + 1e-3 * i, cand, best=best)
            assert ri["fused"]
        r2 = g.update_acq(lam, 1.0, 0.05, cand, best=best, want_acq=True)
        assert np.array_equal(r2["acq"], r["acq"]) and r2["logpdf"] == r["logpdf"]
        cand.close()
    g.close()


def test_update_acq_not_positive_definite(api):
    """duplicate points without noise: PosDefException analogue, the handle stays unfitted, the next call works"""
    d, N, M = 2, 600, 40
    X, y, Xs = make(d, N, M, seed=9)
    X[:, 300:] = X[:, :300]
    g = api.GP(X, y, "sqexp")
    cand = api.Candidates(Xs)
    with pytest.raises(api.BossError) as e:
        g.update_acq(np.full(d, 0.5), 1.0, 0.0, cand, best=0.0)
    assert e.value.code == api.BOSS_E_NOT_PD
    with pytest.raises(api.BossError) as e:
        g.predict(Xs)
    assert e.value.code == api.BOSS_E_NOT_FITTED
    r = g.update_acq(np.full(d, 0.5), 1.0, 0.1, cand, best=0.0, want_acq=True)
    assert r["fused"] and np.isfinite(r["acq"]).all()
    cand.close()
    g.close()


def test_update_acq_small_and_unfused_shapes(api, O):
    """N <= 128 (one-launch update) and fewer than three block columns: the call runs the two phases one after the other."""
    for d, N, M in ((1, 20, 50), (3, 128, 10), (4, 200, 65)):
        X, y, Xs = make(d, N, M, seed=N)
        lam = np.full(d, 0.4)
        post = O.gp_fit(X, y, "matern32", lam, 1.1, 0.1)
        acq_o = O.ei_acquisition([post], Xs, [1.0], None, float(y.max()))
        g = api.GP(X, y, "matern32")
        cand = api.Candidates(Xs)
        r = g.update_acq(lam, 1.1, 0.1, cand, best=float(y.max()), want_acq=True)
        assert not r["fused"]
        assert abs(r["logpdf"] - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf))
        assert np.all(np.abs(r["acq"] - acq_o) <= 1e-9) and r["argmax"] == int(np.argmax(r["acq"]))
        cand.close()
        g.close()


def test_update_acq_falls_back_with_the_chain():
    """BOSS_TEST_DROP_CHAIN=1: the chain kernel is never launched, the rider's gates give up with everybody else, the update is
    repeated without the chain and the acquisition follows the ordinary way — same numbers, `fused` false.  BOSS_NO_RIDER=1:
    the two phases one after the other by request."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from boss_jl_amd import api
from oracle import gp_oracle as O
rng = np.random.default_rng(5)
d, N, M = 3, 1500, 300
X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0) + 0.05*rng.standard_normal(N)
Xs = rng.uniform(0, 1, (d, M))
lam = np.full(d, 0.4)
post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
acq_o = O.ei_acquisition([post], Xs, [1.0], None, float(y.max()))
g = api.GP(X, y, "matern52"); cand = api.Candidates(Xs)
for i in range(3):
    r = g.update_acq(lam, 1.0, 0.05, cand, best=float(y.max()), want_acq=True)
    print("RES", int(r["fused"]), abs(r["logpdf"] - post.logpdf) / (1 + abs(post.logpdf)), float(np.abs(r["acq"] - acq_o).max()), r["argmax"] == int(np.argmax(acq_o)))
''' % ROOT
    for env in ({"BOSS_TEST_DROP_CHAIN": "1"}, {"BOSS_NO_RIDER": "1"}, {"BOSS_NO_CHAIN": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        lines = [ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("RES")]
        assert r.returncode == 0 and len(lines) == 3, r.stdout + r.stderr
        for fused, e_lp, e_acq, same in lines:
            assert fused == "0" and float(e_lp) <= 1e-9 and float(e_acq) <= 1e-9 and same == "True", (env, lines)


def test_batch_maximizer_takes_the_fused_call(api, O):
    """HipBatchAM on a one-output problem: model_posterior(problem) + acq.(eachcol(xs)) + argmax as ONE device call
    (maximize_acquisition, src/acquisition_maximizers/sampling.jl:20-57) — same point and value as the two-phase path, semiparametric
    mean, constraint function and an upper constraint included."""
    import boss_jl_amd as B
    from boss_jl_amd.model import HipGPParams
    d, N, M = 3, 700, 400
    X, y, Xs = make(d, N, M, seed=11)
    model = B.HipGaussianProcess(lengthscale_priors=[B.MvLogNormal([0.] * d, [1.] * d)], amplitude_priors=[B.LogNormal()],
                                 noise_std_priors=[B.LogNormal()], kernel="matern32", mean=lambda x: [0.1 + 0.2 * x[0]])
    for y_max, cons in ((None, None), ([0.9], lambda x: [x[1] - 0.2])):
        problem = B.BossProblem(f=None, domain=B.Domain(bounds=([0.1] * d, [0.9] * d), cons=cons), y_max=y_max,
                                acquisition=B.ExpectedImprovement(B.LinFitness([1.5])), model=model, data=B.ExperimentData(X, y[None, :]))
        problem.params = HipGPParams(np.full((d, 1), 0.4), [1.1], [0.05])
        problem.consistent = True
        x1, v1 = B.HipBatchAM(points=Xs, fused=True).maximize_acquisition(problem)
        x0, v0 = B.HipBatchAM(points=Xs, fused=False).maximize_acquisition(problem)
        assert np.array_equal(x1, x0) and abs(v1 - v0) <= 1e-9
        _, a1 = B.HipBatchAM(points=Xs, fused=True).maximize_acquisition(problem, return_all=True)
        _, a0 = B.HipBatchAM(points=Xs, fused=False).maximize_acquisition(problem, return_all=True)
        assert np.all(np.abs(a1 - a0) <= 1e-9) and a1.max() == v1

"""boss_gp_fit_batch + the one-launch prediction of a set of posteriors (pytest -m gpu).

The reference builds one posterior per hyper-parameter sample of a Bayesian-inference fit
(/root/reference/src/posterior.jl:15-19; samples from ext/TuringExt.jl:88-107) and averages the acquisition over them
(/root/reference/src/acquisitions/expected_improvement.jl:87-90).  Here the S posteriors of an output come out of ONE batched
factorisation as resident handles, and boss_acq_ei walks all of them in one prediction launch.  Tolerances as in
tests/test_gpu_parity.py: |Δμ| <= 1e-9 (1+|μ|), |Δσ²| <= 1e-9 α², |Δlogpdf| <= 1e-9 (1+|logpdf|), acquisition 1e-10 absolute.
"""
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


def draw(d, S, seed):
    rng = np.random.default_rng(seed)
    return (np.exp(-0.7 + 0.3 * rng.standard_normal((d, S))), np.exp(0.3 * rng.standard_normal(S)),
            np.exp(-3 + 0.3 * rng.standard_normal(S)))


@pytest.mark.parametrize("kernel", ["matern32", "matern52", "sqexp"])
@pytest.mark.parametrize("d,N,S,M", [(1, 3, 4, 5), (2, 20, 7, 33), (3, 128, 3, 64), (3, 129, 5, 70), (8, 300, 6, 257), (5, 700, 4, 100)])
def test_fit_batch_members_match_oracle(api, O, kernel, d, N, S, M):
    """Every member of a batch-fitted set is an ordinary posterior handle: logpdf, moments, and the whole factor against the oracle
    (N <= 128 runs the one-workgroup-per-set kernel, above it the batched blocked factorisation)."""
    X, y, Xs = make(d, N, M, seed=N)
    lam, amp, sig = draw(d, S, N)
    gps, ll, st = api.fit_batch(X, y, kernel, lam, amp, sig)
    assert np.all(st == api.BOSS_OK)
    for s in range(S):
        post = O.gp_fit(X, y, kernel, lam[:, s], amp[s], sig[s])
        assert abs(ll[s] - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf)), s
        mu, var = gps[s].predict(Xs)
        mu_o, var_o = O.gp_mean_and_var(post, Xs)
        assert np.allclose(mu, mu_o, rtol=0, atol=1e-9 * (1 + np.abs(mu_o).max())), s
        assert np.allclose(var, var_o, rtol=0, atol=1e-9 * amp[s] ** 2 + 1e-12), s
        L, z = gps[s].factor()
        assert np.allclose(np.tril(L), post.L, rtol=0, atol=1e-9 * (1 + np.abs(post.L).max())), s
    # the handles are freed in an arbitrary order; the shared storage goes with the last one
    for s in np.random.default_rng(0).permutation(S):
        gps[s].close()


@pytest.mark.parametrize("d,N,S,P,M,with_mean", [(2, 40, 5, 1, 33, False), (3, 300, 5, 2, 70, True), (4, 600, 3, 2, 257, False), (8, 1100, 4, 1, 64, True)])
def test_set_prediction_matches_oracle_and_the_per_sample_loop(api, O, d, N, S, P, M, with_mean):
    """EI x feasibility averaged over S posteriors of P outputs (expected_improvement.jl:77-90) through the one-launch set
    prediction, against the oracle — and bit-identical to the launch-per-sample loop (BOSS_NO_SET_PREDICT=1, a child process)."""
    X, _, Xs = make(d, N, M, seed=7 * N)
    rng = np.random.default_rng(N)
    Y = np.stack([np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N),
                  np.cos(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)])[:P]
    theta = rng.standard_normal((S, P, d + 1)) * 0.1

    def mean_fn(s, p, Z):                                        # a parametric mean that differs per sample (Semiparametric under BI)
        return theta[s, p, 0] + theta[s, p, 1:] @ Z

    gps, posts = [], []
    hyp = [draw(d, S, 100 * p + N) for p in range(P)]
    for p in range(P):
        lam, amp, sig = hyp[p]
        mX = np.stack([mean_fn(s, p, X) for s in range(S)]) if with_mean else None
        g, ll, st = api.fit_batch(X, Y[p], "matern52", lam, amp, sig, mean_X=mX)
        assert np.all(st == api.BOSS_OK)
        gps.append(g)
    for s in range(S):
        posts.append([O.gp_fit(X, Y[p], "matern52", hyp[p][0][:, s], hyp[p][1][s], hyp[p][2][s],
                               mean=mean_fn(s, p, X) if with_mean else None) for p in range(P)])
    handles = [[gps[p][s] for p in range(P)] for s in range(S)]
    coefs = [1.0, 0.0][:P]
    y_max = [np.inf, 0.3][:P]
    best = float(Y[0].max())
    means_s = np.array([[mean_fn(s, p, Xs) for p in range(P)] for s in range(S)]) if with_mean else None
    mask = np.random.default_rng(1).uniform(size=M) > 0.1
    cand = api.Candidates(Xs)
    acq, am, mx = api.acq_ei(handles, cand, coefs, y_max, best, valid_mask=mask, mean_Xs=means_s)
    want = np.zeros(M)
    for s in range(S):
        want += O.ei_acquisition(posts[s], Xs, coefs, y_max, best, means_s=None if means_s is None else means_s[s])
    want = np.where(mask, want / S, 0.0)
    assert np.allclose(acq, want, rtol=0, atol=1e-10)
    assert am == int(np.argmax(acq)) and mx == acq[am]
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from boss_jl_amd import api
d, N, S, P, M, with_mean = %r
Z = np.load(sys.argv[1], allow_pickle=True)
gps = []
for p in range(P):
    g, ll, st = api.fit_batch(Z["X"], Z["Y"][p], "matern52", Z["lam"][p], Z["amp"][p], Z["sig"][p], mean_X=Z["mX"][p] if with_mean else None)
    gps.append(g)
handles = [[gps[p][s] for p in range(P)] for s in range(S)]
acq, am, mx = api.acq_ei(handles, api.Candidates(Z["Xs"]), list(Z["coefs"]), list(Z["ymax"]), float(Z["best"]), valid_mask=Z["mask"],
                         mean_Xs=Z["means_s"] if with_mean else None)
np.save(sys.argv[2], acq)
print("RES", am, repr(mx))
''' % (ROOT, (d, N, S, P, M, with_mean))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), X=X, Y=Y, Xs=Xs, lam=np.stack([h[0] for h in hyp]), amp=np.stack([h[1] for h in hyp]),
                 sig=np.stack([h[2] for h in hyp]), mX=np.stack([np.stack([mean_fn(s, p, X) for s in range(S)]) for p in range(P)]),
                 coefs=np.array(coefs), ymax=np.array(y_max), best=best, mask=mask,
                 means_s=means_s if means_s is not None else np.zeros(1))
        r = subprocess.run([sys.executable, "-c", code, os.path.join(td, "in.npz"), os.path.join(td, "out.npy")],
                           env=dict(os.environ, BOSS_NO_SET_PREDICT="1", BOSS_POISON_ALLOC="1"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "RES" in r.stdout, r.stdout + r.stderr
        loop = np.load(os.path.join(td, "out.npy"))
    # the loop takes the fused kernel for 128 < N with fewer than four 256-row steps — the same kernel body and summation order as
    # the set launch: bit-identical; elsewhere (one-wave-per-candidate kernel for N <= 128, step-by-step path) equal to rounding
    Np = -(-N // 256) * 256
    if N > 128 and Np < 1024:
        assert np.array_equal(loop, acq), float(np.abs(loop - acq).max())
    else:
        assert np.allclose(loop, acq, rtol=0, atol=1e-12), float(np.abs(loop - acq).max())
    for row in gps:
        for g in row:
            g.close()


def test_members_are_ordinary_handles(api, O):
    """A member can be updated in place, appended to (it then leaves the shared storage) and freed while its siblings stay valid;
    a set that is not positive definite comes back unfitted with its status, the others are built.  Run on poisoned allocations
    (NaN patterns in every fresh device block) so that nothing passes on memory the batch kernels never wrote."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from boss_jl_amd import api
from oracle import gp_oracle as O
rng = np.random.default_rng(3)
d, N, S, M = 3, 400, 6, 50
X = rng.uniform(0, 1, (d, N)); X[:, 1] = X[:, 0]                       # a duplicated point: singular without noise
y = np.sin(2*np.pi*X).sum(0) + 0.05*rng.standard_normal(N)
Xs = rng.uniform(0, 1, (d, M))
lam = np.exp(-0.7 + 0.3*rng.standard_normal((d, S))); amp = np.exp(0.3*rng.standard_normal(S)); sig = np.exp(-3 + 0.3*rng.standard_normal(S))
sig[2] = 0.0                                                           # member 2: K + 1e-16 I with two equal rows -> not PD
gps, ll, st = api.fit_batch(X, y, "matern52", lam, amp, sig)
assert st[2] == api.BOSS_E_NOT_PD and ll[2] == -np.inf, (st, ll)
assert all(st[s] == 0 for s in range(S) if s != 2)
try:
    gps[2].predict(Xs); raise SystemExit("predict on the unfitted member did not fail")
except api.BossError as e:
    assert e.code == api.BOSS_E_NOT_FITTED
def check(g, lam_s, amp_s, sig_s, Xd=X, yd=y):
    post = O.gp_fit(Xd, yd, "matern52", lam_s, amp_s, sig_s)
    mu, var = g.predict(Xs); mu_o, var_o = O.gp_mean_and_var(post, Xs)
    assert np.allclose(mu, mu_o, rtol=0, atol=1e-9*(1 + np.abs(mu_o).max())) and np.allclose(var, var_o, rtol=0, atol=1e-9*amp_s**2 + 1e-12)
    return post
for s in (0, 1, 3, 4, 5): check(gps[s], lam[:, s], amp[s], sig[s])
# in-place update of member 1 (and of the unfitted member 2, with a proper noise level): the siblings keep their factors
lp = gps[1].update(lam[:, 0]*1.1, 0.9, 0.07); post = check(gps[1], lam[:, 0]*1.1, 0.9, 0.07); assert abs(lp - post.logpdf) <= 1e-9*(1 + abs(post.logpdf))
lp = gps[2].update(lam[:, 2], amp[2], 0.05); check(gps[2], lam[:, 2], amp[2], 0.05)
for s in (0, 3, 4, 5): check(gps[s], lam[:, s], amp[s], sig[s])
# append to member 3: it detaches from the shared (X, y); members 0, 4 still see the original data
xn = rng.uniform(0, 1, (d, 2)); yn = np.array([0.3, -0.2])
gps[3].append(xn, yn)
check(gps[3], lam[:, 3], amp[3], sig[3], np.concatenate([X, xn], 1), np.concatenate([y, yn]))
for s in (0, 4, 5): check(gps[s], lam[:, s], amp[s], sig[s])
# new observations for member 4 alone (set_y detaches it as well)
y4 = y + 0.1; gps[4].set_y(y4); gps[4].update(lam[:, 4], amp[4], sig[4]); check(gps[4], lam[:, 4], amp[4], sig[4], X, y4)
check(gps[0], lam[:, 0], amp[0], sig[0]); check(gps[5], lam[:, 5], amp[5], sig[5])
for s in (5, 0, 3, 1, 4, 2): gps[s].close()
print("RES ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BOSS_POISON_ALLOC="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RES ok" in r.stdout, r.stdout + r.stderr


def test_config5_all_512_posteriors_resident(api, O):
    """BASELINE.json configs[4] end to end: 512 hyper-parameter samples, N = 1024, d = 8 — ALL 512 posteriors resident out of one
    batched factorisation, the acquisition averaged over all of them at 256 candidates against the oracle (512 LAPACK fits)."""
    d, N, S, M = 8, 1024, 512, 256
    X, y, Xs = make(d, N, M, seed=4)
    lam, amp, sig = draw(d, S, 4)
    gps, ll, st = api.fit_batch(X, y, "matern52", lam, amp, sig)
    assert np.all(st == api.BOSS_OK) and np.isfinite(ll).all()
    ll_b, st_b = api.loglike_batch(X, y, "matern52", lam, amp, sig)
    assert np.all(st_b == api.BOSS_OK) and np.allclose(ll, ll_b, rtol=1e-12, atol=0)      # the same batched schedule keeps / drops the handles
    b = float(y.max())
    cand = api.Candidates(Xs)
    acq, am, mx = api.acq_ei([[g] for g in gps], cand, [1.0], [np.inf], b)
    want = np.zeros(M)
    mu_avg = np.zeros(M)
    for s in range(S):
        post = O.gp_fit(X, y, "matern52", lam[:, s], amp[s], sig[s])
        assert abs(ll[s] - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf)), s
        want += O.ei_acquisition([[post]], Xs, [1.0], [np.inf], b)
        mu_avg += O.gp_mean_and_var(post, Xs)[0]
    want /= S
    assert np.allclose(acq, want, rtol=0, atol=1e-10), float(np.abs(acq - want).max())
    assert am == int(np.argmax(acq)) and mx == acq[am]
    # average_mean (src/posterior.jl:177-179) over a spread of the resident members through their own handles
    sub = list(range(0, S, 37))
    mus = np.mean([gps[s].predict(Xs)[0] for s in sub], axis=0)
    want_mu = np.mean([O.gp_mean_and_var(O.gp_fit(X, y, "matern52", lam[:, s], amp[s], sig[s]), Xs)[0] for s in sub], axis=0)
    assert np.allclose(mus, want_mu, rtol=0, atol=1e-9 * (1 + np.abs(want_mu).max()))
    for g in gps:
        g.close()


def test_released_storage_is_reused_without_leaking_into_the_next_set(api, O):
    """The storage of a set whose last member is gone stays with the device context for the next boss_gp_fit_batch (a BI fitter refits
    its samples every iteration).  A larger set, then a smaller one in the block the larger one left behind, then a larger one again
    (the cached block is too small: fresh storage), with a member kept alive across the hand-over: every posterior agrees with the
    oracle, i.e. nothing of the previous tenant's factors, inverses or scalars is read."""
    shapes = [(4, 700, 6), (3, 300, 3), (2, 140, 5), (4, 900, 7), (3, 300, 3)]
    keep = None
    for rnd, (d, N, S) in enumerate(shapes):
        X, y, Xs = make(d, N, 40, seed=100 + rnd)
        lam, amp, sig = draw(d, S, 200 + rnd)
        gps, ll, st = api.fit_batch(X, y, "matern52", lam, amp, sig)
        assert np.all(st == api.BOSS_OK)
        for s in range(S):
            post = O.gp_fit(X, y, "matern52", lam[:, s], amp[s], sig[s])
            assert abs(ll[s] - post.logpdf) <= 1e-9 * (1 + abs(post.logpdf)), (rnd, s)
            mu, var = gps[s].predict(Xs)
            mu_o, var_o = O.gp_mean_and_var(post, Xs)
            assert np.allclose(mu, mu_o, rtol=0, atol=1e-9 * (1 + np.abs(mu_o).max())), (rnd, s)
            assert np.allclose(var, var_o, rtol=0, atol=1e-9 * amp[s] ** 2), (rnd, s)
        if keep is not None:
            # the survivor of the previous round still describes ITS data (its set's block cannot have been handed on)
            g_old, post_old, Xs_old = keep
            mu, var = g_old.predict(Xs_old)
            mu_o, var_o = O.gp_mean_and_var(post_old, Xs_old)
            assert np.allclose(mu, mu_o, rtol=0, atol=1e-9 * (1 + np.abs(mu_o).max())) and np.allclose(var, var_o, rtol=0, atol=1e-9 * 4), rnd
            g_old.close()
        keep = (gps[0], O.gp_fit(X, y, "matern52", lam[:, 0], amp[0], sig[0]), Xs)
        for g in gps[1:]:
            g.close()
    keep[0].close()


def test_parked_set_storage_is_given_back_when_an_allocation_fails(api):
    """The storage of a released set stays with its context for the next fit of that size (hipMalloc / hipFree of gigabytes cost up to
    half a second on some boxes).  It must not stand in the way of anything else: an allocation that fails drops it and tries again
    (dev_malloc, csrc/bosship.hip) — here with a simulated failure, so that the test does not have to fill 288 GB."""
    import ctypes as C
    lib = api.load_library()
    rng = np.random.default_rng(0)
    d, N, S = 4, 700, 6
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(2 * np.pi * X).sum(0)
    gps, ll, st = api.fit_batch(X, y, "matern52", np.full((d, S), 0.4), np.ones(S), np.full(S, 0.1))
    assert (st == 0).all()
    for g in gps:
        g.close()
    nbytes = C.c_size_t(0)
    lib.boss_debug_slab_cache_bytes.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
    lib.boss_debug_slab_cache_bytes(0, C.byref(nbytes))
    if nbytes.value == 0:
        pytest.skip("slab cache off (BOSS_SLAB_CACHE=0 / BOSS_POISON_ALLOC=1)")
    lib.boss_debug_fail_next_alloc(1)
    g = api.GP(X, y, "matern52")                                  # its first allocation "fails": the parked block goes, the retry succeeds
    lib.boss_debug_fail_next_alloc(0)
    lib.boss_debug_slab_cache_bytes(0, C.byref(nbytes))
    assert nbytes.value == 0
    lp = g.update(np.full(d, 0.4), 1.0, 0.1)
    assert abs(lp - ll[0]) <= 1e-9 * (1 + abs(ll[0]))
    g.close()
    lib.boss_debug_fail_next_alloc(1)                             # nothing parked any more: the failure surfaces as BOSS_E_ALLOC
    with pytest.raises(api.BossError) as e:
        api.GP(X, y, "matern52")
    lib.boss_debug_fail_next_alloc(0)
    assert e.value.code == api.BOSS_E_ALLOC
    g = api.GP(X, y, "matern52")
    g.close()

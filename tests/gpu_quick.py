"""Quick GPU sanity + timing script (not a pytest file): python tests/gpu_quick.py [stage...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boss_jl_amd import api  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def problem(d, N, M, seed=1, noise=0.05):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + noise * rng.standard_normal(N)
    Xs = np.random.default_rng(seed + 1).uniform(0, 1, (d, M))
    return X, y, Xs


def parity(d, N, M, kernel="matern52", noise=0.05):
    X, y, Xs = problem(d, N, M, noise=noise)
    lam = np.full(d, 0.5)
    t = time.time()
    post = O.gp_fit(X, y, kernel, lam, 1.0, noise)
    mu_o, var_o = O.gp_mean_and_var(post, Xs, clip=False)
    t_cpu = time.time() - t
    g = api.GP(X, y, kernel)
    lp = g.update(lam, 1.0, noise)
    L, z = g.factor()
    mu, var = g.predict(Xs)
    eL = np.abs(L - post.L).max()
    print(f"d={d} N={N} M={M} {kernel}: logpdf gpu={lp:.10f} cpu={post.logpdf:.10f} "
          f"|dL|={eL:.2e} |dmu|={np.abs(mu - mu_o).max():.2e} |dvar|={np.abs(var - var_o).max():.2e} cpu_s={t_cpu:.2f}",
          flush=True)
    g.close()


def timing(d=8, N=4096, M=8192):
    X, y, Xs = problem(d, N, M)
    lam = np.full(d, 0.5)
    g = api.GP(X, y, "matern52")
    g.update(lam, 1.0, 0.05)
    for _ in range(2):
        g.update(lam, 1.0, 0.05)
    t = time.time()
    K = 10
    for _ in range(K):
        g.update(lam, 1.0, 0.05)
    dt = (time.time() - t) / K
    print(f"update N={N}: {dt * 1e3:.3f} ms  ({1 / dt:.1f} updates/s)", flush=True)
    api.prof_enable(0, True)
    api.prof_reset(0)
    g.update(lam, 1.0, 0.05)
    for k in ("prep", "gram", "potrf_diag", "potrf_trsm", "potrf_syrk", "logdet"):
        ms, n = api.prof_get(0, k)
        print(f"   {k:12s} {ms:8.3f} ms over {n} launches", flush=True)
    api.prof_enable(0, False)
    cand = api.Candidates(Xs)
    b = float(y.max())
    api.acq_ei([[g]], cand, [1.0], None, b, want_acq=False)
    t = time.time()
    for _ in range(3):
        acq, am, mx = api.acq_ei([[g]], cand, [1.0], None, b, want_acq=False)
    dt = (time.time() - t) / 3
    print(f"acq M={M}: {dt * 1e3:.3f} ms ({M / dt / 1e6:.3f} M evals/s) argmax={am} max={mx:.6g}", flush=True)
    api.prof_enable(0, True)
    api.prof_reset(0)
    api.acq_ei([[g]], cand, [1.0], None, b, want_acq=False)
    for k in ("dinv", "predict", "ei", "argmax"):
        ms, n = api.prof_get(0, k)
        print(f"   {k:12s} {ms:8.3f} ms over {n} launches", flush=True)
    api.prof_enable(0, False)


def append_timing(d=8, N=4096, steps=12):
    """SequentialBatchAM-style speculative loop: one point appended per step (block Cholesky
    append) against a fresh factorisation of the same augmented data."""
    X, y, Xs = problem(d, N + steps, 256)
    lam = np.full(d, 0.5)
    g = api.GP(X[:, :N], y[:N], "matern52")
    g.update(lam, 1.0, 0.05)
    ts = []
    for i in range(steps):
        t = time.time()
        lp = g.append(X[:, N + i], y[N + i])
        ts.append(time.time() - t)
    mu, var = g.predict(Xs)
    g2 = api.GP(X, y, "matern52")
    g2.update(lam, 1.0, 0.05)
    t = time.time()
    lp2 = g2.update(lam, 1.0, 0.05)
    t_full = time.time() - t
    mu2, var2 = g2.predict(Xs)
    print(f"append N={N}+1 x{steps}: first (grows storage) {ts[0]*1e3:.3f} ms, then {np.median(ts[1:])*1e3:.3f} ms/append "
          f"vs full refit {t_full*1e3:.3f} ms;  |dlogpdf|={abs(lp-lp2):.2e} |dmu|={np.abs(mu-mu2).max():.2e} "
          f"|dvar|={np.abs(var-var2).max():.2e}", flush=True)
    t = time.time()
    lp = g.append(X[:, :64] + 1e-3, y[:64])
    print(f"append 64 points at once: {(time.time()-t)*1e3:.3f} ms", flush=True)


def batch_cfg(S=512, N=1024, d=8):
    """BASELINE config 5: S hyper-parameter sets, each its own N=1024 Cholesky."""
    X, y, _ = problem(d, N, 1, seed=4)
    rng = np.random.default_rng(4)
    lam = np.exp(-0.7 + 0.3 * rng.standard_normal((d, S)))
    amp = np.exp(0.3 * rng.standard_normal(S))
    sig = np.exp(-3 + 0.3 * rng.standard_normal(S))
    api.loglike_batch(X, y, "matern52", lam[:, :8], amp[:8], sig[:8])
    t = time.time()
    ll, st = api.loglike_batch(X, y, "matern52", lam, amp, sig)
    t1 = time.time() - t
    t = time.time()
    ll, st = api.loglike_batch(X, y, "matern52", lam, amp, sig)
    dt = time.time() - t
    want = [O.gp_data_loglike_slice(X, y, "matern52", lam[:, s], amp[s], sig[s]) for s in (0, 7, S - 1)]
    err = max(abs(ll[s] - w) / (1 + abs(w)) for s, w in zip((0, 7, S - 1), want))
    fl = S * (N ** 3 / 3)
    print(f"loglike_batch S={S} N={N}: first {t1*1e3:.1f} ms, steady {dt*1e3:.1f} ms  ({S/dt:.0f} factorizations/s, {fl/dt/1e12:.2f} TF)  rel err {err:.1e} ok={int((st==0).sum())}", flush=True)


def grad_timing(d=8, N=4096, M=8192):
    """Posterior moments + analytic gradients for 8192 candidates (forward + adjoint substitution)."""
    X, y, Xs = problem(d, N, M)
    lam = np.full(d, 0.5)
    g = api.GP(X, y, "matern52")
    g.update(lam, 1.0, 0.05)
    g.predict_grad(Xs)
    t = time.time()
    for _ in range(3):
        mu, var, dmu, dvar = g.predict_grad(Xs)
    dt = (time.time() - t) / 3
    t = time.time()
    for _ in range(3):
        g.predict(Xs)
    dp = (time.time() - t) / 3
    post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
    mu_o, var_o, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs[:, :64])
    print(f"predict_grad N={N} M={M}: {dt*1e3:.2f} ms ({M/dt/1e6:.2f} M gradient evals/s) vs predict {dp*1e3:.2f} ms;  "
          f"|d dmu|={np.abs(dmu[:, :64]-dmu_o).max():.1e} |d dvar|={np.abs(dvar[:, :64]-dvar_o).max():.1e}", flush=True)


def seqbatch_timing(d=8, N=4096, M=8192, nb=8):
    """SequentialBatchAM's speculative loop at the BASELINE size: resident posterior + block Cholesky
    appends, with and without tracked candidates, against the reference's pattern (re-factorise +
    re-predict per selection)."""
    import boss_jl_amd as B
    X, y, Xs = problem(d, N, M)
    prm = B.HipGPParams(np.full((d, 1), 0.5), [1.0], [0.05])
    model = B.HipGaussianProcess([None], [None], [None])
    prob = B.BossProblem(None, B.Domain((np.zeros(d), np.ones(d))), B.ExpectedImprovement(B.LinFitness([1.0])), model,
                         B.ExperimentData(X, y[None, :]), None, prm)
    res = {}
    for name, am in (("tracked", B.HipBatchAM(points=Xs)), ("append only", B.HipBatchAM(points=Xs, shard="candidates"))):
        sb = B.HipSequentialBatchAM(am, nb)
        if name == "append only":
            am.seed = None
            pts = am.points
            am.points = None
            am.candidates = lambda problem, _p=pts: _p            # hides the fixed set from the tracker
        sb.maximize_acquisition(prob)
        sb1 = B.HipSequentialBatchAM(am, 1)
        t = time.time()
        sb1.maximize_acquisition(prob)
        t1 = time.time() - t                                     # set-up (handles, factorisation, tracks) + 1 selection
        t = time.time()
        Xb, _ = sb.maximize_acquisition(prob)
        res[name] = ((time.time() - t - t1) / (nb - 1), Xb, t1)
    # reference pattern: fresh factorisation + prediction per selection
    g = api.GP(X, y, "matern52")
    cand = api.Candidates(Xs)
    t = time.time()
    for _ in range(nb):
        g.update(np.full(d, 0.5), 1.0, 0.05)
        api.acq_ei([[g]], cand, [1.0], None, float(y.max()), want_acq=False)
    t_ref = (time.time() - t) / nb
    same = np.array_equal(res["tracked"][1], res["append only"][1])
    print(f"sequential batch N={N} M={M}: tracked {res['tracked'][0]*1e3:.2f} ms per further selection (set-up + first "
          f"{res['tracked'][2]*1e3:.1f} ms), append only {res['append only'][0]*1e3:.2f} ms (set-up + first "
          f"{res['append only'][2]*1e3:.1f} ms), re-factorise + re-predict {t_ref*1e3:.2f} ms;  same batch: {same}", flush=True)


def llgrad_timing(d=8, N=4096):
    """log-likelihood + its hyper-parameter gradient at the BASELINE size."""
    X, y, _ = problem(d, N, 1)
    lam = np.full(d, 0.5)
    g = api.GP(X, y, "matern52")
    g.update(lam, 1.0, 0.05)
    g.loglike_grad()
    t = time.time()
    for i in range(5):
        g.update(lam, 1.0, 0.05 + 1e-4 * i)
        lp, grad = g.loglike_grad()
    dt = (time.time() - t) / 5
    t = time.time()
    for i in range(5):
        g.update(lam, 1.0, 0.05 + 1e-4 * i)
    du = (time.time() - t) / 5
    _, grad_o = O.gp_data_loglike_grad(X[:, :512], y[:512], "matern52", lam, 1.0, 0.05)
    g2 = api.GP(X[:, :512], y[:512], "matern52")
    g2.update(lam, 1.0, 0.05)
    _, g512 = g2.loglike_grad()
    print(f"loglike + gradient N={N}: {dt*1e3:.2f} ms per (update + gradient) vs update alone {du*1e3:.2f} ms;  "
          f"rel err at N=512: {np.abs(g512-grad_o).max()/(1+np.abs(grad_o).max()):.1e}", flush=True)


def append_fast_timing(d=8, N=4000, count=40):
    """Single-observation appends: block path (first), inverse build (second), rank-one path (from then on)."""
    X, y, _ = problem(d, N + count, 1)
    g = api.GP(X[:, :N], y[:N], "matern52")
    g.reserve(N + count)
    g.update(np.full(d, 0.5), 1.0, 0.05)
    ts = []
    for i in range(count):
        t = time.time()
        g.append(X[:, N + i], y[N + i])
        ts.append(time.time() - t)
    print(f"append N={N}: first {ts[0]*1e3:.3f} ms, second (builds the inverses) {ts[1]*1e3:.3f} ms, "
          f"then {np.median(ts[2:])*1e3:.3f} ms median ({min(ts[2:])*1e3:.3f} min)", flush=True)
    g.close()


def llgrad_batch_timing(d=8, N=4096):
    X, y, _ = problem(d, N, 1)
    rng = np.random.default_rng(0)
    for S in (1, 4, 8, 16):
        lam = rng.uniform(0.4, 0.6, (d, S)); amp = rng.uniform(0.9, 1.1, S); sig = np.full(S, 0.05)
        api.loglike_batch(X, y, "matern52", lam, amp, sig, want_grad=True)
        t = time.time()
        for _ in range(3):
            api.loglike_batch(X, y, "matern52", lam, amp, sig, want_grad=True)
        dt = (time.time() - t) / 3
        print(f"batched loglike + gradient N={N} S={S}: {dt*1e3:.2f} ms per call = {dt/S*1e3:.2f} ms per set ({S/dt:.0f} /s)", flush=True)


def few_timing(d=8, N=4096):
    """The reference's call pattern: one candidate per call."""
    X, y, Xs = problem(d, N, 64)
    g = api.GP(X, y, "matern52")
    g.update(np.full(d, 0.5), 1.0, 0.05)
    X, y, Xs = problem(d, N, 8300)
    for M in (1, 4):                                         # steady state of the explicit-inverse path
        for i in range(3):
            g.predict(Xs[:, i:i + M])
        t = time.time()
        for i in range(300):
            g.predict(Xs[:, i:i + M])
        dt = (time.time() - t) / 300
        api.prof_enable(0, True)
        api.prof_reset(0)
        g.predict(Xs[:, :M])
        print(f"steady-state predict M={M}: {dt*1e3:.3f} ms per call ({M/dt:.0f} evals/s); device 'predict' scope "
              f"{api.prof_get(0, 'predict')[0]*1e3:.1f} us", flush=True)
        api.prof_enable(0, False)
    g.update(np.full(d, 0.5), 1.0, 0.05)
    for M in (1, 32, 224, 1024, 2048, 4096, 8192):
        g.predict(Xs[:, :M])
        t = time.time()
        for i in range(20):
            g.predict(Xs[:, i:i + M])
        dt = (time.time() - t) / 20
        tg = 0.0
        if M <= 4096:
            g.predict_grad(Xs[:, :M])
            t = time.time()
            for i in range(5):
                g.predict_grad(Xs[:, i:i + M])
            tg = (time.time() - t) / 5
        print(f"predict M={M} N={N}: {dt*1e3:.3f} ms per call ({M/dt:.0f} evals/s in the reference's call pattern); "
              f"with gradients {tg*1e3:.3f} ms", flush=True)


def batch_big(N=4096, d=8):
    """Batched posterior updates at the BASELINE size: S hyper-parameter sets on the same data."""
    X, y, _ = problem(d, N, 1)
    rng = np.random.default_rng(5)
    for S in (8, 32):
        lam = np.exp(-0.7 + 0.1 * rng.standard_normal((d, S)))
        amp = np.exp(0.1 * rng.standard_normal(S))
        sig = np.full(S, 0.05)
        api.loglike_batch(X, y, "matern52", lam, amp, sig)
        t = time.time()
        ll, st = api.loglike_batch(X, y, "matern52", lam, amp, sig)
        dt = time.time() - t
        want = O.gp_data_loglike_slice(X, y, "matern52", lam[:, S - 1], amp[S - 1], sig[S - 1])
        print(f"loglike_batch S={S} N={N}: {dt*1e3:.1f} ms  ({S/dt:.0f} updates/s, {S*N**3/3/dt/1e12:.1f} TF)  rel err {abs(ll[S-1]-want)/(1+abs(want)):.1e} ok={int((st==0).sum())}", flush=True)
        api.prof_enable(0, True)
        api.prof_reset(0)
        api.loglike_batch(X, y, "matern52", lam, amp, sig)
        for k in ("gram", "potrf_diag", "potrf_trsm", "potrf_syrk", "logdet"):
            ms, n = api.prof_get(0, k)
            print(f"   {k:12s} {ms:8.3f} ms over {n} launches", flush=True)
        api.prof_enable(0, False)


def multi_output(N=2048, d=6, M=8192):
    """BASELINE config 4: 2 constrained outputs, parametric mean, N=2048 d=6."""
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), np.cos(2 * X).sum(0) - 1.0]) + 0.05 * rng.standard_normal((2, N))
    Xs = rng.uniform(0, 1, (d, M))
    th = np.array([0.1, 0.2])
    mX = [th[0] + th[1] * X.sum(0), th[0] - th[1] * X.sum(0)]
    ms = np.stack([th[0] + th[1] * Xs.sum(0), th[0] - th[1] * Xs.sum(0)])[None]
    lam = np.full(d, 0.5)
    gps = [api.GP(X, Y[p], "matern52") for p in range(2)]
    cand = api.Candidates(Xs)
    for _ in range(2):
        t = time.time()
        for p in range(2):
            gps[p].update(lam, 1.0, 0.05, mean_X=mX[p])
        t_fit = time.time() - t
        t = time.time()
        acq, am, mx = api.acq_ei([gps], cand, [1.0, 0.0], [np.inf, 0.5], float(Y[0].max()), None, ms)
        t_acq = time.time() - t
    posts = [O.gp_fit(X, Y[p], "matern52", lam, 1.0, 0.05, mean=mX[p]) for p in range(2)]
    want = O.ei_acquisition(posts, Xs[:, :256], [1.0, 0.0], [np.inf, 0.5], float(Y[0].max()), means_s=[ms[0, 0, :256], ms[0, 1, :256]])
    print(f"multi-output N={N} P=2 M={M}: fit {t_fit*1e3:.2f} ms, acq {t_acq*1e3:.2f} ms ({M/t_acq/1e6:.2f} M evals/s)  |dacq|={np.abs(acq[:256]-want).max():.1e}", flush=True)


def grad_problem(d, n, seed=3):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (d, n))
    w = np.linspace(1.0, 2.0, d)[:, None]
    y = np.sin(2 * np.pi * w * X).sum(0) / np.sqrt(d)
    dY = 2 * np.pi * w * np.cos(2 * np.pi * w * X) / np.sqrt(d)
    return X, y, dY


def ggp_parity(d, n, M, kernel, dup=False):
    X, y, dY = grad_problem(d, n)
    if dup:
        X[:, 1] = X[:, 0]
    Xs = np.random.default_rng(5).uniform(0, 1, (d, M))
    Xs[:, 0] = X[:, 2]                                   # a candidate on a training point: the perturbed derivative entries
    lam = np.full(d, 0.4)
    t = time.time()
    post = O.gradient_gp_fit(X, y, dY, kernel, lam, 1.2, 0.05, 0.1)
    mu_o, var_o = O.gradient_gp_mean_and_var(post, Xs)
    t_cpu = time.time() - t
    g = api.GradGP(X, y, dY, kernel)
    lp = g.update(lam, 1.2, 0.05, 0.1)
    L, z = g.factor()
    mu, var = g.predict(Xs)
    print(f"ggp d={d} n={n} N={g.N} M={M} {kernel}: logpdf gpu={lp:.10f} cpu={post.logpdf:.10f} |dL|={np.abs(L - post.L).max():.2e} "
          f"|dmu|={np.abs(mu - mu_o).max():.2e} |dvar|={np.abs(var - var_o).max():.2e} cpu_s={t_cpu:.1f}", flush=True)
    g.close()


def ggp_timing(d=8, n=4096, M=8192):
    X, y, dY = grad_problem(d, n)
    Xs = np.random.default_rng(5).uniform(0, 1, (d, M))
    t = time.time()
    g = api.GradGP(X, y, dY, "matern52")
    print(f"ggp create n={n} N={g.N}: {time.time()-t:.2f} s", flush=True)
    lam = np.full(d, 0.4)
    for it in range(3):
        t = time.time()
        lp = g.update(lam, 1.2, 1e-3, 1e-2)
        t_up = time.time() - t
        print(f"  update {t_up*1e3:.1f} ms ({g.N**3/3/t_up/1e12:.1f} TF)  logpdf {lp:.6e}", flush=True)
    for it in range(2):
        t = time.time()
        mu, var = g.predict(Xs)
        t_pr = time.time() - t
        print(f"  predict M={M}: {t_pr*1e3:.1f} ms ({g.N**2*M/t_pr/1e12:.1f} TF)", flush=True)
    mu_t, var_t = g.predict(X[:, :64])
    print(f"  interpolation at training points: max|mu-y|={np.abs(mu_t - y[:64]).max():.2e}  max var={var_t.max():.2e}", flush=True)
    t = time.time()
    mu1, var1 = g.predict(Xs[:, :1])
    print(f"  single candidate: {(time.time()-t)*1e3:.1f} ms  |dmu|={abs(mu1[0]-mu[0]):.1e}", flush=True)
    g.close()


def ngp_timing(d=8, N=4096, M=8192):
    X, y, Xs = problem(d, N, M)
    lamX = 0.25 + 0.5 * X ** 2
    ampX = 1.0 + 0.4 * np.sin(3 * X[0])
    noiX = 0.03 + 0.05 * X[-1] ** 2
    lamS = 0.25 + 0.5 * Xs ** 2
    ampS = 1.0 + 0.4 * np.sin(3 * Xs[0])
    g = api.GibbsGP(X, y)
    g.update(lamX, ampX, noiX)
    g.predict(Xs, lamS, ampS)
    t = time.time()
    for _ in range(10):
        g.update(lamX, ampX, noiX)
    t_up = (time.time() - t) / 10
    t = time.time()
    for _ in range(10):
        g.predict(Xs, lamS, ampS)
    t_pr = (time.time() - t) / 10
    t = time.time()
    for i in range(20):
        g.predict(Xs[:, i:i + 1], lamS[:, i:i + 1], ampS[i:i + 1])
    t_one = (time.time() - t) / 20
    api.prof_enable(0, True)
    api.prof_reset(0)
    g.update(lamX, ampX, noiX)
    g.predict(Xs, lamS, ampS)
    print(f"nonstationary N={N} d={d}: update {t_up*1e3:.2f} ms, predict({M}) {t_pr*1e3:.2f} ms, one candidate {t_one*1e3:.2f} ms; "
          f"gram {api.prof_get(0, 'gram')[0]:.3f} ms, predict scope {api.prof_get(0, 'predict')[0]:.3f} ms", flush=True)
    api.prof_enable(0, False)
    g.close()


if __name__ == "__main__":
    stages = sys.argv[1:] or ["mfma", "parity", "timing"]
    if "mfma" in stages:
        print("mfma f64 TFLOP/s:", api.bench_mfma_f64(0, 20000), flush=True)
    if "parity" in stages:
        parity(1, 3, 4, "matern32", 0.1)
        parity(2, 100, 33)
        parity(8, 300, 70, "sqexp")
        parity(8, 1000, 257)
        parity(8, 1000, 257, noise=1e-3)
    if "ggp" in stages:
        for kern in ("matern52", "sqexp", "matern32"):
            ggp_parity(3, 40, 70, kern)
        ggp_parity(2, 30, 10, "matern52", dup=False)
        ggp_parity(8, 150, 40, "matern52")
        ggp_parity(8, 150, 40, "sqexp", dup=True)
    if "ggp_big" in stages:
        ggp_timing(n=int(os.environ.get("GGP_N", "1024")))
    if "append_fast" in stages:
        append_fast_timing()
    if "llgrad_batch" in stages:
        llgrad_batch_timing()
    if "ngp" in stages:
        ngp_timing()
    if "parity_big" in stages:
        parity(8, 4096, 512)
    if "timing" in stages:
        timing()
    if "append" in stages:
        append_timing()
    if "batch" in stages:
        batch_cfg()
    if "few" in stages:
        few_timing()
    if "llgrad" in stages:
        llgrad_timing()
    if "seqbatch" in stages:
        seqbatch_timing()
    if "grad" in stages:
        grad_timing()
    if "batch_big" in stages:
        batch_big()
    if "multi" in stages:
        multi_output()

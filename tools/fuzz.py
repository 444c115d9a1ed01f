"""One-off randomized stress run (not part of the test suite): random shapes and random sequences of
update / predict (all call sizes) / predict_grad / predict_cov / append / loglike_grad on one handle, every result
checked against the CPU oracle.  python tools/fuzz.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
from oracle import gp_oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
KERN = ["matern32", "matern52", "sqexp"]
NS = [1, 2, 17, 127, 128, 129, 255, 256, 257, 300, 511, 513, 700, 1023, 1024, 1025, 1100, 1536, 2047, 2300]
MS = [1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 100, 224, 500, 1100, 1537]
worst = 0.0
nfused = 0
t_start = time.time()
for case in range(first, ncases):
    rng = np.random.default_rng(seed0 * 1000 + case)
    d = int(rng.choice([1, 2, 3, 8, 17, 33]))
    N = int(rng.choice(NS))
    kern = KERN[int(rng.integers(3))]
    disc = None
    if rng.random() < 0.25 and d > 1:
        disc = rng.random(d) < 0.4
        if not disc.any():
            disc = None
    scale = 6.0 if disc is not None else 1.0
    X = rng.uniform(0, scale, (d, N))
    y = np.sin(2 * np.pi * X / scale).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    use_mean = rng.random() < 0.5
    mfun = (lambda Z: 0.2 - 0.1 * Z[0]) if use_mean else (lambda Z: None)
    g = api.GP(X, y, kern, disc)
    if rng.random() < 0.5:
        g.reserve(N + 40)
    lam = rng.uniform(0.3, 0.9, d) * scale
    amp, sig = float(rng.uniform(0.7, 1.5)), float(rng.uniform(0.03, 0.1))
    lp = g.update(lam, amp, sig, mfun(X))
    post = O.gp_fit(X, y, kern, lam, amp, sig, mean=mfun(X), discrete=disc)
    K = post.L @ post.L.T
    tol = max(1e-9, np.linalg.cond(K) * max(N, 2) * 2.0 ** -53 * 16)
    assert abs(lp - post.logpdf) <= tol * (1 + abs(post.logpdf)), ("logpdf", case)
    ops = []
    M = 0
    track = tcand = tXs = None
    for step in range(int(rng.integers(4, 10))):
        op = rng.choice(["predict", "predict", "predict", "grad", "cov", "append", "append", "update", "llgrad", "track", "acq", "acqgrad",
                         "batch", "update_acq", "update_acq"])
        ops.append(op)
        try:
            if op == "predict":
                M = int(rng.choice(MS))
                Xs = rng.uniform(-0.05 * scale, 1.05 * scale, (d, M))
                mu, var = g.predict(Xs, mfun(Xs))
                mu_o, var_o = O.gp_mean_and_var(post, Xs, mfun(Xs))
                e = max(np.abs(mu - mu_o).max() / (1 + np.abs(mu_o).max()), np.abs(var - var_o).max() / amp ** 2)
            elif op == "grad":
                M = int(rng.choice([1, 3, 20, 65, 224]))
                Xs = rng.uniform(0, scale, (d, M))
                mg = np.vstack([np.full(M, -0.1), np.zeros((d - 1, M))]) if use_mean else None
                mu, var, dmu, dvar = g.predict_grad(Xs, mfun(Xs), mg)
                mu_o, var_o, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs, mfun(Xs), mg)
                e = max(np.abs(mu - mu_o).max() / (1 + np.abs(mu_o).max()), np.abs(dmu - dmu_o).max() / (1 + np.abs(dmu_o).max()),
                        np.abs(dvar - dvar_o).max() / (1 + np.abs(dvar_o).max())) / 10
            elif op == "cov":
                M = int(rng.choice([1, 2, 7, 33, 70]))
                Xs = rng.uniform(0, scale, (d, M))
                mu, cov = g.predict_cov(Xs, mfun(Xs))
                mu_o, cov_o = O.gp_mean_and_cov(post, Xs, mfun(Xs))
                e = max(np.abs(mu - mu_o).max() / (1 + np.abs(mu_o).max()), np.abs(cov - cov_o).max() / amp ** 2)
                if e > tol:
                    print("cov detail: M", M, "mu err", np.abs(mu - mu_o).max(), "cov err", np.abs(cov - cov_o).max(),
                          "diag err", np.abs(np.diag(cov) - np.diag(cov_o)).max(), "few_calls so far", ops, flush=True)
                    mu2, var2 = g.predict(Xs, mfun(Xs))
                    print("  predict on same Xs: mu err", np.abs(mu2 - mu_o).max(), "var err", np.abs(var2 - np.diag(cov_o)).max(), flush=True)
            elif op == "append":
                n = int(rng.choice([1, 1, 1, 2, 5, 33]))
                Xn = rng.uniform(0, scale, (d, n))
                yn = np.sin(2 * np.pi * Xn / scale).sum(0) / np.sqrt(d)
                lp = g.append(Xn, yn, mfun(Xn))
                X, y = np.hstack([X, Xn]), np.concatenate([y, yn])
                post = O.gp_fit(X, y, kern, lam, amp, sig, mean=mfun(X), discrete=disc)
                e = abs(lp - post.logpdf) / (1 + abs(post.logpdf))
            elif op == "update":
                lam = rng.uniform(0.3, 0.9, d) * scale
                amp, sig = float(rng.uniform(0.7, 1.5)), float(rng.uniform(0.03, 0.1))
                lp = g.update(lam, amp, sig, mfun(X))
                post = O.gp_fit(X, y, kern, lam, amp, sig, mean=mfun(X), discrete=disc)
                e = abs(lp - post.logpdf) / (1 + abs(post.logpdf))
                if track is not None:                      # a track belongs to one set of hyper-parameters
                    track.close()
                    track = None
            elif op == "update_acq":
                # boss_gp_update_acq: new hyper-parameters and the first acquisition in one call (rides along the
                # factorisation when the handle and candidate count allow it, two enqueued steps otherwise)
                lam = rng.uniform(0.3, 0.9, d) * scale
                amp, sig = float(rng.uniform(0.7, 1.5)), float(rng.uniform(0.03, 0.1))
                M = int(rng.choice([1, 5, 63, 64, 129, 500, 1024, 1100, 2048]))
                Xs = np.asfortranarray(rng.uniform(-0.05 * scale, 1.05 * scale, (d, M)))
                mask = O.in_bounds(Xs, np.zeros(d), np.full(d, scale)) if rng.random() < 0.5 else None
                ymax = None if rng.random() < 0.5 else float(y.max()) + 0.5
                best = float(np.median(y)) if rng.random() < 0.8 else None
                r = g.update_acq(lam, amp, sig, api.Candidates(Xs), 1.0, ymax, best, mfun(X), mfun(Xs), mask, want_acq=True, want_moments=True)
                post = O.gp_fit(X, y, kern, lam, amp, sig, mean=mfun(X), discrete=disc)
                mu_o, var_o = O.gp_mean_and_var(post, Xs, mfun(Xs), clip=False)
                want = O.ei_acquisition([post], Xs, [1.0], None if ymax is None else [ymax], best, valid_mask=mask,
                                        means_s=None if not use_mean else [mfun(Xs)])
                e = max(abs(r["logpdf"] - post.logpdf) / (1 + abs(post.logpdf)), np.abs(r["mu"] - mu_o).max() / (1 + np.abs(mu_o).max()),
                        np.abs(r["var"] - var_o).max() / amp ** 2, np.abs(r["acq"] - want).max())
                assert r["argmax"] == int(np.argmax(r["acq"])) and r["max"] == r["acq"].max(), ("update_acq argmax", case)
                nfused = nfused + int(r["fused"])
                if track is not None:
                    track.close()
                    track = None
            elif op == "batch":
                # batched likelihoods on the current data (boss_gp_loglike_batch): S hyper-parameter sets, one of them
                # possibly not positive definite (duplicated data without noise cannot be made here: use a huge lengthscale)
                S = int(rng.choice([1, 2, 3, 5, 8, 17, 40])) if X.shape[1] <= 1100 else int(rng.choice([1, 3, 5]))
                lams = rng.uniform(0.3, 0.9, (d, S)) * scale
                amps, sigs = rng.uniform(0.7, 1.5, S), rng.uniform(0.03, 0.1, S)
                per_set_mean = use_mean and rng.random() < 0.5
                mX = None if not use_mean else (np.stack([mfun(X) + 0.01 * k for k in range(S)]) if per_set_mean else mfun(X))
                with_grad = d <= 32 and rng.random() < 0.5
                if with_grad:
                    ll, st, gb = api.loglike_batch(X, y, kern, lams, amps, sigs, mX, disc, want_grad=True)
                else:
                    ll, st = api.loglike_batch(X, y, kern, lams, amps, sigs, mX, disc)
                e = 0.0
                for k in range(min(S, 2) if with_grad else 0):
                    mk = None if mX is None else (mX[k] if per_set_mean else mX)
                    _, gr_o = O.gp_data_loglike_grad(X, y, kern, lams[:, k], amps[k], sigs[k], mean=mk, discrete=disc)
                    e = max(e, np.abs(gb[:, k] - gr_o).max() / (1 + np.abs(gr_o).max()) / 10)
                for k in range(S):
                    mk = None if mX is None else (mX[k] if per_set_mean else mX)
                    want = O.gp_data_loglike_slice(X, y, kern, lams[:, k], amps[k], sigs[k], mean=mk, discrete=disc)
                    e = max(e, abs(ll[k] - want) / (1 + abs(want)))
            elif op == "track":
                # tracked candidates (resident V slabs): created once, must follow every later append / update
                if track is None:
                    M = int(rng.choice([1, 31, 64, 100, 300]))
                    tXs = np.asfortranarray(rng.uniform(0, scale, (d, M)))
                    tcand = api.Candidates(tXs)
                    track = api.Track(g, tcand, mfun(tXs))
                mu, var = track.moments()
                mu_o, var_o = O.gp_mean_and_var(post, tXs, mfun(tXs), clip=False)
                e = max(np.abs(mu - mu_o).max() / (1 + np.abs(mu_o).max()), np.abs(var - var_o).max() / amp ** 2)
                if rng.random() < 0.5:
                    acq, am, mx = api.acq_ei_tracks([[track]], [1.0], None, float(y.max()), None)
                    want = O.ei_acquisition([post], tXs, [1.0], None, float(y.max()), means_s=None if not use_mean else [mfun(tXs)],
                                            constrained=False)
                    e = max(e, np.abs(acq - want).max())
            elif op == "acq":
                M = int(rng.choice([1, 4, 33, 200, 1100]))
                Xs = np.asfortranarray(rng.uniform(-0.05 * scale, 1.05 * scale, (d, M)))
                mask = O.in_bounds(Xs, np.zeros(d), np.full(d, scale))
                ms = None if not use_mean else mfun(Xs).reshape(1, 1, M)
                acq, am, mx = api.acq_ei([[g]], api.Candidates(Xs), [1.0], [float(y.max()) + 0.5], float(np.median(y)), mask, ms)
                want = O.ei_acquisition([post], Xs, [1.0], [float(y.max()) + 0.5], float(np.median(y)), valid_mask=mask,
                                        means_s=None if not use_mean else [mfun(Xs)])
                e = np.abs(acq - want).max()
                assert am == int(np.argmax(acq)), ("argmax", case)
            elif op == "acqgrad":
                M = int(rng.choice([1, 3, 40, 224]))
                Xs = np.asfortranarray(rng.uniform(0, scale, (d, M)))
                mg = np.vstack([np.full(M, -0.1), np.zeros((d - 1, M))]) if use_mean else None
                acq, dacq = api.acq_ei_grad([g], Xs, [1.0], None, float(np.median(y)), None,
                                            None if not use_mean else mfun(Xs).reshape(1, M), None if mg is None else mg.reshape(1, d, M))
                want, dwant = O.ei_acquisition_grad([post], Xs, [1.0], None, float(np.median(y)),
                                                    means_s=None if not use_mean else [mfun(Xs)],
                                                    mean_grads_s=None if mg is None else [mg])
                e = max(np.abs(acq - want).max(), np.abs(dacq - dwant).max() / (1 + np.abs(dwant).max()) / 10)
            else:
                if d > 32:
                    continue
                lpg, gr = g.loglike_grad()
                _, gr_o = O.gp_data_loglike_grad(X, y, kern, lam, amp, sig, mean=mfun(X), discrete=disc)
                e = np.abs(gr - gr_o).max() / (1 + np.abs(gr_o).max()) / 10
        except Exception as ex:
            print(f"CASE {case} d={d} N={N} {kern} disc={disc is not None} ops={ops} M={M}: EXCEPTION {type(ex).__name__}: {ex}", flush=True)
            try:
                if op == "predict":
                    for rep in range(2):
                        try:
                            g.predict(Xs, mfun(Xs))
                            print("  same candidates again: ok", flush=True)
                        except Exception as e4:
                            print("  same candidates again: fails at", getattr(e4, "bad_index", None), flush=True)
                L, z = g.factor()
                print("  diag: N now", X.shape[1], "L finite", np.isfinite(L).all(), "z finite", np.isfinite(z).all(),
                      "|L-Lo|", np.abs(L - post.L).max(), flush=True)
                for Mt in (1, 5, 40):
                    Xt = rng.uniform(0, scale, (d, Mt))
                    for rep in range(3):
                        try:
                            mu_t, var_t = g.predict(Xt, mfun(Xt))
                            print("  retry predict M", Mt, "rep", rep, "ok, nan:", np.isnan(mu_t).sum(), np.isnan(var_t).sum(), flush=True)
                        except Exception as e2:
                            print("  retry predict M", Mt, "rep", rep, "fails:", getattr(e2, "bad_index", None), flush=True)
            except Exception as e3:
                print("  diag failed", e3, flush=True)
            raise
        worst = max(worst, e / tol)
        if e > tol:
            print(f"CASE {case} d={d} N={N} {kern} disc={disc is not None} mean={use_mean} ops={ops}: error {e:.3e} > tol {tol:.1e}", flush=True)
            sys.exit(1)
    if track is not None:
        track.close()
    g.close()
    if case % 5 == 4:
        print(f"  {case + 1} cases ok, worst error/tolerance so far {worst:.2e}, {time.time() - t_start:.0f} s", flush=True)
print(f"fuzz: {ncases} cases passed, worst error/tolerance {worst:.2e}, {nfused} fused update_acq calls")

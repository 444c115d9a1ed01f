"""world_size-2 gloo tests (CPU) of the N>1 path: candidate / sample sharding and the 16-byte
(value, index) arg-max exchange.  The device call is replaced by the oracle so the sharding and
collective logic is what is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import boss_jl_amd as B
        from boss_jl_amd import distributed as D
        from boss_jl_amd import maximizer as MX
        from oracle import gp_oracle as O
        # ---- raw exchange
        v, i = D.argmax_exchange(float(rank), 100 - rank)
        assert (v, i) == (float(world - 1), 100 - (world - 1))
        v, i = D.argmax_exchange(2.5, 40 + rank)          # tie -> smallest global index
        assert (v, i) == (2.5, 40)
        cat = D.allgather_concat(np.arange(3 + rank, dtype=float) + 10 * rank)
        assert cat.shape[0] == sum(3 + r for r in range(world))
        # ---- maximizer sharding with an oracle-backed acquisition
        rng = np.random.default_rng(0)
        d, N, M = 2, 30, 101
        X = rng.uniform(0, 1, (d, N))
        y = np.sin(3 * X).sum(0)
        post = O.gp_fit(X, y, "matern52", [0.4, 0.6], 1.0, 0.05)
        Xs = np.asfortranarray(np.random.default_rng(5).uniform(-0.2, 1.2, (d, M)))

        def fake_acq(problem, posts, Xc, cand=None):
            mask = O.in_bounds(Xc, [0., 0.], [1., 1.])
            a = O.ei_acquisition([post], Xc, [1.0], [np.inf], float(y.max()), valid_mask=mask)
            j = int(np.argmax(a))
            return a, j, float(a[j])

        MX.acquisition_values = fake_acq
        MX.posteriors_of = lambda problem: [None]
        am = B.HipBatchAM(points=Xs)
        x, val = am.maximize_acquisition(problem=None)
        full = O.ei_acquisition([post], Xs, [1.0], [np.inf], float(y.max()), valid_mask=O.in_bounds(Xs, [0., 0.], [1., 1.]))
        j = int(np.argmax(full))
        assert np.array_equal(x, Xs[:, j]) and val == full[j]
        _, allv = am.maximize_acquisition(problem=None, return_all=True)
        assert np.array_equal(allv, full)
        _shard_modes(B, D, MX, O, rank, world)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _shard_modes(B, D, MX, O, rank, world):
    """shard='samples' (all-reduce of partial sums) and shard='outputs' (all-gather of moment
    rows): P = 3 outputs (ragged over 2 ranks), S = 5 samples (ragged), oracle-backed device calls."""
    rng = np.random.default_rng(11)
    d, N, M, P, S = 2, 25, 57, 3, 5
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), np.cos(2 * X).sum(0), X[0] - X[1]])
    Xs = np.asfortranarray(rng.uniform(-0.1, 1.1, (d, M)))
    y_max = np.array([np.inf, 1.5, 0.2])
    coefs = [1.0, 0.5, 0.0]
    prm = [B.HipGPParams(rng.uniform(0.3, 0.8, (d, P)), rng.uniform(0.8, 1.5, P), rng.uniform(0.03, 0.1, P)) for _ in range(S)]
    model = B.HipGaussianProcess([None] * P, [None] * P, [None] * P)
    dom = B.Domain((np.zeros(d), np.ones(d)))
    prob = B.BossProblem(None, dom, B.ExpectedImprovement(B.LinFitness(coefs)), model, B.ExperimentData(X, Y), y_max, prm)

    def opost(s, i):
        return O.gp_fit(X, Y[i], "matern52", prm[s].lengthscales[:, i], prm[s].amplitudes[i], prm[s].noise_std[i])

    all_posts = [[opost(s, i) for i in range(P)] for s in range(S)]
    b = O.best_so_far(coefs, Y, y_max)
    mask = O.in_bounds(Xs, np.zeros(d), np.ones(d))
    full = O.ei_acquisition(all_posts, Xs, coefs, y_max, b, valid_mask=mask)
    calls = {"posts": 0, "moments": []}

    def fake_posteriors_of(problem, lo=None, hi=None):
        calls["posts"] += 1
        return all_posts[lo:hi]

    def fake_acq(problem, posts, Xc, cand=None):
        a = O.ei_acquisition(posts, Xc, coefs, y_max, b, valid_mask=mask)
        j = int(np.argmax(a))
        return a, j, float(a[j])

    def fake_acq_nomask(problem, posts, Xc, cand=None):
        a = O.ei_acquisition(posts, Xc, coefs, y_max, b)
        j = int(np.argmax(a))
        return a, j, float(a[j])

    def fake_output_moments(problem, i, Xc):
        calls["moments"].append(i)
        return np.stack([np.stack(O.gp_mean_and_var(all_posts[s][i], Xc)) for s in range(S)])

    def fake_moments_acq(problem, mu, var, Xc):
        acc = np.zeros(Xc.shape[1])
        for s in range(mu.shape[0]):
            acc += O.expected_improvement_lin(coefs, mu[s], var[s], b) * O.feas_prob(mu[s], var[s], y_max)
        a = np.where(mask, acc / mu.shape[0], 0.0)
        j = int(np.argmax(a))
        return a, j, float(a[j])

    MX.posteriors_of, MX.acquisition_values = fake_posteriors_of, fake_acq
    MX.output_moments, MX.moments_acquisition = fake_output_moments, fake_moments_acq
    j = int(np.argmax(full))
    # ---- samples: rank r factorises only its shard; one all-reduce(sum)
    am = B.HipBatchAM(points=Xs, shard="samples")
    x, val = am.maximize_acquisition(prob)
    assert np.array_equal(x, Xs[:, j]) and abs(val - full[j]) <= 1e-15 * (1 + abs(full[j]))
    _, allv = am.maximize_acquisition(prob, return_all=True)
    assert np.allclose(allv, full, rtol=0, atol=1e-15)
    # ---- outputs: rank r factorises outputs i = r mod world only; one all-gather of rows
    am = B.HipBatchAM(points=Xs, shard="outputs")
    x, val = am.maximize_acquisition(prob)
    assert sorted(set(calls["moments"])) == [i for i in range(P) if i % world == rank]
    assert np.array_equal(x, Xs[:, j]) and abs(val - full[j]) <= 1e-15 * (1 + abs(full[j]))
    _, allv = am.maximize_acquisition(prob, return_all=True)
    assert np.allclose(allv, full, rtol=0, atol=1e-15)
    # ---- unseeded maximiser: rank 0's seed is broadcast, so every rank draws the same candidates and returns the same point
    MX.posteriors_of, MX.acquisition_values = fake_posteriors_of, fake_acq_nomask
    for _ in range(2):
        amu = B.HipBatchAM(x_prior=lambda r: r.uniform(0, 1, d), samples=23)          # seed=None
        xu, vu = amu.maximize_acquisition(prob)
        both = D.allgather_concat(np.concatenate([xu, [vu]]))
        assert np.array_equal(both[:d + 1], both[d + 1:]), both
        assert abs(O.ei_acquisition(all_posts, xu[:, None], coefs, y_max, b)[0] - vu) <= 1e-12
    s1, s2 = D.shared_seed(None), D.shared_seed(None)
    assert s1 != s2 and D.shared_seed(17) == 17
    both = D.allgather_concat(np.array([float(s1 % 2 ** 40), float(s2 % 2 ** 40)]))
    assert np.array_equal(both[:2], both[2:])
    assert np.array_equal(D.broadcast_array(np.arange(3.0) + rank if rank == 1 else None, (3,), 1), np.arange(3.0) + 1)
    assert [D.owner_of_index(i, 5, 2) for i in range(5)] == [0, 0, 0, 1, 1]
    MX.posteriors_of, MX.acquisition_values = fake_posteriors_of, fake_acq
    # ---- gradient multistart: starts sharded across ranks, identical winner on every rank
    post1 = O.gp_fit(X, Y[0], "matern52", [0.4, 0.5], 1.0, 0.05)
    bb = float(Y[0].max())

    def fake_vg(self, problem, posts, Xc):
        return O.ei_acquisition_grad([post1], Xc, [1.0], None, bb, valid_mask=O.in_bounds(Xc, np.zeros(d), np.ones(d)))

    B.HipGradientAM._value_and_grad = fake_vg
    gam = B.HipGradientAM(x_prior=lambda r: r.uniform(0, 1, d), multistart=11, iters=15, seed=9)
    xg, vg = gam.maximize_acquisition(prob, posts=[None])
    both = D.allgather_concat(np.concatenate([xg, [vg]]))
    assert np.array_equal(both[:d + 1], both[d + 1:]), both          # same answer on both ranks
    assert abs(O.ei_acquisition([post1], xg[:, None], [1.0], None, bb)[0] - vg) <= 1e-12
    # ---- fitters: hyper-parameter samples / multistart starts sharded across ranks, identical winner everywhere
    Xf = rng.uniform(0, 1, (d, 25))
    Yf = np.stack([np.sin(3 * Xf).sum(0), np.cos(2 * Xf).sum(0)])
    fmodel = B.HipGaussianProcess(lengthscale_priors=[B.MvLogNormal([-0.7] * d, [0.5] * d)] * 2,
                                  amplitude_priors=[B.LogNormal(0.0, 0.5)] * 2, noise_std_priors=[B.Dirac(0.05)] * 2)
    fprob = B.BossProblem(None, B.Domain((np.zeros(d), np.ones(d))), B.ExpectedImprovement(B.LinFitness([1.0, 0.0])), fmodel,
                          B.ExperimentData(Xf, Yf))
    seen = {"batch": [], "grad": []}

    def fake_data_loglike_batch(self, data, samples):
        seen["batch"].append(len(samples))
        return np.array([O.data_loglike(data.X, data.Y, "matern52", p.lengthscales, p.amplitudes, p.noise_std) for p in samples])

    def fake_objective_batch(self, model, prior_ll, data, plist):
        seen["grad"].append(len(plist))
        S, (dd, P) = len(plist), plist[0].lengthscales.shape
        tot, gl, ga, gs = np.zeros(S), np.zeros((S, dd, P)), np.zeros((S, P)), np.zeros((S, P))
        for k, p in enumerate(plist):
            tot[k] = prior_ll(p)
            for i in range(P):
                ll, gr = O.gp_data_loglike_grad(data.X, data.Y[i], "matern52", p.lengthscales[:, i], p.amplitudes[i], p.noise_std[i])
                tot[k] += ll
                gl[k, :, i] = gr[:dd] + np.atleast_1d(model.lengthscale_priors[i].grad_logpdf(p.lengthscales[:, i]))
                ga[k, i] = gr[dd] + model.amplitude_priors[i].grad_logpdf(p.amplitudes[i])
                gs[k, i] = gr[dd + 1] + model.noise_std_priors[i].grad_logpdf(p.noise_std[i])
        return tot, gl, ga, gs

    B.HipGaussianProcess.data_loglike_batch = fake_data_loglike_batch
    B.HipGradientMAP._objective_batch = fake_objective_batch
    best = B.HipBatchedMAP(samples=11, seed=4).estimate_parameters(fprob)
    assert sum(seen["batch"]) == len(range(*D.shard_range(11, rank, world)))          # only this rank's shard was evaluated
    allp = B.HipBatchedMAP(samples=11, seed=4).estimate_parameters(fprob, return_all=True)
    assert len(allp) == 11 and best.loglike == max(r.loglike for r in allp)
    fit = B.HipGradientMAP(multistart=5, iters=6, seed=4).estimate_parameters(fprob)
    lo_, hi_ = D.shard_range(5, rank, world)
    assert seen["grad"][0] == hi_ - lo_                                                # first round: this rank's starts
    flat = np.concatenate([fit.params.lengthscales.reshape(-1), fit.params.amplitudes, fit.params.noise_std, [fit.loglike]])
    both = D.allgather_concat(flat)
    assert np.array_equal(both[:flat.size], both[flat.size:])                          # same winner on both ranks
    assert fit.loglike >= best.loglike - 1e-9 and np.all(fit.params.noise_std == 0.05)
    so = B.HipSampleOptMAP(samples=11, multistart=3, iters=4, seed=4).estimate_parameters(fprob)
    assert so.loglike >= best.loglike - 1e-9
    # unseeded fitters: the broadcast seed makes every rank draw the same samples / starts and return the same winner
    for fitter in (B.HipBatchedMAP(samples=9), B.HipGradientMAP(multistart=4, iters=3), B.HipSampleOptMAP(samples=9, multistart=2, iters=3)):
        r = fitter.estimate_parameters(fprob)
        flat = np.concatenate([r.params.lengthscales.reshape(-1), r.params.amplitudes, r.params.noise_std, [r.loglike]])
        both = D.allgather_concat(flat)
        assert np.array_equal(both[:flat.size], both[flat.size:]), type(fitter).__name__
    allu = B.HipBatchedMAP(samples=9).estimate_parameters(fprob, return_all=True)
    chk = np.array([O.data_loglike(fprob.data.X, fprob.data.Y, "matern52", q.params.lengthscales, q.params.amplitudes, q.params.noise_std)
                    + fmodel.params_loglike()(q.params) for q in allu])
    assert np.allclose(chk, [q.loglike for q in allu], rtol=0, atol=1e-9)        # draws paired with THEIR values on every rank
    # ---- raw collectives
    tot = D.allreduce_sum(np.arange(4.0) + rank)
    assert np.array_equal(tot, world * np.arange(4.0) + sum(range(world)))
    rows = D.allgather_owned({i: np.full((2,), float(i)) for i in range(5) if i % world == rank}, 5, (2,))
    assert np.array_equal(rows, np.repeat(np.arange(5.0)[:, None], 2, axis=1))


@pytest.mark.timeout(180)
def test_world2_gloo_sharding_and_argmax():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"

"""Worker of tests/test_gpu_two_ranks.py: one rank of a world-size-2 gloo group, BOTH ranks on GPU 0, every device call
through the real libbosship.so (the CPU gloo test replaces the device by the oracle; this one does not).  Checks the
three shard modes of HipBatchAM, the sequential-batch maximiser on tracked candidates, the gradient maximiser and the
sample-sharded fitters against the oracle's unsharded answer, and that both ranks return the same result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import boss_jl_amd as B
    from boss_jl_amd import api
    from boss_jl_amd import distributed as D
    from oracle import gp_oracle as O
    api.load_library()
    assert api.device_count() >= 1

    def same_on_all_ranks(vec):
        both = D.allgather_concat(np.asarray(vec, float).reshape(-1))
        n = both.size // world
        for r in range(1, world):
            assert np.array_equal(both[:n], both[r * n:(r + 1) * n]), both

    rng = np.random.default_rng(11)
    d, N, M, P, S = 3, 200, 301, 3, 5                       # ragged over two ranks: M odd, P = 3, S = 5
    X = rng.uniform(0, 1, (d, N))
    Y = np.stack([np.sin(3 * X).sum(0), np.cos(2 * X).sum(0), X[0] - X[1] + 0.2 * X[2]])
    Xs = np.asfortranarray(rng.uniform(-0.1, 1.1, (d, M)))
    y_max = np.array([np.inf, 1.5, 0.4])
    coefs = [1.0, 0.5, 0.0]
    prm = [B.HipGPParams(rng.uniform(0.3, 0.8, (d, P)), rng.uniform(0.8, 1.5, P), rng.uniform(0.03, 0.1, P)) for _ in range(S)]
    model = B.HipGaussianProcess([None] * P, [None] * P, [None] * P, mean=lambda x: [0.1, -0.2, 0.3])
    dom = B.Domain((np.zeros(d), np.ones(d)))
    prob = B.BossProblem(None, dom, B.ExpectedImprovement(B.LinFitness(coefs)), model, B.ExperimentData(X, Y), y_max, prm)
    means = [0.1, -0.2, 0.3]
    oposts = [[O.gp_fit(X, Y[i], "matern52", p.lengthscales[:, i], p.amplitudes[i], p.noise_std[i], mean=means[i]) for i in range(P)]
              for p in prm]
    b = O.best_so_far(coefs, Y, y_max)
    mask = O.in_bounds(Xs, np.zeros(d), np.ones(d))
    ms = [np.full(M, m) for m in means]
    want = O.ei_acquisition(oposts, Xs, coefs, y_max, b, valid_mask=mask, means_s=ms)
    j = int(np.argmax(want))
    for mode in ("candidates", "outputs", "samples"):
        am = B.HipBatchAM(points=Xs, shard=mode)
        x, val = am.maximize_acquisition(prob)
        assert np.array_equal(x, Xs[:, j]) and abs(val - want[j]) <= 1e-11, (mode, val, want[j])
        _, allv = am.maximize_acquisition(prob, return_all=True)
        assert allv.shape == (M,) and np.allclose(allv, want, rtol=0, atol=1e-11), mode
        same_on_all_ranks(np.concatenate([x, [val]]))
    # unseeded random candidates: same draw and same winner on both ranks (rank 0's seed is broadcast)
    amu = B.HipBatchAM(x_prior=lambda r: r.uniform(0, 1, d), samples=77)
    xu, vu = amu.maximize_acquisition(prob)
    same_on_all_ranks(np.concatenate([xu, [vu]]))
    assert abs(O.ei_acquisition(oposts, xu[:, None], coefs, y_max, b, means_s=[np.full(1, m) for m in means])[0] - vu) <= 1e-11
    # sequential batch on tracked candidates (candidates sharded, the owner broadcasts the speculative observation)
    prob1 = B.BossProblem(None, dom, B.ExpectedImprovement(B.LinFitness(coefs)), model, B.ExperimentData(X, Y), y_max, prm[0])
    sb = B.HipSequentialBatchAM(B.HipBatchAM(points=Xs), batch_size=3)
    Xb, _ = sb.maximize_acquisition(prob1)
    same_on_all_ranks(Xb)
    Xo, Yo, sel = X.copy(), Y.copy(), []
    for _ in range(3):                                       # the reference's loop (batch.jl:26-38) restated with the oracle
        po = [O.gp_fit(Xo, Yo[i], "matern52", prm[0].lengthscales[:, i], prm[0].amplitudes[i], prm[0].noise_std[i], mean=means[i])
              for i in range(P)]
        a = O.ei_acquisition(po, Xs, coefs, y_max, O.best_so_far(coefs, Yo, y_max), valid_mask=mask, means_s=ms)
        jj = int(np.argmax(a))
        xs_ = Xs[:, jj]
        yh = np.array([O.gp_mean_and_var(po[i], xs_[:, None], np.full(1, means[i]))[0][0] for i in range(P)])
        Xo, Yo = np.hstack([Xo, xs_[:, None]]), np.hstack([Yo, yh[:, None]])
        sel.append(xs_)
    assert np.array_equal(Xb, np.stack(sel, axis=1))
    # gradient multistart: starts sharded, winner's point broadcast from its owner
    gam = B.HipGradientAM(x_prior=lambda r: r.uniform(0, 1, d), multistart=9, iters=8, seed=5)
    xg, vg = gam.maximize_acquisition(prob1)
    same_on_all_ranks(np.concatenate([xg, [vg]]))
    po = [O.gp_fit(X, Y[i], "matern52", prm[0].lengthscales[:, i], prm[0].amplitudes[i], prm[0].noise_std[i], mean=means[i]) for i in range(P)]
    assert abs(O.ei_acquisition(po, xg[:, None], coefs, y_max, b, means_s=[np.full(1, m) for m in means])[0] - vg) <= 1e-10
    # fitters: hyper-parameter samples sharded across the ranks
    fmodel = B.HipGaussianProcess(lengthscale_priors=[B.MvLogNormal([-0.7] * d, [0.5] * d)] * P,
                                  amplitude_priors=[B.LogNormal(0.0, 0.5)] * P, noise_std_priors=[B.Dirac(0.05)] * P)
    fprob = B.BossProblem(None, dom, B.ExpectedImprovement(B.LinFitness(coefs)), fmodel, B.ExperimentData(X, Y), y_max)
    for fitter in (B.HipBatchedMAP(samples=13, seed=4), B.HipBatchedMAP(samples=13), B.HipGradientMAP(multistart=3, iters=3, seed=2)):
        r = fitter.estimate_parameters(fprob)
        same_on_all_ranks(np.concatenate([r.params.lengthscales.reshape(-1), r.params.amplitudes, r.params.noise_std, [r.loglike]]))
        chk = O.data_loglike(X, Y, "matern52", r.params.lengthscales, r.params.amplitudes, r.params.noise_std) + fmodel.params_loglike()(r.params)
        assert abs(chk - r.loglike) <= 1e-9 * (1 + abs(chk)), (type(fitter).__name__, chk, r.loglike)
    allp = B.HipBatchedMAP(samples=13, seed=4).estimate_parameters(fprob, return_all=True)
    best = B.HipBatchedMAP(samples=13, seed=4).estimate_parameters(fprob)
    assert len(allp) == 13 and best.loglike == max(q.loglike for q in allp)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: ok", flush=True)


if __name__ == "__main__":
    main()

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
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


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

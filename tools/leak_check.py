"""Device-memory leak check: create / use / close handles of every kind in a loop and watch the free device memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from boss_jl_amd import api
rng = np.random.default_rng(0)
d, N, M = 4, 1500, 300
X = rng.uniform(0, 1, (d, N)); y = np.sin(X).sum(0); Xs = rng.uniform(0, 1, (d, M))
dY = np.cos(X)


def cycle():
    g = api.GP(X, y, "matern52")
    g.update(np.full(d, 0.5), 1.0, 0.05)
    for _ in range(3):
        g.predict(Xs[:, :1])
        g.predict(Xs[:, :40])
    g.predict(Xs); g.predict_grad(Xs[:, :50]); g.predict_cov(Xs[:, :20]); g.loglike_grad()
    g.append(rng.uniform(0, 1, (d, 600)), np.zeros(600))          # grows the storage
    cand = api.Candidates(Xs)
    tr = api.Track(g, cand)
    g.append(rng.uniform(0, 1, (d, 1)), [0.0])
    tr.moments()
    api.acq_ei([[g]], cand, [1.0], None, 0.5, None)
    tr.close(); cand.close(); g.close()
    gg = api.GradGP(X[:, :200], y[:200], dY[:, :200], "sqexp")
    gg.update(np.full(d, 0.5), 1.0, 0.05, 0.1); gg.predict(Xs[:, :3]); gg.predict(Xs[:, :3]); gg.close()
    gn = api.GibbsGP(X, y)
    gn.update(np.full((d, N), 0.5), np.ones(N), np.full(N, 0.05)); gn.predict(Xs[:, :2], np.full((d, 2), 0.5), np.ones(2)); gn.close()
    api.loglike_batch(X, y, "matern52", np.full((d, 5), 0.5), np.ones(5), np.full(5, 0.05))


for _ in range(3):
    cycle()                                                      # grow-only per-device workspaces reach their size
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for _ in range(30):
    cycle()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"free device memory before {free0/2**20:.1f} MiB, after 30 cycles {free1/2**20:.1f} MiB, difference {(free0-free1)/2**20:.2f} MiB")
assert free0 - free1 < 8 * 2 ** 20, "device memory leak"
print("leak check passed")

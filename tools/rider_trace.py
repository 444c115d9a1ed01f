import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
entry.build()
from boss_jl_amd import api
N, D, M = 4096, 8, int(os.environ.get("M", 1024))
rng = np.random.default_rng(1)
X = rng.uniform(0, 1, (D, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(D) + 0.05 * rng.standard_normal(N)
Xs = np.random.default_rng(2).uniform(0, 1, (D, M))
g = api.GP(X, y, "matern52"); cand = api.Candidates(Xs)
for i in range(int(os.environ.get("REP", 8))):
    g.update_acq(np.full(D, 0.5), 1.0, 0.05 + 1e-4 * i, cand, best=float(y.max()))

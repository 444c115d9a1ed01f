import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from boss_jl_amd import api
from oracle import gp_oracle as O
api.load_library()
for N in (300, 384, 700, 1000, 1536, 2048, 2100, 4096):
    rng = np.random.default_rng(N); d = 8
    X = rng.uniform(0, 1, (d, N)); y = np.sin(2*np.pi*X).sum(0)/np.sqrt(d) + 0.05*rng.standard_normal(N)
    lam = np.full(d, 0.5)
    g = api.GP(X, y, "matern52"); lp = g.update(lam, 1.0, 0.05)
    post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05)
    L, z = g.factor()
    print(N, "dlogpdf %.2e" % (abs(lp - post.logpdf)/(1+abs(post.logpdf))), "dL %.2e" % np.abs(L - post.L).max(), flush=True)
    lp2 = g.update(lam, 1.0, 0.05); assert lp2 == lp, (lp, lp2)
    g.close()

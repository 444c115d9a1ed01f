"""update + first acquisition: two calls (boss_gp_update, boss_acq_ei) against one (boss_gp_update_acq), N = 4096, d = 8."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
entry.build()
import os as _os
from boss_jl_amd import api
if os.environ.get('BOSS_LIB_PATH'): api.load_library(os.environ['BOSS_LIB_PATH'])
N, D = int(os.environ.get("N", 4096)), 8
rng = np.random.default_rng(1)
X = rng.uniform(0, 1, (D, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(D) + 0.05 * rng.standard_normal(N)
Xs = np.random.default_rng(2).uniform(0, 1, (D, 8192))
lam = np.full(D, 0.5); best = float(y.max())
g = api.GP(X, y, "matern52")
REP = int(os.environ.get("REP", 40))
def med(f):
    ts = []
    for i in range(REP):
        t0 = time.perf_counter(); f(i); ts.append(time.perf_counter() - t0)
    ts = np.sort(np.array(ts[3:])) * 1e3
    return float(np.median(ts)), float(ts[0]), float(ts[-1])
print("update alone         p50 %.3f min %.3f max %.3f ms" % med(lambda i: g.update(lam, 1.0, 0.05 + 1e-4 * (i % 7))), flush=True)
for M in [int(v) for v in os.environ.get("MS", "1024,2048,4096,8192").split(",")]:
    cand = api.Candidates(Xs[:, :M])
    def two(i):
        g.update(lam, 1.0, 0.05 + 1e-4 * (i % 7))
        api.acq_ei([[g]], cand, [1.0], None, best, want_acq=False)
    fl = []
    def one(i):
        fl.append(g.update_acq(lam, 1.0, 0.05 + 1e-4 * (i % 7), cand, best=best)["fused"])
    a = med(two); b = med(one)
    print("M %5d  two calls p50 %.3f min %.3f max %.3f | one call p50 %.3f min %.3f max %.3f ms  fused %d/%d" % (M, *a, *b, sum(fl), len(fl)), flush=True)
    cand.close()
print("update alone (again) p50 %.3f min %.3f max %.3f ms" % med(lambda i: g.update(lam, 1.0, 0.05 + 1e-4 * (i % 7))), flush=True)

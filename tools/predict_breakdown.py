import sys,os,time
sys.path.insert(0,'/root/repo')
import numpy as np
from boss_jl_amd import api
if os.environ.get('BOSS_LIB'): api.load_library(os.environ['BOSS_LIB'])
def problem(d, N, M, seed=1, noise=0.05):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (d, N))
    y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + noise * rng.standard_normal(N)
    return X, y, np.random.default_rng(seed + 1).uniform(0, 1, (d, M))
X,y,Xs=problem(8,4096,8192)
g=api.GP(X,y,"matern52"); g.update(np.full(8,.5),1.0,0.05)
cand=api.Candidates(Xs); b=float(y.max())
api.acq_ei([[g]],cand,[1.0],None,b,want_acq=False)
api.prof_enable(0,True); api.prof_reset(0)
for _ in range(3): api.acq_ei([[g]],cand,[1.0],None,b,want_acq=False)
ms,n=api.prof_get(0,"predict"); print("lib=%s BOSS_DBG=%s predict %.3f ms"%(os.path.basename(os.environ.get("BOSS_LIB","new")),os.environ.get("BOSS_DBG","0"),ms/n),flush=True)

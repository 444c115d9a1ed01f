import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
from boss_jl_amd import api
rng = np.random.default_rng(0)
for N in (4096, 8192):
    X = rng.uniform(0, 1, (8, N)); y = np.sin(X).sum(0)
    g = api.GP(X, y, "matern52"); lam = np.full(8, .5)
    for _ in range(2): g.update(lam, 1.0, 0.05)
    ts = []
    for r in range(3):
        t = time.perf_counter()
        for i in range(10): g.update(lam, 1.0, 0.05 + 1e-4 * i)
        ts.append((time.perf_counter() - t) / 10)
    print(f"pairs={'off' if os.environ.get('BOSS_NO_PAIRS')=='1' else 'on'} mask={os.environ.get('BOSS_SIDE_CU_MASK','0')} N={N}: update min {min(ts)*1e3:.3f} ms  median {sorted(ts)[1]*1e3:.3f} ms", flush=True)
    g.close()

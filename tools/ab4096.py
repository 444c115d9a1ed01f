import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from boss_jl_amd import api
lib = sys.argv[1] if len(sys.argv) > 1 else None
if lib: api.load_library(lib)
rng = np.random.default_rng(0)
N = 4096
X = rng.uniform(0, 1, (8, N)); y = np.sin(X).sum(0)
g = api.GP(X, y, "matern52"); lam = np.full(8, .5)
for _ in range(3): g.update(lam, 1.0, 0.05)
ts = []
for r in range(7):
    t = time.perf_counter()
    for i in range(10): g.update(lam, 1.0, 0.05 + 1e-4 * i)
    ts.append((time.perf_counter() - t) / 10)
print(f"[{lib or 'default'}] N={N}: update min {min(ts)*1e3:.3f} ms median {sorted(ts)[3]*1e3:.3f} ms", flush=True)

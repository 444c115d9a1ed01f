"""Host/launch overhead of one boss_acq_ei call: tiny posterior (N=128) so the kernels are short."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
rng = np.random.default_rng(0)
for N, M in ((128, 8192), (128, 256), (4096, 8192)):
    X = rng.uniform(0, 1, (8, N)); y = np.sin(X).sum(0)
    g = api.GP(X, y, "matern52"); g.update(np.full(8, .5), 1.0, 0.05)
    cand = api.Candidates(rng.uniform(0, 1, (8, M)))
    for _ in range(3): api.acq_ei([[g]], cand, [1.0], None, 1.0, want_acq=False)
    t = time.perf_counter()
    for _ in range(20): api.acq_ei([[g]], cand, [1.0], None, 1.0, want_acq=False)
    dt = (time.perf_counter() - t) / 20
    t = time.perf_counter()
    for _ in range(20):
        g.update(np.full(8, .5), 1.0, 0.05)
    du = (time.perf_counter() - t) / 20
    t = time.perf_counter()
    for _ in range(20):
        g.update(np.full(8, .5), 1.0, 0.05); api.acq_ei([[g]], cand, [1.0], None, 1.0, want_acq=False)
    db = (time.perf_counter() - t) / 20
    print(f"N={N} M={M}: acq {dt*1e6:.0f} us, update {du*1e6:.0f} us, update+acq {db*1e6:.0f} us", flush=True)

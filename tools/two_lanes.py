"""Two contexts on ONE GPU (BOSS_VIRTUAL_DEVICES=2): do two updates issued from two host threads overlap on the device?
python tools/two_lanes.py [N]   (try with and without GPU_MAX_HW_QUEUES=8: eight streams on four hardware queues serialise)"""
import os, sys, time, threading
os.environ.setdefault("BOSS_VIRTUAL_DEVICES", "2")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boss_jl_amd import api
api.load_library()
import ctypes as C
lib = api.load_library()
lib.boss_debug_fallbacks.argtypes = [C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_int)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 8
rng = np.random.default_rng(1)
X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
lam = np.full(d, 0.5)
gs = [api.GP(X, y + 0.01 * i, "matern52", device=i) for i in range(2)]
ref = []
for g in gs:
    for i in range(10):
        lp = g.update(lam, 1.0, 0.05)
    ref.append(lp)
REP = 300
def one(g, out, k):
    ts = []
    bad = 0
    for i in range(REP):
        t = time.perf_counter(); lp = g.update(lam, 1.0, 0.05); ts.append(time.perf_counter() - t)
        bad += lp != ref[k]
    out[k] = (np.median(ts) * 1e3, bad)
res = [None, None]
t0 = time.perf_counter(); one(gs[0], res, 0); t_single = time.perf_counter() - t0
print(f"one handle alone: {REP} updates in {t_single * 1e3:.1f} ms = {t_single / REP * 1e3:.3f} ms each (p50 {res[0][0]:.3f})", flush=True)
th = [threading.Thread(target=one, args=(gs[k], res, k)) for k in range(2)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
t_both = time.perf_counter() - t0
print(f"two handles on two contexts, two threads: 2 x {REP} updates in {t_both * 1e3:.1f} ms = {t_both / REP * 1e3:.3f} ms per pair (p50 per call {res[0][0]:.3f} / {res[1][0]:.3f}), results changed: {res[0][1]} / {res[1][1]}", flush=True)
for dev in range(2):
    n, code = C.c_long(0), C.c_int(0)
    lib.boss_debug_fallbacks(dev, C.byref(n), C.byref(code))
    print(f"  context {dev}: fallbacks {n.value} (last code {code.value})")

"""2000+ back-to-back N = 4096 updates: latency distribution, and (BOSS_LAUNCH_STAMPS=1) where the slow ones lost their time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
entry.build()
from boss_jl_amd import api
N, D = int(os.environ.get("N", 4096)), 8
rng = np.random.default_rng(1)
X = rng.uniform(0, 1, (D, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(D) + 0.05 * rng.standard_normal(N)
g = api.GP(X, y, "matern52"); lam = np.full(D, 0.5)
n = int(os.environ.get("REP", 2000))
import ctypes as C
lib = api.load_library()
lib.boss_debug_launch_stamps.argtypes = [C.c_int, C.c_double]
lib.boss_debug_stall_report.argtypes = [C.c_char_p, C.c_int]
rep = C.create_string_buffer(16384)
for i in range(20):
    g.update(lam, 1.0, 0.05)
lib.boss_debug_launch_stamps(1, 2.0)                                    # host-side stamps per launch; updates beyond 2 ms leave a report
lib.boss_debug_stall_report(rep, 16384)
ts = np.empty(n)
for i in range(n):
    t0 = time.perf_counter(); g.update(lam, 1.0, 0.05 + 1e-4 * (i % 7)); ts[i] = time.perf_counter() - t0
ts *= 1e3
slow = np.flatnonzero(ts > 2 * np.median(ts))
print(f"n {n} p50 {np.median(ts):.3f} p99 {np.percentile(ts,99):.3f} p999 {np.percentile(ts,99.9):.3f} max {ts.max():.3f} mean {ts.mean():.3f} ms; updates > 2 x p50: {len(slow)} at {slow[:40].tolist()} -> {np.round(ts[slow][:40],2).tolist()}")
lib.boss_debug_stall_report(rep, 16384)
lib.boss_debug_launch_stamps(0, 0.0)
for ln in rep.value.decode(errors="replace").splitlines():
    if ln:
        print("  stall report:", ln)

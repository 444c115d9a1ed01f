"""Trailing-update schedules of the chain side by side: python tools/trail_ab.py N [reps]
Runs in ONE process per mode (the mode is read once): spawns itself with BOSS_CHAIN_TRAIL=<mode> --child."""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--child" in sys.argv:
    import numpy as np
    from boss_jl_amd import api
    api.load_library()
    N = int(sys.argv[1]); reps = int(sys.argv[2])
    rng = np.random.default_rng(N); d = 8
    X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    g = api.GP(X, y, "matern52"); lam = np.full(d, 0.5)
    lps = []
    for i in range(8):
        lps.append(g.update(lam, 1.0, 0.05 + 1e-4 * (i % 4)))
    ts = []
    for i in range(reps):
        t = time.perf_counter(); lp = g.update(lam, 1.0, 0.05 + 1e-4 * (i % 4)); ts.append(time.perf_counter() - t)
        if lp.hex() != lps[i % 4].hex(): print("MISMATCH", i, lp, lps[i % 4], flush=True)
    ts.sort()
    print(f"mode {os.environ.get('BOSS_CHAIN_TRAIL')} N={N}: p50 {ts[len(ts)//2]*1e3:.4f} ms min {ts[0]*1e3:.4f} max {ts[-1]*1e3:.4f}  logpdf {lps[0]!r}", flush=True)
    sys.exit(0)
N = sys.argv[1]; reps = sys.argv[2] if len(sys.argv) > 2 else "200"
modes = os.environ.get("MODES", "0,1").split(",")   # (0: one launch per step, 1: main-stream pairs, 2: two-stream gated; mode 3 was a removed experiment)
for m in modes:
    env = dict(os.environ, BOSS_CHAIN_TRAIL=m, PYTHONUNBUFFERED="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), N, reps, "--child"], env=env, timeout=300)
    if r.returncode != 0:
        print("mode", m, "exit", r.returncode, flush=True); sys.exit(r.returncode)

"""The round-3 CU-mask experiment again (csrc/bosship.hip, BOSS_CU_MASK): the resident chain and strips on reserved CUs, the update's
other kernels on the complement.  `record` (unmasked) stores logpdf and a hash of the whole factor for a cycle of updates; `check`
(run with BOSS_CU_MASK=1 or 2) repeats them and compares bit for bit.
python tools/cumask_check.py record|check [reps]"""
import os, sys, json, hashlib, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boss_jl_amd import api
mode = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 70
REF = "/tmp/cumask_ref.json"
NH = int(os.environ.get("NHASH", 7))
LONG = int(os.environ.get("LONG", 300))
lib = api.load_library()
lib.boss_debug_fallbacks.argtypes = [C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_int)]
out = {}
bad = 0
ref = json.load(open(REF)) if mode == "check" else None
for N in [int(v) for v in os.environ.get("NS", "1408,4096").split(",")]:
    rng = np.random.default_rng(N)
    d = 8
    X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
    g = api.GP(X, y, "matern52")
    lam = np.full(d, 0.5)
    rec = []
    t0 = time.perf_counter()
    for i in range(reps):
        lp = g.update(lam, 1.0, 0.05 + 1e-3 * (i % 7))
        h = ""
        if i < NH or i >= reps - NH:
            L, z = g.factor()
            h = hashlib.sha256(np.ascontiguousarray(L).tobytes() + np.ascontiguousarray(z).tobytes()).hexdigest()
            if mode == "record" and i == 0:
                np.save(f"/tmp/cumask_L_{N}.npy", L)
            if mode == "check" and i == 0:
                L0 = np.load(f"/tmp/cumask_L_{N}.npy")
                print(f"N {N}: max |L - L_unmasked| at the first update {np.abs(L - L0).max():.3e}, finite {np.isfinite(L).all()}", flush=True)
        rec.append([float(lp).hex(), h])
        if ref is not None:
            want = ref[str(N)][i]
            if want[0] != rec[-1][0] or (h and want[1] != h):
                bad += 1
                if bad <= 10:
                    print(f"N {N} update {i}: logpdf {lp!r} against {float.fromhex(want[0])!r}, factor hash {'differs' if h and want[1] != h else 'equal / not taken'}", flush=True)
    dt = (time.perf_counter() - t0) / reps
    # a longer run without downloads: logpdf only, then the fused update + acquisition (its rider runs on the side stream)
    ts = []
    for i in range(LONG):
        t1 = time.perf_counter()
        lp = g.update(lam, 1.0, 0.05 + 1e-3 * (i % 7))
        ts.append(time.perf_counter() - t1)
        rec.append([float(lp).hex(), ""])
    cand = api.Candidates(np.random.default_rng(5).uniform(0, 1, (d, 1024)))
    tf = []
    nf = 0
    for i in range(LONG // 3):
        t1 = time.perf_counter()
        r = g.update_acq(lam, 1.0, 0.05 + 1e-3 * (i % 7), cand, best=float(y.max()))
        tf.append(time.perf_counter() - t1)
        nf += int(r["fused"])
        rec.append([float(r["logpdf"]).hex(), f"{r['argmax']}:{float(r['max']).hex()}"])
    if ref is not None:
        for i in range(reps, len(rec)):
            if ref[str(N)][i] != rec[i]:
                bad += 1
                if bad <= 10:
                    print(f"N {N} call {i}: {rec[i]} against {ref[str(N)][i]}", flush=True)
    print(f"N {N}: {LONG} updates p50 {np.median(ts) * 1e3:.3f} ms max {np.max(ts) * 1e3:.3f}; {LONG // 3} update_acq (1024 candidates) p50 {np.median(tf) * 1e3:.3f} ms, fused {nf}", flush=True)
    n, code = C.c_long(0), C.c_int(0)
    lib.boss_debug_fallbacks(0, C.byref(n), C.byref(code))
    print(f"N {N}: {reps} updates, {dt * 1e3:.3f} ms each incl. factor downloads; fallbacks so far {n.value} (last code {code.value})", flush=True)
    out[str(N)] = rec
    g.close()
if mode == "record":
    json.dump(out, open(REF, "w"))
    print("recorded", REF)
else:
    print("MISMATCHES:", bad)
    sys.exit(1 if bad else 0)

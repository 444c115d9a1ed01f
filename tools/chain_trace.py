"""True device-side timeline of one posterior update (in-kernel s_memrealtime stamps of workgroup 0 of the chain
kernels; build the instrumented side library first: python tools/build_trace_lib.py): where does the time between chain kernels go?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
lib = api.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbosship_tr.so"))
lib.boss_debug_trace.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint), C.c_int]
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (8, 4096)); y = np.sin(X).sum(0)
g = api.GP(X, y, "matern52")
for _ in range(3): g.update(np.full(8, .5), 1.0, 0.05)
buf = (C.c_ulonglong * 4096)(); n = C.c_uint(0)
lib.boss_debug_trace(buf, C.byref(n), 1)
g.update(np.full(8, .5), 1.0, 0.05)
api.device_sync(0) if hasattr(api, "device_sync") else None
lib.boss_debug_trace(buf, C.byref(n), 1)
recs = sorted([(buf[i + 1], buf[i + 2], int(buf[i])) for i in range(0, n.value, 3)])
t0 = recs[0][0]
names = {1: "diag", 2: "trsm", 3: "colupd", 4: "syrk"}
chain = [r for r in recs if r[2] != 4]
prev = None
tot = {"diag": 0, "trsm": 0, "colupd": 0, "gap": 0}
for s, e, c in chain:
    gap = (s - prev) / 100.0 if prev else 0.0
    tot[names[c]] += (e - s) / 100.0
    tot["gap"] += gap
    prev = e
print("records", len(recs), " span %.1f us" % ((recs[-1][1] - t0) / 100.0))
print("chain sums (us):", {k: round(v, 1) for k, v in tot.items()})
print("step: diag_dur | gap->trsm trsm_wg0 | gap->colupd colupd_wg0 | gap->next diag   (us)")
i = 0
step = 0
while i + 2 < len(chain) and chain[i][2] == 1 and chain[i + 1][2] == 2 and chain[i + 2][2] == 3:
    d_, t_, c_ = chain[i], chain[i + 1], chain[i + 2]
    nxt = chain[i + 3][0] if i + 3 < len(chain) else c_[1]
    print(f"{step:3d}: {(d_[1]-d_[0])/100:5.1f} | {(t_[0]-d_[1])/100:5.1f} {(t_[1]-t_[0])/100:5.1f} | {(c_[0]-t_[1])/100:5.1f} {(c_[1]-c_[0])/100:5.1f} | {(nxt-c_[1])/100:5.1f}")
    i += 3
    step += 1
syrk = [r for r in recs if r[2] == 4]
print("syrk wg0 durations (us):", " ".join(f"{(e-s)/100:.0f}" for s, e, c in syrk))

import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
lib = api.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbosship_clk.so"))
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (8, 4096)); y = np.sin(X).sum(0)
g = api.GP(X, y, "matern52")
os.environ["BOSS_NO_LOOKAHEAD"] = "1"
for _ in range(3): g.update(np.full(8, .5), 1.0, 0.05)
buf = (C.c_ulonglong * 64)()
lib.boss_debug_diag_clk.argtypes = [C.POINTER(C.c_ulonglong)]
lib.boss_debug_diag_clk(buf)
t = np.array(buf[:], dtype=np.float64)
t = t[t > 0]
d = np.diff(t)   # s_memtime ticks = shader cycles
print("events:", len(t), " total cycles %.0f" % (t[-1] - t[0]))
print("deltas (cycles):", " ".join(f"{x:.0f}" for x in d))

"""One-off robustness check at sizes beyond the benchmark (index arithmetic, workspaces): N = 16384."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boss_jl_amd import api
from oracle import gp_oracle as O
rng = np.random.default_rng(0)
d, N, M = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 600
X = rng.uniform(0, 1, (d, N)); y = np.sin(2 * np.pi * X).sum(0) / np.sqrt(d) + 0.05 * rng.standard_normal(N)
Xs = rng.uniform(0, 1, (d, M)); lam = np.full(d, 0.5)
g = api.GP(X, y, "matern52")
t = time.time(); lp = g.update(lam, 1.0, 0.05); tu = time.time() - t
t = time.time(); lp = g.update(lam, 1.0, 0.05); tu = time.time() - t
t = time.time(); mu, var = g.predict(Xs); tp = time.time() - t
cand = api.Candidates(rng.uniform(0, 1, (d, 8192)))
api.acq_ei([[g]], cand, [1.0], None, 1.0, want_acq=False)
t = time.time(); api.acq_ei([[g]], cand, [1.0], None, 1.0, want_acq=False); ta = time.time() - t
_, _, dmu, dvar = g.predict_grad(Xs[:, :64])
lpg, grad = g.loglike_grad()
t = time.time(); post = O.gp_fit(X, y, "matern52", lam, 1.0, 0.05); tc = time.time() - t
mu_o, var_o, dmu_o, dvar_o = O.gp_mean_and_var_grad(post, Xs[:, :64])
print(f"N={N}: update {tu*1e3:.1f} ms ({N**3/3/tu/1e12:.1f} TF), predict({M}) {tp*1e3:.1f} ms, acq(8192) {ta*1e3:.1f} ms "
      f"({8192*N*N/ta/1e12:.1f} TF); cpu fit {tc:.1f} s; |dlogpdf|={abs(lp-post.logpdf)/(1+abs(post.logpdf)):.1e} "
      f"|dmu|={np.abs(mu[:64]-mu_o).max():.1e} |dvar|={np.abs(var[:64]-O.clip_var(var_o)).max():.1e} "
      f"|ddmu|={np.abs(dmu-dmu_o).max():.1e} |ddvar|={np.abs(dvar-dvar_o).max():.1e}", flush=True)
lam2 = lam.copy(); eps = 1e-5
fp = g.update(lam, 1.0 + eps, 0.05); fm = g.update(lam, 1.0 - eps, 0.05)
print(f"   dlogpdf/dalpha: analytic {grad[d]:.6f}  central difference {(fp-fm)/(2*eps):.6f}", flush=True)

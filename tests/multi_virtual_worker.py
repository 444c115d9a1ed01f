"""Worker of tests/test_gpu_parity.py::test_multi_device_entry_points_on_virtual_devices: the one-process multi-device entry
points (csrc/host_multi.inc) with G = BOSS_VIRTUAL_DEVICES > 1 contexts on one physical GPU — shard arithmetic, the mean_Xs
re-layout, owners on different devices, ties / NaN across shards, an unfitted replica, fewer hyper-parameter sets than devices.
Every result is held against the single-device entry points and the CPU oracle.  Prints 'VIRTUAL-OK G' at the end."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boss_jl_amd import api                     # noqa: E402
from oracle import gp_oracle as O               # noqa: E402

G = int(os.environ["BOSS_VIRTUAL_DEVICES"])
api.load_library()
assert api.device_count() == G
assert api.init() == G
ndev, rccl = api.comm_info()
assert ndev == G and not rccl                   # (a communicator cannot hold one device twice: host-path exchanges)

rng = np.random.default_rng(40 + G)
d, N, M, P, S = 3, 150, 211, 2, 3               # M is not a multiple of G; S = 3 samples (split 2 + 1 at G = 2, 1 + 1 + 1 at G = 3)
X = rng.uniform(0, 1, (d, N))
Y = np.stack([np.sin(3 * X).sum(0), X[0] - X[1] ** 2]) + 0.02 * rng.standard_normal((P, N))
Xs = rng.uniform(0, 1, (d, M))
coefs, y_max = [1.0, 0.3], np.array([np.inf, 0.4])
lam = [rng.uniform(0.3, 0.8, (P, d)) for _ in range(S)]
amp = [rng.uniform(0.8, 1.4, P) for _ in range(S)]
mean_fn = [lambda x: 0.1 * x.sum(0), lambda x: -0.2 * x[0]]
mX = [f(X) for f in mean_fn]
mS = np.stack([np.stack([f(Xs) for f in mean_fn]) for _ in range(S)])        # [S][P][M]
mask = rng.uniform(size=M) > 0.2
b = O.best_so_far(coefs, Y, y_max)

posts = [[O.gp_fit(X, Y[p], "matern52", lam[s][p], amp[s][p], 0.05, mean=mX[p]) for p in range(P)] for s in range(S)]
want = np.mean([O.ei_acquisition(posts[s], Xs, coefs, y_max, b, means_s=[mS[s, 0], mS[s, 1]]) for s in range(S)], axis=0)
want = np.where(mask, want, 0.0)


def fitted(dev, s, p):
    g = api.GP(X, Y[p], "matern52", device=dev)
    g.update(lam[s][p], amp[s][p], 0.05, mX[p])
    return g


# ---- candidates mode: every device holds a replica of every posterior, the candidates are split M/G (ragged)
repl = [[[fitted(g, s, p) for p in range(P)] for s in range(S)] for g in range(G)]
acq, am, mx = api.multi_acq_ei(repl, Xs, coefs, y_max, b, mask, mS)
assert np.allclose(acq, want, rtol=0, atol=1e-10), np.abs(acq - want).max()
assert am == int(np.argmax(acq)) and mx == acq[am]
cand0 = api.Candidates(Xs, device=0)
single = api.acq_ei(repl[0], cand0, coefs, y_max, b, mask, mS)
assert np.allclose(acq, single[0], rtol=0, atol=1e-13) and (am, mx) == (single[1], single[2])
_, am2, mx2 = api.multi_acq_ei(repl, Xs, coefs, y_max, b, mask, mS, want_acq=False)
assert (am2, mx2) == (am, mx)
# resident candidate shards: uploaded once, used by several calls
mcand = api.MultiCandidates(Xs, G)
for _ in range(2):
    a_r, am_r, mx_r = api.multi_acq_ei_cand(repl, mcand, coefs, y_max, b, mask, mS)
    assert np.array_equal(a_r, acq) and (am_r, mx_r) == (am, mx)
_, am_r, mx_r = api.multi_acq_ei_cand(repl, mcand, coefs, y_max, b, mask, mS, want_acq=False)
assert (am_r, mx_r) == (am, mx)
mcand.close()
# more devices than candidates: empty shards take no part
a_e, am_e, mx_e = api.multi_acq_ei(repl, Xs[:, :1], coefs, y_max, b, None, mS[:, :, :1])
assert am_e == 0 and abs(a_e[0] - O.ei_acquisition(posts[0], Xs[:, :1], coefs, y_max, b, means_s=[mS[0, 0, :1], mS[0, 1, :1]])[0] / S
                         - sum(O.ei_acquisition(posts[s], Xs[:, :1], coefs, y_max, b, means_s=[mS[s, 0, :1], mS[s, 1, :1]])[0] for s in range(1, S)) / S) <= 1e-10
# fewer devices than the communicator holds (a subset): still the same
if G > 2:
    a2, am3, mx3 = api.multi_acq_ei(repl[:2], Xs, coefs, y_max, b, mask, mS)
    assert np.allclose(a2, acq, rtol=0, atol=1e-13) and (am3, mx3) == (am, mx)
# ties across shards: the first index wins wherever it lives — duplicate the best candidate into every shard
Xt = Xs.copy()
jbest = int(np.argmax(want))
lo = [g * (M // G) + min(g, M % G) for g in range(G)] + [M]
dup = sorted({lo[g] + (5 + g) % max(1, lo[g + 1] - lo[g]) for g in range(G)} | {jbest})
mt = mS.copy()
for j in dup:
    Xt[:, j] = Xs[:, jbest]
    mt[:, :, j] = mS[:, :, jbest]
mk = mask.copy()
mk[dup] = True
at, amt, mxt = api.multi_acq_ei(repl, Xt, coefs, y_max, b, mk, mt)
assert amt == min(dup) and amt == int(np.argmax(at)), (amt, dup)
# NaN counts as the largest value (Julia argmax): a NaN prior mean in the LAST shard must win over everything before it
mn = mS.copy()
jn = M - 2
mn[:, :, jn] = np.nan
an, amn, mxn = api.multi_acq_ei(repl, Xs, coefs, y_max, b, None, mn)
assert amn == jn and np.isnan(mxn) and np.isnan(an[jn]), (amn, mxn)
# an unfitted replica on device 1 is reported, not ignored
g_un = api.GP(X, Y[0], "matern52", device=1)
bad = [[[repl[g][s][p] for p in range(P)] for s in range(S)] for g in range(G)]
bad[1][0][0] = g_un
try:
    api.multi_acq_ei(bad, Xs, coefs, y_max, b, mask, mS)
    raise SystemExit("an unfitted replica was accepted")
except api.BossError as e:
    assert e.code == api.BOSS_E_NOT_FITTED, e
# a replica on the wrong device is refused
wrong = [[[repl[0][s][p] for p in range(P)] for s in range(S)] for g in range(G)]
try:
    api.multi_acq_ei(wrong, Xs, coefs, y_max, b, mask, mS)
    raise SystemExit("replicas on the wrong device were accepted")
except api.BossError as e:
    assert e.code == api.BOSS_E_INVALID, e
g_un.close()

# ---- outputs mode: output p of sample s lives on device (p + s) mod G only
own = [[fitted((p + s) % G, s, p) for p in range(P)] for s in range(S)]
a_o, am_o, mx_o = api.multi_acq_ei_outputs(own, Xs, coefs, y_max, b, mask, mS)
assert np.allclose(a_o, want, rtol=0, atol=1e-10) and am_o == int(np.argmax(a_o)) and mx_o == a_o[am_o]
assert np.allclose(a_o, acq, rtol=0, atol=1e-13)

# ---- samples mode: all outputs of sample s on device s mod G (2 + 1 at G = 2)
smp = [[fitted(s % G, s, p) for p in range(P)] for s in range(S)]
a_s, am_s, mx_s = api.multi_acq_ei_samples(smp, Xs, coefs, y_max, b, mask, mS)
assert np.allclose(a_s, want, rtol=0, atol=1e-10) and am_s == int(np.argmax(a_s)) and mx_s == a_s[am_s]
# outputs of one sample on different devices are refused in this mode
try:
    api.multi_acq_ei_samples(own, Xs, coefs, y_max, b, mask, mS)
    raise SystemExit("samples mode accepted a sample spread over devices")
except api.BossError as e:
    assert e.code == api.BOSS_E_INVALID, e

# ---- replicated update: the same hyper-parameters everywhere, concurrently
lp = api.multi_update([repl[g][0][0] for g in range(G)], lam[1][0], amp[1][0], 0.05, mX[0])
w = O.gp_fit(X, Y[0], "matern52", lam[1][0], amp[1][0], 0.05, mean=mX[0]).logpdf
assert abs(lp - w) <= 1e-10 * (1 + abs(w))
for g in range(G):
    mu, var = repl[g][0][0].predict(Xs[:, :9], mS[0, 0, :9])
    mu0, var0 = repl[0][0][0].predict(Xs[:, :9], mS[0, 0, :9])
    assert np.array_equal(mu, mu0) and np.array_equal(var, var0)

# ---- likelihood batches: more sets than devices (ragged) and FEWER sets than devices
for nset in (7, 1, G - 1):
    if nset < 1:
        continue
    lamS = np.exp(-0.7 + 0.3 * rng.standard_normal((d, nset)))
    ampS, sigS = np.exp(0.3 * rng.standard_normal(nset)), np.full(nset, 0.05)
    ll_m, st_m = api.multi_loglike_batch(G, X, Y[0], "matern52", lamS, ampS, sigS, mean_X=mX[0])
    ll_1, st_1 = api.loglike_batch(X, Y[0], "matern52", lamS, ampS, sigS, mean_X=mX[0])
    assert np.allclose(ll_m, ll_1, rtol=1e-13, atol=0) and np.array_equal(st_m, st_1), nset

for grp in (repl, [own], [smp]):
    for a in grp:
        for row in a:
            for h in row:
                h.close()
api.shutdown()
print("VIRTUAL-OK", G, flush=True)

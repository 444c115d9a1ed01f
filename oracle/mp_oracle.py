"""50-digit mpmath evaluation of the same GP formulas as gp_oracle.py (TEST INFRASTRUCTURE).

Independent of numpy/LAPACK rounding: used to pin the fp64 oracle (and, through it, the HIP path)
for small N.  Formulas: src/models/gradient_gp.jl:323-361,403 (explicit Cholesky algebra) and
src/models/gaussian_process.jl:216-248 (the +1e-8 parameter offsets, α²·κ(r/λ), σ²I noise).
"""
from __future__ import annotations

import mpmath as mp

MATERN32, MATERN52, SQEXP = 0, 1, 2


def _kappa(kernel, r):
    if kernel == MATERN32:
        s = mp.sqrt(3) * r
        return (1 + s) * mp.e ** (-s)
    if kernel == MATERN52:
        s = mp.sqrt(5) * r
        return (1 + s + s * s / 3) * mp.e ** (-s)
    if kernel == SQEXP:
        return mp.e ** (-r * r / 2)
    raise ValueError(kernel)


def _k(kernel, xa, xb, lam, amp):
    r2 = mp.mpf(0)
    for k in range(len(lam)):
        t = (xa[k] - xb[k]) / lam[k]
        r2 += t * t
    return amp * amp * _kappa(kernel, mp.sqrt(r2))


def posterior(X, y, kernel, lengthscale, amplitude, noise_std, Xs, mean_X=None, mean_s=None,
              dps: int = 50):
    """Returns (logpdf, mu[M], var[M]) as python floats rounded from `dps`-digit arithmetic.
    X: d×N nested list/array, Xs: d×M.  var includes the +1e-18 prediction jitter, unclipped."""
    mp.mp.dps = dps
    d = len(X)
    N = len(X[0])
    M = len(Xs[0])
    off = mp.mpf("1e-8")
    lam = [mp.mpf(float(l)) + off for l in lengthscale]
    amp = mp.mpf(float(amplitude)) + off
    sig = mp.mpf(float(noise_std)) + off
    cols = [[mp.mpf(float(X[k][j])) for k in range(d)] for j in range(N)]
    scol = [[mp.mpf(float(Xs[k][j])) for k in range(d)] for j in range(M)]
    K = mp.matrix(N, N)
    for i in range(N):
        for j in range(N):
            K[i, j] = _k(kernel, cols[i], cols[j], lam, amp)
        K[i, i] += sig * sig
    L = mp.cholesky(K)
    delta = mp.matrix([mp.mpf(float(y[j])) - (mp.mpf(float(mean_X[j])) if mean_X is not None else 0)
                       for j in range(N)])
    z = mp.lu_solve(L, delta)            # L is triangular; LU of it is itself
    a = mp.lu_solve(L.T, z)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(N))
    logpdf = -(N * mp.log(2 * mp.pi) + logdet + sum(z[i] * z[i] for i in range(N))) / 2
    mus, vars_ = [], []
    for j in range(M):
        ks = mp.matrix([_k(kernel, cols[i], scol[j], lam, amp) for i in range(N)])
        mu = sum(ks[i] * a[i] for i in range(N))
        if mean_s is not None:
            mu += mp.mpf(float(mean_s[j]))
        v = mp.lu_solve(L, ks)
        var = amp * amp - sum(v[i] * v[i] for i in range(N)) + mp.mpf("1e-18")
        mus.append(float(mu))
        vars_.append(float(var))
    return float(logpdf), mus, vars_

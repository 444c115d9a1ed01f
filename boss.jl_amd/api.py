"""ctypes binding of libbosship.so (include/bosship.h) — the executable twin of the Julia `ccall`
layer described in INTEGRATION.md.  There is NO CPU fallback: if the HIP library is missing or no
GPU is visible, compute calls raise.

Array conventions follow the reference (src/types/data.jl:10-12): X is d×N with one observation
per COLUMN.  numpy arrays are converted to Fortran order so that the memory the C ABI sees is the
same column-major block a Julia Matrix{Float64} would hand to `ccall`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BOSS_LIB_PATH") or os.path.join(_HERE, "libbosship.so")      # (BOSS_LIB_PATH: A/B timing of another build, tools/)

BOSS_OK, BOSS_E_INVALID, BOSS_E_NO_DEVICE, BOSS_E_NOT_PD, BOSS_E_NEG_VAR, BOSS_E_NOT_FITTED, BOSS_E_ALLOC = range(7)
KERNELS = {"matern32": 0, "matern52": 1, "sqexp": 2}
FIT_NO_SYNC = 1

_c_dp = C.POINTER(C.c_double)
_c_ucp = C.POINTER(C.c_ubyte)

# name -> (restype, argtypes); must list EVERY symbol include/bosship.h declares (tests check it)
SIGNATURES = {
    "boss_version": (C.c_char_p, []),
    "boss_last_error": (C.c_char_p, []),
    "boss_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "boss_set_stream": (C.c_int, [C.c_int, C.c_void_p]),
    "boss_device_sync": (C.c_int, [C.c_int]),
    "boss_gp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_ucp, C.POINTER(C.c_void_p)]),
    "boss_gp_update": (C.c_int, [C.c_void_p, _c_dp, C.c_double, C.c_double, _c_dp, C.c_int, _c_dp]),
    "boss_gp_sync": (C.c_int, [C.c_void_p, _c_dp]),
    "boss_ngp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_ucp, C.POINTER(C.c_void_p)]),
    "boss_ngp_update": (C.c_int, [C.c_void_p, _c_dp, _c_dp, _c_dp, _c_dp, C.c_int, _c_dp]),
    "boss_ngp_predict": (C.c_int, [C.c_void_p, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, C.POINTER(C.c_long)]),
    "boss_ggp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, C.POINTER(C.c_void_p)]),
    "boss_ggp_update": (C.c_int, [C.c_void_p, _c_dp, C.c_double, C.c_double, C.c_double, C.c_int, _c_dp]),
    "boss_ngp_loglike_grad": (C.c_int, [C.c_void_p, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]),
    "boss_ngp_append": (C.c_int, [C.c_void_p, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]),
    "boss_ggp_loglike_grad": (C.c_int, [C.c_void_p, _c_dp, _c_dp]),
    "boss_ggp_append": (C.c_int, [C.c_void_p, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp]),
    "boss_gp_fit": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, C.c_double, C.c_double,
                              _c_ucp, C.POINTER(C.c_void_p), _c_dp]),
    "boss_gp_set_y": (C.c_int, [C.c_void_p, _c_dp]),
    "boss_gp_loglike_grad": (C.c_int, [C.c_void_p, _c_dp, _c_dp]),
    "boss_gp_reserve": (C.c_int, [C.c_void_p, C.c_int]),
    "boss_gp_n": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "boss_gp_append": (C.c_int, [C.c_void_p, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp]),
    "boss_gp_free": (None, [C.c_void_p]),
    "boss_gp_get_factor": (C.c_int, [C.c_void_p, _c_dp, _c_dp]),
    "boss_gp_loglike_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, C.c_int, _c_ucp,
                                        C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, C.POINTER(C.c_int)]),
    "boss_gp_loglike_grad_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, C.c_int, _c_ucp,
                                             C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, C.POINTER(C.c_int)]),
    "boss_gp_fit_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, C.c_int, _c_ucp,
                                    C.c_int, _c_dp, _c_dp, _c_dp, C.POINTER(C.c_void_p), _c_dp, C.POINTER(C.c_int)]),
    "boss_gp_predict": (C.c_int, [C.c_void_p, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, C.POINTER(C.c_long)]),
    "boss_gp_predict_grad": (C.c_int, [C.c_void_p, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp,
                                       C.POINTER(C.c_long)]),
    "boss_gp_predict_cov": (C.c_int, [C.c_void_p, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, C.POINTER(C.c_long)]),
    "boss_cand_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _c_dp, C.POINTER(C.c_void_p)]),
    "boss_cand_free": (None, [C.c_void_p]),
    "boss_acq_ei": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, _c_dp, _c_dp, _c_dp, C.c_int,
                              C.c_double, _c_ucp, _c_dp, C.POINTER(C.c_long), _c_dp]),
    "boss_gp_update_acq": (C.c_int, [C.c_void_p, _c_dp, C.c_double, C.c_double, _c_dp, C.c_void_p, _c_dp, C.c_double, C.c_double,
                                     C.c_int, C.c_double, _c_ucp, _c_dp, _c_dp, _c_dp, _c_dp, C.POINTER(C.c_long), _c_dp,
                                     C.POINTER(C.c_int)]),
    "boss_acq_ei_moments": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, C.c_int,
                                      C.c_double, _c_ucp, _c_dp, C.POINTER(C.c_long), _c_dp]),
    "boss_ngp_predict_grad": (C.c_int, [C.c_void_p, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp,
                                        C.POINTER(C.c_long)]),
    "boss_acq_ei_grad_moments": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, C.c_int,
                                           C.c_double, _c_ucp, _c_dp, _c_dp]),
    "boss_acq_ei_grad": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, C.c_int,
                                   C.c_double, _c_ucp, _c_dp, _c_dp]),
    "boss_track_create": (C.c_int, [C.c_void_p, C.c_void_p, _c_dp, C.POINTER(C.c_void_p)]),
    "boss_track_free": (None, [C.c_void_p]),
    "boss_track_sync": (C.c_int, [C.c_void_p]),
    "boss_track_moments": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _c_dp, _c_dp]),
    "boss_acq_ei_tracks": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), _c_dp, _c_dp, C.c_int, C.c_double, _c_ucp,
                                     _c_dp, C.POINTER(C.c_long), _c_dp]),
    "boss_init": (C.c_int, [C.POINTER(C.c_int)]),
    "boss_comm_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "boss_shutdown": (None, []),
    "boss_multi_gp_update": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), _c_dp, C.c_double, C.c_double, _c_dp, _c_dp]),
    "boss_multi_acq_ei": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, C.c_int,
                                    C.c_double, _c_ucp, _c_dp, C.POINTER(C.c_long), _c_dp]),
    "boss_multi_cand_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _c_dp, C.POINTER(C.c_void_p)]),
    "boss_multi_cand_free": (None, [C.c_void_p]),
    "boss_multi_acq_ei_cand": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, _c_dp, _c_dp, _c_dp, C.c_int,
                                         C.c_double, _c_ucp, _c_dp, C.POINTER(C.c_long), _c_dp]),
    "boss_multi_acq_ei_outputs": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, C.c_int,
                                            C.c_double, _c_ucp, _c_dp, C.POINTER(C.c_long), _c_dp]),
    "boss_multi_acq_ei_samples": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, _c_dp, _c_dp, _c_dp, _c_dp, C.c_int,
                                            C.c_double, _c_ucp, _c_dp, C.POINTER(C.c_long), _c_dp]),
    "boss_multi_loglike_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, C.c_int, _c_ucp, C.c_int,
                                           _c_dp, _c_dp, _c_dp, _c_dp, C.POINTER(C.c_int)]),
    "boss_bench_mfma_f64": (C.c_int, [C.c_int, C.c_int, _c_dp]),
    "boss_prof_enable": (C.c_int, [C.c_int, C.c_int]),
    "boss_prof_reset": (C.c_int, [C.c_int]),
    "boss_prof_get": (C.c_int, [C.c_int, C.c_char_p, _c_dp, C.POINTER(C.c_long)]),
}


class BossError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[bosship status {code}] {msg}")
        self.code = code


class PosDefException(BossError):
    """LinearAlgebra.PosDefException analogue (BOSS_E_NOT_PD)."""


class DomainError(BossError):
    """DomainError of `_clip_var` (src/models/gaussian_process.jl:186-194) (BOSS_E_NEG_VAR)."""

    bad_index: int = -1


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libbosship.so and attach prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(bosship has no CPU fallback)")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("BOSS_LIB_PATH") and not hasattr(lib, name):
            continue                                   # (an older build under A/B timing lacks the newer entry points)
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc: int):
    if rc == BOSS_OK:
        return
    msg = load_library().boss_last_error().decode()
    if rc == BOSS_E_NOT_PD:
        raise PosDefException(rc, msg)
    if rc == BOSS_E_NEG_VAR:
        raise DomainError(rc, msg)
    raise BossError(rc, msg)


def _f64(a, ndim=None):
    a = np.asfortranarray(a, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise ValueError(f"expected a {ndim}-d array, got shape {a.shape}")
    return a


def _dp(a):
    return None if a is None else a.ctypes.data_as(_c_dp)


def _ucp(a):
    return None if a is None else a.ctypes.data_as(_c_ucp)


def _kernel_id(kernel) -> int:
    return KERNELS[kernel] if isinstance(kernel, str) else int(kernel)


def device_count() -> int:
    n = C.c_int(0)
    load_library().boss_device_count(C.byref(n))
    return n.value


def set_stream(device: int, stream_ptr: Optional[int]):
    _check(load_library().boss_set_stream(device, C.c_void_p(stream_ptr or 0)))


def device_sync(device: int = 0):
    _check(load_library().boss_device_sync(device))


class GP:
    """One output slice's posterior, resident on a GPU (boss_gp_t)."""

    def __init__(self, X, y, kernel="matern52", discrete=None, device: int = 0):
        lib = load_library()
        X = _f64(X, 2)
        y = _f64(np.asarray(y).reshape(-1), 1)
        self.d, self.N = X.shape
        if y.shape[0] != self.N:
            raise ValueError("y must have one entry per column of X")
        self.device = device
        self.kernel = _kernel_id(kernel)
        disc = None if discrete is None else np.ascontiguousarray(np.asarray(discrete, dtype=bool).astype(np.uint8))
        h = C.c_void_p()
        _check(lib.boss_gp_create(device, self.kernel, self.d, self.N, _dp(X), _dp(y), _ucp(disc), C.byref(h)))
        self._h = h
        self.logpdf = None

    def update(self, lengthscale, amplitude, noise_std, mean_X=None, sync: bool = True) -> Optional[float]:
        lib = load_library()
        lam = _f64(np.asarray(lengthscale).reshape(-1), 1)
        if lam.shape[0] != self.d:
            # gaussian_process.jl:233 @assert length(lengthscales) == size(X, 1)
            raise BossError(BOSS_E_INVALID, "length(lengthscales) must equal x_dim")
        m = None if mean_X is None else _f64(np.asarray(mean_X).reshape(-1), 1)
        if m is not None and m.shape[0] != self.N:
            raise ValueError("mean_X must have N entries")
        out = C.c_double(0.0)
        rc = lib.boss_gp_update(self._h, _dp(lam), float(amplitude), float(noise_std), _dp(m),
                                0 if sync else FIT_NO_SYNC, C.byref(out))
        _check(rc)
        if sync:
            self.logpdf = out.value
            return out.value
        return None

    def update_acq(self, lengthscale, amplitude, noise_std, cand: "Candidates", fit_coef: float = 1.0, y_max: Optional[float] = None,
                   best=None, mean_X=None, mean_Xs=None, valid_mask=None, want_acq: bool = False, want_moments: bool = False):
        """One BO iteration's posterior update with its first acquisition riding along the factorisation
        (boss_gp_update_acq): `update` followed by `acq_ei([[self]], cand, [fit_coef], [y_max], best)` in one call
        (y_max None: `constraints === nothing`; +Inf: a constraint that always holds).
        Returns a dict: logpdf, argmax, max, fused (True when the substitution rode along), and acq / mu / var on request
        (mu, var unclipped)."""
        lam = _f64(np.asarray(lengthscale).reshape(-1), 1)
        if lam.shape[0] != self.d:
            raise BossError(BOSS_E_INVALID, "length(lengthscales) must equal x_dim")
        m = None if mean_X is None else _f64(np.asarray(mean_X).reshape(-1), 1)
        if m is not None and m.shape[0] != self.N:
            raise ValueError("mean_X must have N entries")
        M = cand.M
        ms = None if mean_Xs is None else _f64(np.asarray(mean_Xs).reshape(-1), 1)
        if ms is not None and ms.shape[0] != M:
            raise ValueError("mean_Xs must have one entry per candidate")
        mask = None if valid_mask is None else np.ascontiguousarray(np.asarray(valid_mask, dtype=bool).astype(np.uint8))
        if mask is not None and mask.shape[0] != M:
            raise ValueError("valid_mask must have one entry per candidate")
        acq = np.zeros(M) if want_acq else None
        mu = np.zeros(M) if want_moments else None
        var = np.zeros(M) if want_moments else None
        lp, am, mx, fused = C.c_double(0.0), C.c_long(-1), C.c_double(0.0), C.c_int(0)
        _check(load_library().boss_gp_update_acq(self._h, _dp(lam), float(amplitude), float(noise_std), _dp(m), cand._h, _dp(ms),
                                                 float(fit_coef), float("nan") if y_max is None else float(y_max), 0 if best is None else 1,
                                                 0.0 if best is None else float(best), _ucp(mask), C.byref(lp), _dp(mu), _dp(var),
                                                 _dp(acq), C.byref(am), C.byref(mx), C.byref(fused)))
        self.logpdf = lp.value
        return {"logpdf": lp.value, "argmax": am.value, "max": mx.value, "fused": bool(fused.value), "acq": acq, "mu": mu, "var": var}

    def sync(self) -> float:
        out = C.c_double(0.0)
        _check(load_library().boss_gp_sync(self._h, C.byref(out)))
        self.logpdf = out.value
        return out.value

    def set_y(self, y):
        y = _f64(np.asarray(y).reshape(-1), 1)
        _check(load_library().boss_gp_set_y(self._h, _dp(y)))

    def loglike_grad(self):
        """(logpdf, grad[d+2]) at the hyper-parameters of the last update: gradient w.r.t.
        (lengthscale[d], amplitude, noise_std), analytic, on the device."""
        out = C.c_double(0.0)
        grad = np.zeros(self.d + 2)
        _check(load_library().boss_gp_loglike_grad(self._h, C.byref(out), _dp(grad)))
        return out.value, grad

    def reserve(self, N_total: int):
        """Reserve storage for N_total observations (later appends need no re-allocation); the handle
        must be (re-)updated afterwards."""
        _check(load_library().boss_gp_reserve(self._h, int(N_total)))

    def append(self, X_new, y_new, mean_new=None) -> float:
        """augment_dataset! + model_posterior with unchanged hyper-parameters (block Cholesky
        append): X_new d×n (or a length-d vector), y_new n.  Returns the logpdf of all N+n points."""
        X_new = _f64(X_new)
        if X_new.ndim == 1:
            X_new = _f64(X_new.reshape(-1, 1))
        if X_new.shape[0] != self.d:
            raise ValueError("X_new must be d×n")
        n = X_new.shape[1]
        y_new = _f64(np.asarray(y_new).reshape(-1), 1)
        if y_new.shape[0] != n:
            raise ValueError("y_new must have one entry per new point")
        m = None if mean_new is None else _f64(np.asarray(mean_new).reshape(-1), 1)
        if m is not None and m.shape[0] != n:
            raise ValueError("mean_new must have one entry per new point")
        out = C.c_double(0.0)
        rc = load_library().boss_gp_append(self._h, n, _dp(X_new), _dp(y_new), _dp(m), C.byref(out))
        cnt = C.c_int(self.N)                # the device's own count: observations stay appended when the factorisation
        load_library().boss_gp_n(self._h, C.byref(cnt))   # fails, and a failed multi-append may have taken only some
        self.N = cnt.value
        _check(rc)
        self.logpdf = out.value
        return out.value

    def factor(self):
        L = np.zeros((self.N, self.N), order="F")
        z = np.zeros(self.N)
        _check(load_library().boss_gp_get_factor(self._h, _dp(L), _dp(z)))
        return L, z

    def predict(self, Xs, mean_Xs=None):
        """mean_and_var(post, X::Matrix): returns (mu[M], var[M]) with _clip_var applied."""
        Xs = _f64(Xs)
        if Xs.ndim == 1:
            Xs = _f64(Xs.reshape(-1, 1))
        if Xs.shape[0] != self.d:
            raise ValueError("candidates must be d×M")
        M = Xs.shape[1]
        ms = None if mean_Xs is None else _f64(np.asarray(mean_Xs).reshape(-1), 1)
        mu = np.zeros(M)
        var = np.zeros(M)
        bad = C.c_long(-1)
        rc = load_library().boss_gp_predict(self._h, M, _dp(Xs), _dp(ms), _dp(mu), _dp(var), C.byref(bad))
        if rc == BOSS_E_NEG_VAR:
            e = DomainError(rc, load_library().boss_last_error().decode())
            e.bad_index = bad.value
            raise e
        _check(rc)
        return mu, var

    def predict_grad(self, Xs, mean_Xs=None, mean_grad=None):
        """mean_and_var(post, X) and its gradient w.r.t. the candidates: returns
        (mu[M], var[M], dmu[d,M], dvar[d,M]) — analytic, evaluated on the device."""
        Xs = _f64(Xs)
        if Xs.ndim == 1:
            Xs = _f64(Xs.reshape(-1, 1))
        if Xs.shape[0] != self.d:
            raise ValueError("candidates must be d×M")
        M = Xs.shape[1]
        ms = None if mean_Xs is None else _f64(np.asarray(mean_Xs).reshape(-1), 1)
        mg = None if mean_grad is None else _f64(mean_grad, 2)
        if mg is not None and mg.shape != (self.d, M):
            raise ValueError("mean_grad must be d×M")
        mu, var = np.zeros(M), np.zeros(M)
        dmu, dvar = np.zeros((self.d, M), order="F"), np.zeros((self.d, M), order="F")
        bad = C.c_long(-1)
        rc = load_library().boss_gp_predict_grad(self._h, M, _dp(Xs), _dp(ms), _dp(mg), _dp(mu), _dp(var), _dp(dmu),
                                                 _dp(dvar), C.byref(bad))
        if rc == BOSS_E_NEG_VAR:
            e = DomainError(rc, load_library().boss_last_error().decode())
            e.bad_index = bad.value
            raise e
        _check(rc)
        return mu, var, dmu, dvar

    def predict_cov(self, Xs, mean_Xs=None):
        """mean_and_cov(post, X::Matrix): returns (mu[M], cov[M,M]) with the diagonal clipped."""
        Xs = _f64(Xs, 2)
        if Xs.shape[0] != self.d:
            raise ValueError("candidates must be d×M")
        M = Xs.shape[1]
        ms = None if mean_Xs is None else _f64(np.asarray(mean_Xs).reshape(-1), 1)
        mu = np.zeros(M)
        cov = np.zeros((M, M), order="F")
        bad = C.c_long(-1)
        rc = load_library().boss_gp_predict_cov(self._h, M, _dp(Xs), _dp(ms), _dp(mu), _dp(cov), C.byref(bad))
        if rc == BOSS_E_NEG_VAR:
            e = DomainError(rc, load_library().boss_last_error().decode())
            e.bad_index = bad.value
            raise e
        _check(rc)
        return mu, cov

    def close(self):
        if getattr(self, "_h", None):
            load_library().boss_gp_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GradGP(GP):
    """One output slice's posterior conditioned on values AND gradients (GradientGaussianProcess,
    src/models/gradient_gp.jl): X d×n, y n, dY d×n.  `N` is the size of the augmented system n(1+d)."""

    def __init__(self, X, y, dY, kernel="matern52", device: int = 0):
        lib = load_library()
        X = _f64(X, 2)
        y = _f64(np.asarray(y).reshape(-1), 1)
        dY = np.asfortranarray(np.asarray(dY, dtype=np.float64))
        self.d, self.n = X.shape
        if y.shape[0] != self.n or dY.shape != (self.d, self.n):
            raise ValueError("y must have n entries and dY must be d×n")
        self.N = self.n * (1 + self.d)
        self.device = device
        self.kernel = _kernel_id(kernel)
        h = C.c_void_p()
        _check(lib.boss_ggp_create(device, self.kernel, self.d, self.n, _dp(X), _dp(y),
                                   dY.ctypes.data_as(_c_dp), C.byref(h)))
        self._h = h
        self.logpdf = None

    def update(self, lengthscale, amplitude, noise_std, grad_noise_std, sync: bool = True) -> Optional[float]:
        lam = _f64(np.asarray(lengthscale).reshape(-1), 1)
        if lam.shape[0] != self.d:
            raise BossError(BOSS_E_INVALID, "length(lengthscales) must equal x_dim")
        out = C.c_double(0.0)
        _check(load_library().boss_ggp_update(self._h, _dp(lam), float(amplitude), float(noise_std), float(grad_noise_std),
                                              0 if sync else FIT_NO_SYNC, C.byref(out)))
        if sync:
            self.logpdf = out.value
            return out.value
        return None

    def loglike_grad(self):
        """(logpdf, grad[d+3]) at the parameters of the last update: gradient w.r.t. (lengthscale[d], amplitude, noise_std,
        grad_noise_std) — what ForwardDiff yields through data_loglike (gradient_gp.jl:367-397) inside OptimizationMAP."""
        out = C.c_double(0.0)
        grad = np.zeros(self.d + 3)
        _check(load_library().boss_ggp_loglike_grad(self._h, C.byref(out), _dp(grad)))
        return out.value, grad

    def append(self, X_new, y_new, dY_new) -> float:
        """augment_dataset! + the posterior at unchanged hyper-parameters (the augmented system is rebuilt and factorised, as in the
        reference): X_new d×m (or a length-d vector), y_new m, dY_new d×m.  Returns the logpdf of all n + m points."""
        X_new = _f64(np.asarray(X_new, dtype=np.float64).reshape(self.d, -1), 2)
        m = X_new.shape[1]
        y_new = _f64(np.asarray(y_new).reshape(-1), 1)
        dY_new = np.asfortranarray(np.asarray(dY_new, dtype=np.float64).reshape(self.d, -1))
        if y_new.shape[0] != m or dY_new.shape != (self.d, m):
            raise ValueError("y_new must have one entry and dY_new one column per new point")
        out = C.c_double(0.0)
        _check(load_library().boss_ggp_append(self._h, m, _dp(X_new), _dp(y_new), dY_new.ctypes.data_as(_c_dp), C.byref(out)))
        self.n += m
        self.N = self.n * (1 + self.d)
        self.logpdf = out.value
        return out.value


class GibbsGP(GP):
    """One output slice's posterior under the NonstationaryGP's Gibbs kernel (nonstationary_gp.jl:61-107): the
    latent λ(·), α(·), σ(·) are evaluated by the caller and passed as arrays."""

    def __init__(self, X, y, discrete=None, device: int = 0):
        lib = load_library()
        X = _f64(X, 2)
        y = _f64(np.asarray(y).reshape(-1), 1)
        self.d, self.N = X.shape
        if y.shape[0] != self.N:
            raise ValueError("y must have one entry per column of X")
        self.device = device
        self.kernel = None
        disc = None if discrete is None else np.ascontiguousarray(np.asarray(discrete, dtype=bool).astype(np.uint8))
        h = C.c_void_p()
        _check(lib.boss_ngp_create(device, self.d, self.N, _dp(X), _dp(y), _ucp(disc), C.byref(h)))
        self._h = h
        self.logpdf = None

    def update(self, lam_X, amp_X, noise_X, mean_X=None, sync: bool = True) -> Optional[float]:
        lam = _f64(lam_X, 2)
        amp = _f64(np.asarray(amp_X).reshape(-1), 1)
        noi = _f64(np.asarray(noise_X).reshape(-1), 1)
        if lam.shape != (self.d, self.N) or amp.shape[0] != self.N or noi.shape[0] != self.N:
            raise ValueError("lam_X must be d×N, amp_X and noise_X length N")
        m = None if mean_X is None else _f64(np.asarray(mean_X).reshape(-1), 1)
        if m is not None and m.shape[0] != self.N:
            raise ValueError("mean_X must have N entries")
        out = C.c_double(0.0)
        _check(load_library().boss_ngp_update(self._h, _dp(lam), _dp(amp), _dp(noi), _dp(m), 0 if sync else FIT_NO_SYNC,
                                              C.byref(out)))
        if sync:
            self.logpdf = out.value
            return out.value
        return None

    def loglike_grad(self):
        """(logpdf, dlam[d, N], damp[N], dnoise[N], dmean[N]): the log-likelihood of the last update and its partial derivatives w.r.t.
        the latent models' values at the training points (boss_ngp_loglike_grad)."""
        out = C.c_double(0.0)
        dlam = np.zeros((self.d, self.N), order="F")
        damp, dnoi, dmean = np.zeros(self.N), np.zeros(self.N), np.zeros(self.N)
        _check(load_library().boss_ngp_loglike_grad(self._h, C.byref(out), _dp(dlam), _dp(damp), _dp(dnoi), _dp(dmean)))
        return out.value, dlam, damp, dnoi, dmean

    def append(self, X_new, y_new, lam_new, amp_new, noise_new, mean_new=None) -> float:
        """augment_dataset! + the posterior with the latent models evaluated at the new points (lam_new d×m, amp_new m, noise_new m):
        rebuilt and factorised (boss_ngp_append).  Returns the logpdf of all N + m points."""
        X_new = _f64(np.asarray(X_new, dtype=np.float64).reshape(self.d, -1), 2)
        m = X_new.shape[1]
        y_new = _f64(np.asarray(y_new).reshape(-1), 1)
        lam = _f64(np.asarray(lam_new, dtype=np.float64).reshape(self.d, -1), 2)
        amp = _f64(np.asarray(amp_new).reshape(-1), 1)
        noi = _f64(np.asarray(noise_new).reshape(-1), 1)
        mn = None if mean_new is None else _f64(np.asarray(mean_new).reshape(-1), 1)
        if y_new.shape[0] != m or lam.shape != (self.d, m) or amp.shape[0] != m or noi.shape[0] != m or (mn is not None and mn.shape[0] != m):
            raise ValueError("one entry (column) per new point in y_new, lam_new, amp_new, noise_new, mean_new")
        out = C.c_double(0.0)
        _check(load_library().boss_ngp_append(self._h, m, _dp(X_new), _dp(y_new), _dp(lam), _dp(amp), _dp(noi), _dp(mn), C.byref(out)))
        self.N += m
        self.logpdf = out.value
        return out.value

    def predict(self, Xs, lam_Xs, amp_Xs, mean_Xs=None):
        """mean_and_var with _clip_var; lam_Xs d×M and amp_Xs M are λ(x*), α(x*)."""
        Xs = _f64(Xs)
        if Xs.ndim == 1:
            Xs = _f64(Xs.reshape(-1, 1))
        M = Xs.shape[1]
        lam = _f64(np.asarray(lam_Xs, dtype=np.float64).reshape(self.d, M), 2)
        amp = _f64(np.asarray(amp_Xs).reshape(-1), 1)
        if Xs.shape[0] != self.d or amp.shape[0] != M:
            raise ValueError("candidates must be d×M with lam_Xs d×M and amp_Xs M")
        ms = None if mean_Xs is None else _f64(np.asarray(mean_Xs).reshape(-1), 1)
        mu = np.zeros(M)
        var = np.zeros(M)
        bad = C.c_long(-1)
        rc = load_library().boss_ngp_predict(self._h, M, _dp(Xs), _dp(lam), _dp(amp), _dp(ms), _dp(mu), _dp(var), C.byref(bad))
        if rc == BOSS_E_NEG_VAR:
            e = DomainError(rc, load_library().boss_last_error().decode())
            e.bad_index = bad.value
            raise e
        _check(rc)
        return mu, var

    def predict_grad(self, Xs, lam_Xs, amp_Xs, dlam_Xs=None, damp_Xs=None, mean_Xs=None, mean_grad=None):
        """mean_and_var and its gradient w.r.t. the candidates (boss_ngp_predict_grad): dlam_Xs d×d×M with [l, m, j] = ∂λ_l/∂x_m at
        candidate j (None: constant λ), damp_Xs d×M (None: constant α).  Returns (mu[M], var[M], dmu[d,M], dvar[d,M])."""
        Xs = _f64(Xs)
        if Xs.ndim == 1:
            Xs = _f64(Xs.reshape(-1, 1))
        M = Xs.shape[1]
        lam = _f64(np.asarray(lam_Xs, dtype=np.float64).reshape(self.d, M), 2)
        amp = _f64(np.asarray(amp_Xs).reshape(-1), 1)
        if Xs.shape[0] != self.d or amp.shape[0] != M:
            raise ValueError("candidates must be d×M with lam_Xs d×M and amp_Xs M")
        dl = None if dlam_Xs is None else np.asfortranarray(np.asarray(dlam_Xs, dtype=np.float64).reshape(self.d, self.d, M))
        da = None if damp_Xs is None else _f64(np.asarray(damp_Xs, dtype=np.float64).reshape(self.d, M), 2)
        ms = None if mean_Xs is None else _f64(np.asarray(mean_Xs).reshape(-1), 1)
        mg = None if mean_grad is None else _f64(np.asarray(mean_grad, dtype=np.float64).reshape(self.d, M), 2)
        mu, var = np.zeros(M), np.zeros(M)
        dmu, dvar = np.zeros((self.d, M), order="F"), np.zeros((self.d, M), order="F")
        bad = C.c_long(-1)
        rc = load_library().boss_ngp_predict_grad(self._h, M, _dp(Xs), _dp(lam), _dp(amp), _dp(dl), _dp(da), _dp(ms), _dp(mg), _dp(mu),
                                                  _dp(var), _dp(dmu), _dp(dvar), C.byref(bad))
        if rc == BOSS_E_NEG_VAR:
            e = DomainError(rc, load_library().boss_last_error().decode())
            e.bad_index = bad.value
            raise e
        _check(rc)
        return mu, var, dmu, dvar


class Candidates:
    """A resident batch of candidate points (boss_cand_t)."""

    def __init__(self, Xs, device: int = 0):
        Xs = _f64(Xs, 2)
        self.d, self.M = Xs.shape
        self.device = device
        h = C.c_void_p()
        _check(load_library().boss_cand_create(device, self.d, self.M, _dp(Xs), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            load_library().boss_cand_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fit(X, y, kernel, lengthscale, amplitude, noise_std, mean_X=None, discrete=None, device: int = 0,
        reserve: int = 0) -> GP:
    """posterior_gp (gaussian_process.jl:199-211) through the one-shot boss_gp_fit entry point.
    reserve > 0: room for that many later appends (create + reserve + update instead)."""
    if reserve > 0:
        g = GP(X, y, kernel, discrete, device)
        g.reserve(g.N + reserve)
        g.update(lengthscale, amplitude, noise_std, mean_X)
        return g
    lib = load_library()
    X = _f64(X, 2)
    y = _f64(np.asarray(y).reshape(-1), 1)
    lam = _f64(np.asarray(lengthscale).reshape(-1), 1)
    d, N = X.shape
    if lam.shape[0] != d:
        raise BossError(BOSS_E_INVALID, "length(lengthscales) must equal x_dim")
    m = None if mean_X is None else _f64(np.asarray(mean_X).reshape(-1), 1)
    disc = None if discrete is None else np.ascontiguousarray(np.asarray(discrete, dtype=bool).astype(np.uint8))
    h = C.c_void_p()
    out = C.c_double(0.0)
    _check(lib.boss_gp_fit(device, _kernel_id(kernel), d, N, _dp(X), _dp(y), _dp(m), _dp(lam), float(amplitude),
                           float(noise_std), _ucp(disc), C.byref(h), C.byref(out)))
    g = GP.__new__(GP)
    g.d, g.N, g.device, g.kernel, g._h, g.logpdf = d, N, device, _kernel_id(kernel), h, out.value
    return g


def fit_batch(X, y, kernel, lengthscales, amplitudes, noise_stds, mean_X=None, discrete=None, device: int = 0):
    """S resident posteriors of one output slice from ONE batched factorisation (boss_gp_fit_batch): what
    `model_posterior.(Ref(model), params, Ref(data))` builds for the S samples of a BI fit (src/posterior.jl:15-19).
    lengthscales is d×S.  Returns (gps[S], logpdf[S], status[S]); members with status != 0 are unfitted handles."""
    X = _f64(X, 2)
    y = _f64(np.asarray(y).reshape(-1), 1)
    lam = _f64(lengthscales, 2)
    d, N = X.shape
    S = lam.shape[1]
    if lam.shape[0] != d:
        raise BossError(BOSS_E_INVALID, "lengthscales must be d×S")
    amp = _f64(np.asarray(amplitudes).reshape(-1), 1)
    sig = _f64(np.asarray(noise_stds).reshape(-1), 1)
    if amp.shape[0] != S or sig.shape[0] != S:
        raise BossError(BOSS_E_INVALID, "amplitudes and noise_stds must have one entry per set")
    if y.shape[0] != N:
        raise ValueError("y must have one entry per column of X")           # (as GP.__init__: the library copies N entries)
    if not (np.isfinite(amp).all() and np.isfinite(sig).all() and np.isfinite(lam).all()):
        raise BossError(BOSS_E_INVALID, "hyper-parameters must be finite")
    stride = 0
    m = None
    if mean_X is not None:
        m = np.asarray(mean_X, dtype=np.float64)
        if m.ndim == 2:
            if m.shape != (S, N):
                raise ValueError("mean_X must be S×N (one prior-mean row per set) or a vector of N entries")
            m = np.ascontiguousarray(m)
            stride = N
        else:
            m = np.ascontiguousarray(m.reshape(-1))
            if m.shape[0] != N:
                raise ValueError("mean_X must be S×N (one prior-mean row per set) or a vector of N entries")
    disc = None if discrete is None else np.ascontiguousarray(np.asarray(discrete, dtype=bool).astype(np.uint8))
    hs = (C.c_void_p * S)()
    ll = np.zeros(S)
    st = np.zeros(S, dtype=np.int32)
    _check(load_library().boss_gp_fit_batch(device, _kernel_id(kernel), d, N, _dp(X), _dp(y), _dp(m), stride, _ucp(disc), S,
                                            _dp(lam), _dp(amp), _dp(sig), hs, _dp(ll), st.ctypes.data_as(C.POINTER(C.c_int))))
    gps = []
    for s in range(S):
        g = GP.__new__(GP)
        g.d, g.N, g.device, g.kernel, g._h = d, N, device, _kernel_id(kernel), C.c_void_p(hs[s])
        g.logpdf = float(ll[s]) if st[s] == 0 else None
        gps.append(g)
    return gps, ll, st


def loglike_batch(X, y, kernel, lengthscales, amplitudes, noise_stds, mean_X=None, discrete=None, device: int = 0,
                  want_grad: bool = False):
    """S log marginal likelihoods on the same (X, y) slice; lengthscales is d×S.
    Returns (ll[S], status[S]); ll = -Inf where the matrix is not PD (safe_data_loglike).
    want_grad: additionally grad[(d+2), S] = ∂ll/∂(lengthscale[d], amplitude, noise_std) per set -> (ll, status, grad)."""
    X = _f64(X, 2)
    y = _f64(np.asarray(y).reshape(-1), 1)
    lam = _f64(lengthscales, 2)
    d, N = X.shape
    S = lam.shape[1]
    if lam.shape[0] != d:
        raise BossError(BOSS_E_INVALID, "lengthscales must be d×S")
    amp = _f64(np.asarray(amplitudes).reshape(-1), 1)
    sig = _f64(np.asarray(noise_stds).reshape(-1), 1)
    stride = 0
    m = None
    if mean_X is not None:
        m = np.asarray(mean_X, dtype=np.float64)
        if m.ndim == 2:          # S×N, row s = mean of set s  → contiguous rows
            m = np.ascontiguousarray(m)
            stride = N
        else:
            m = np.ascontiguousarray(m.reshape(-1))
    disc = None if discrete is None else np.ascontiguousarray(np.asarray(discrete, dtype=bool).astype(np.uint8))
    ll = np.zeros(S)
    st = np.zeros(S, dtype=np.int32)
    if want_grad:
        grad = np.zeros((d + 2, S), order="F")
        _check(load_library().boss_gp_loglike_grad_batch(device, _kernel_id(kernel), d, N, _dp(X), _dp(y), _dp(m), stride,
                                                         _ucp(disc), S, _dp(lam), _dp(amp), _dp(sig), _dp(ll), _dp(grad),
                                                         st.ctypes.data_as(C.POINTER(C.c_int))))
        return ll, st, grad
    _check(load_library().boss_gp_loglike_batch(device, _kernel_id(kernel), d, N, _dp(X), _dp(y), _dp(m), stride,
                                                _ucp(disc), S, _dp(lam), _dp(amp), _dp(sig), _dp(ll),
                                                st.ctypes.data_as(C.POINTER(C.c_int))))
    return ll, st


def acq_ei(gps: Sequence[Sequence[GP]], cand: Candidates, fit_coefs, y_max=None, best=None, valid_mask=None,
           mean_Xs=None, want_acq: bool = True):
    """EI·feas over resident candidates.  gps[s][p] = output p of hyper-parameter sample s.
    mean_Xs: None or array [S][P][M].  Returns (acq[M] or None, argmax, max)."""
    S = len(gps)
    P = len(gps[0])
    arr = (C.c_void_p * (P * S))()
    for s in range(S):
        for p in range(P):
            arr[p + P * s] = gps[s][p]._h
    coefs = _f64(np.asarray(fit_coefs).reshape(-1), 1)
    ym = None if y_max is None else _f64(np.asarray(y_max).reshape(-1), 1)
    mask = None if valid_mask is None else np.ascontiguousarray(np.asarray(valid_mask, dtype=bool).astype(np.uint8))
    M = cand.M
    ms = None
    if mean_Xs is not None:
        a = np.asarray(mean_Xs, dtype=np.float64).reshape(S, P, M)
        ms = np.ascontiguousarray(a.transpose(0, 2, 1))       # index p + P*(j + M*s)
    acq = np.zeros(M) if want_acq else None
    am = C.c_long(-1)
    mx = C.c_double(0.0)
    _check(load_library().boss_acq_ei(P, S, arr, cand._h, _dp(ms), _dp(coefs), _dp(ym), 0 if best is None else 1,
                                      0.0 if best is None else float(best), _ucp(mask), _dp(acq), C.byref(am),
                                      C.byref(mx)))
    return acq, am.value, mx.value


def acq_ei_grad_moments(mu, var, dmu, dvar, fit_coefs, y_max=None, best=None, valid_mask=None, device: int = 0):
    """EI × feasibility and its gradient w.r.t. the candidates from moments and moment gradients already on the host
    (boss_acq_ei_grad_moments): mu / var [P][M], dmu / dvar [P][d][M].  Returns (acq[M], dacq[d, M])."""
    mu = _f64(np.atleast_2d(mu), 2)
    var = _f64(np.atleast_2d(var), 2)
    P, M = mu.shape
    gm = np.asarray(dmu, dtype=np.float64).reshape(P, -1, M)
    gv = np.asarray(dvar, dtype=np.float64).reshape(P, -1, M)
    d = gm.shape[1]
    gm = np.ascontiguousarray(gm.transpose(0, 2, 1))          # [p][j*d + m]
    gv = np.ascontiguousarray(gv.transpose(0, 2, 1))
    coefs = _f64(np.asarray(fit_coefs).reshape(-1), 1)
    ym = None if y_max is None else _f64(np.asarray(y_max).reshape(-1), 1)
    mask = None if valid_mask is None else np.ascontiguousarray(np.asarray(valid_mask, dtype=bool).astype(np.uint8))
    acq = np.zeros(M)
    dacq = np.zeros((d, M), order="F")
    _check(load_library().boss_acq_ei_grad_moments(device, P, M, d, _dp(mu), _dp(var), _dp(gm), _dp(gv), _dp(coefs), _dp(ym),
                                                   0 if best is None else 1, 0.0 if best is None else float(best), _ucp(mask), _dp(acq),
                                                   _dp(dacq)))
    return acq, dacq


class Track:
    """Resident predictive state of one posterior at one candidate set (boss_track_t): after
    GP.append the moments are extended in O(N·M) instead of re-solved.  Keep `gp` and `cand` alive."""

    def __init__(self, gp: GP, cand: Candidates, mean_Xs=None):
        ms = None if mean_Xs is None else _f64(np.asarray(mean_Xs).reshape(-1), 1)
        if ms is not None and ms.shape[0] != cand.M:
            raise ValueError("mean_Xs must have one entry per candidate")
        h = C.c_void_p()
        _check(load_library().boss_track_create(gp._h, cand._h, _dp(ms), C.byref(h)))
        self._h, self.gp, self.cand, self.M = h, gp, cand, cand.M

    def sync(self):
        _check(load_library().boss_track_sync(self._h))

    def moments(self, first: int = 0, count: Optional[int] = None):
        """(mu, var) of candidates [first, first+count) — var unclipped (apply _clip_var yourself)."""
        count = self.M - first if count is None else count
        mu, var = np.zeros(count), np.zeros(count)
        _check(load_library().boss_track_moments(self._h, first, count, _dp(mu), _dp(var)))
        return mu, var

    def close(self):
        if getattr(self, "_h", None):
            load_library().boss_track_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def acq_ei_tracks(tracks: Sequence[Sequence[Track]], fit_coefs, y_max=None, best=None, valid_mask=None,
                  want_acq: bool = True):
    """acq_ei on tracked states: tracks[s][p] = output p of hyper-parameter sample s."""
    S, P = len(tracks), len(tracks[0])
    arr = (C.c_void_p * (P * S))()
    for s in range(S):
        for p in range(P):
            arr[p + P * s] = tracks[s][p]._h
    M = tracks[0][0].M
    coefs = _f64(np.asarray(fit_coefs).reshape(-1), 1)
    ym = None if y_max is None else _f64(np.asarray(y_max).reshape(-1), 1)
    mask = None if valid_mask is None else np.ascontiguousarray(np.asarray(valid_mask, dtype=bool).astype(np.uint8))
    acq = np.zeros(M) if want_acq else None
    am = C.c_long(-1)
    mx = C.c_double(0.0)
    _check(load_library().boss_acq_ei_tracks(P, S, arr, _dp(coefs), _dp(ym), 0 if best is None else 1,
                                             0.0 if best is None else float(best), _ucp(mask), _dp(acq), C.byref(am),
                                             C.byref(mx)))
    return acq, am.value, mx.value


def acq_ei_grad(gps: Sequence[GP], Xs, fit_coefs, y_max=None, best=None, valid_mask=None, mean_Xs=None, mean_grad=None):
    """EI·feas and its gradient w.r.t. the candidates for one hyper-parameter sample.  gps[p] = output p.
    mean_Xs: None or [P][M]; mean_grad: None or [P][d][M].  Returns (acq[M], dacq[d, M])."""
    P = len(gps)
    Xs = _f64(Xs, 2)
    d, M = Xs.shape
    arr = (C.c_void_p * P)()
    for p in range(P):
        arr[p] = gps[p]._h
    coefs = _f64(np.asarray(fit_coefs).reshape(-1), 1)
    ym = None if y_max is None else _f64(np.asarray(y_max).reshape(-1), 1)
    mask = None if valid_mask is None else np.ascontiguousarray(np.asarray(valid_mask, dtype=bool).astype(np.uint8))
    ms = None if mean_Xs is None else np.ascontiguousarray(np.asarray(mean_Xs, dtype=np.float64).reshape(P, M))
    mg = None
    if mean_grad is not None:
        a = np.asarray(mean_grad, dtype=np.float64).reshape(P, d, M)
        mg = np.ascontiguousarray(a.transpose(0, 2, 1))           # [p][j*d + m]
    acq = np.zeros(M)
    dacq = np.zeros((d, M), order="F")
    _check(load_library().boss_acq_ei_grad(P, arr, M, _dp(Xs), _dp(ms), _dp(mg), _dp(coefs), _dp(ym),
                                           0 if best is None else 1, 0.0 if best is None else float(best), _ucp(mask),
                                           _dp(acq), _dp(dacq)))
    return acq, dacq


def acq_ei_moments(mu, var, fit_coefs, y_max=None, best=None, valid_mask=None, device: int = 0):
    """EI·feas + arg-max from posterior moments already on the host (outputs fitted on other
    ranks).  mu, var: [S][P][M] (or [P][M] for S = 1).  Returns (acq[M], argmax, max)."""
    mu = np.asarray(mu, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    if mu.ndim == 2:
        mu, var = mu[None], var[None]
    S, P, M = mu.shape
    assert var.shape == mu.shape
    mu, var = np.ascontiguousarray(mu), np.ascontiguousarray(var)
    coefs = _f64(np.asarray(fit_coefs).reshape(-1), 1)
    assert coefs.shape[0] == P
    ym = None if y_max is None else _f64(np.asarray(y_max).reshape(-1), 1)
    mask = None if valid_mask is None else np.ascontiguousarray(np.asarray(valid_mask, dtype=bool).astype(np.uint8))
    acq = np.zeros(M)
    am = C.c_long(-1)
    mx = C.c_double(0.0)
    _check(load_library().boss_acq_ei_moments(device, P, S, M, _dp(mu), _dp(var), _dp(coefs), _dp(ym),
                                              0 if best is None else 1, 0.0 if best is None else float(best),
                                              _ucp(mask), _dp(acq), C.byref(am), C.byref(mx)))
    return acq, am.value, mx.value


# ---------------------------------------------------------------- several GPUs from one process (boss_multi_*)
def init() -> int:
    """boss_init: open every visible device and (when RCCL loads) a communicator on each; returns the device count."""
    n = C.c_int(0)
    _check(load_library().boss_init(C.byref(n)))
    return n.value


def comm_info():
    """(devices opened by init(), exchanges over RCCL?)"""
    n, r = C.c_int(0), C.c_int(0)
    _check(load_library().boss_comm_info(C.byref(n), C.byref(r)))
    return n.value, bool(r.value)


def shutdown():
    load_library().boss_shutdown()


def multi_update(replicas: Sequence[GP], lengthscale, amplitude, noise_std, mean_X=None) -> float:
    """The same hyper-parameters on the replicas of one posterior (replicas[g] on device g), factorised concurrently."""
    G = len(replicas)
    arr = (C.c_void_p * G)(*[r._h for r in replicas])
    lam = _f64(np.asarray(lengthscale).reshape(-1), 1)
    if lam.shape[0] != replicas[0].d:
        raise BossError(BOSS_E_INVALID, "length(lengthscales) must equal x_dim")
    m = None if mean_X is None else _f64(np.asarray(mean_X).reshape(-1), 1)
    out = C.c_double(0.0)
    _check(load_library().boss_multi_gp_update(G, arr, _dp(lam), float(amplitude), float(noise_std), _dp(m), C.byref(out)))
    for r in replicas:
        r.logpdf = out.value
    return out.value


def _acq_common(P, S, M, fit_coefs, y_max, valid_mask, mean_Xs):
    coefs = _f64(np.asarray(fit_coefs).reshape(-1), 1)
    ym = None if y_max is None else _f64(np.asarray(y_max).reshape(-1), 1)
    mask = None if valid_mask is None else np.ascontiguousarray(np.asarray(valid_mask, dtype=bool).astype(np.uint8))
    ms = None
    if mean_Xs is not None:
        a = np.asarray(mean_Xs, dtype=np.float64).reshape(S, P, M)
        ms = np.ascontiguousarray(a.transpose(0, 2, 1))       # index p + P*(j + M*s)
    return coefs, ym, mask, ms


def multi_acq_ei(replicas: Sequence[Sequence[Sequence[GP]]], Xs, fit_coefs, y_max=None, best=None, valid_mask=None,
                 mean_Xs=None, want_acq: bool = True):
    """boss_multi_acq_ei: candidates sharded over the devices.  replicas[g][s][p] = replica on device g of output p,
    hyper-parameter sample s.  Xs d×M (host); mean_Xs None or [S][P][M].  Returns (acq[M] or None, argmax, max)."""
    G, S, P = len(replicas), len(replicas[0]), len(replicas[0][0])
    Xs = _f64(Xs, 2)
    M = Xs.shape[1]
    arr = (C.c_void_p * (G * S * P))()
    for g in range(G):
        for s in range(S):
            for p in range(P):
                arr[p + P * (s + S * g)] = replicas[g][s][p]._h
    coefs, ym, mask, ms = _acq_common(P, S, M, fit_coefs, y_max, valid_mask, mean_Xs)
    acq = np.zeros(M) if want_acq else None
    am, mx = C.c_long(-1), C.c_double(0.0)
    _check(load_library().boss_multi_acq_ei(G, P, S, arr, M, _dp(Xs), _dp(ms), _dp(coefs), _dp(ym), 0 if best is None else 1,
                                            0.0 if best is None else float(best), _ucp(mask), _dp(acq), C.byref(am), C.byref(mx)))
    return acq, am.value, mx.value


class MultiCandidates:
    """boss_multi_cand_create: candidates resident on G devices (shard g = the g-th balanced contiguous range of the columns)."""

    def __init__(self, Xs, G: int):
        Xs = _f64(Xs, 2)
        self.d, self.M, self.G = Xs.shape[0], Xs.shape[1], int(G)
        self._h = C.c_void_p()
        _check(load_library().boss_multi_cand_create(self.G, self.d, self.M, _dp(Xs), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load_library().boss_multi_cand_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multi_acq_ei_cand(replicas: Sequence[Sequence[Sequence[GP]]], cand: MultiCandidates, fit_coefs, y_max=None, best=None,
                      valid_mask=None, mean_Xs=None, want_acq: bool = True):
    """boss_multi_acq_ei_cand: multi_acq_ei on resident candidate shards.  Returns (acq[M] or None, argmax, max)."""
    G, S, P = len(replicas), len(replicas[0]), len(replicas[0][0])
    M = cand.M
    arr = (C.c_void_p * (G * S * P))()
    for g in range(G):
        for s in range(S):
            for p in range(P):
                arr[p + P * (s + S * g)] = replicas[g][s][p]._h
    coefs, ym, mask, ms = _acq_common(P, S, M, fit_coefs, y_max, valid_mask, mean_Xs)
    acq = np.zeros(M) if want_acq else None
    am, mx = C.c_long(-1), C.c_double(0.0)
    _check(load_library().boss_multi_acq_ei_cand(G, P, S, arr, cand._h, _dp(ms), _dp(coefs), _dp(ym), 0 if best is None else 1,
                                                 0.0 if best is None else float(best), _ucp(mask), _dp(acq), C.byref(am), C.byref(mx)))
    return acq, am.value, mx.value


def _multi_handles(fn_name, gps, Xs, fit_coefs, y_max, best, valid_mask, mean_Xs, want_acq):
    S, P = len(gps), len(gps[0])
    Xs = _f64(Xs, 2)
    M = Xs.shape[1]
    arr = (C.c_void_p * (P * S))()
    for s in range(S):
        for p in range(P):
            arr[p + P * s] = gps[s][p]._h
    coefs, ym, mask, ms = _acq_common(P, S, M, fit_coefs, y_max, valid_mask, mean_Xs)
    acq = np.zeros(M) if want_acq else None
    am, mx = C.c_long(-1), C.c_double(0.0)
    _check(getattr(load_library(), fn_name)(P, S, arr, M, _dp(Xs), _dp(ms), _dp(coefs), _dp(ym), 0 if best is None else 1,
                                            0.0 if best is None else float(best), _ucp(mask), _dp(acq), C.byref(am), C.byref(mx)))
    return acq, am.value, mx.value


def multi_acq_ei_outputs(gps: Sequence[Sequence[GP]], Xs, fit_coefs, y_max=None, best=None, valid_mask=None, mean_Xs=None,
                         want_acq: bool = True):
    """boss_multi_acq_ei_outputs: gps[s][p] may live on any device (outputs sharded one per GPU)."""
    return _multi_handles("boss_multi_acq_ei_outputs", gps, Xs, fit_coefs, y_max, best, valid_mask, mean_Xs, want_acq)


def multi_acq_ei_samples(gps: Sequence[Sequence[GP]], Xs, fit_coefs, y_max=None, best=None, valid_mask=None, mean_Xs=None,
                         want_acq: bool = True):
    """boss_multi_acq_ei_samples: all outputs of sample s on one device, different samples on different devices."""
    return _multi_handles("boss_multi_acq_ei_samples", gps, Xs, fit_coefs, y_max, best, valid_mask, mean_Xs, want_acq)


def multi_loglike_batch(G: int, X, y, kernel, lengthscales, amplitudes, noise_stds, mean_X=None, discrete=None):
    """boss_multi_loglike_batch: the S hyper-parameter sets split over devices 0..G-1.  Returns (ll[S], status[S])."""
    X = _f64(X, 2)
    y = _f64(np.asarray(y).reshape(-1), 1)
    lam = _f64(lengthscales, 2)
    d, N = X.shape
    S = lam.shape[1]
    if lam.shape[0] != d:
        raise BossError(BOSS_E_INVALID, "lengthscales must be d×S")
    amp = _f64(np.asarray(amplitudes).reshape(-1), 1)
    sig = _f64(np.asarray(noise_stds).reshape(-1), 1)
    stride, m = 0, None
    if mean_X is not None:
        m = np.asarray(mean_X, dtype=np.float64)
        if m.ndim == 2:
            m, stride = np.ascontiguousarray(m), N
        else:
            m = np.ascontiguousarray(m.reshape(-1))
    disc = None if discrete is None else np.ascontiguousarray(np.asarray(discrete, dtype=bool).astype(np.uint8))
    ll = np.zeros(S)
    st = np.zeros(S, dtype=np.int32)
    _check(load_library().boss_multi_loglike_batch(G, _kernel_id(kernel), d, N, _dp(X), _dp(y), _dp(m), stride, _ucp(disc), S,
                                                   _dp(lam), _dp(amp), _dp(sig), _dp(ll), st.ctypes.data_as(C.POINTER(C.c_int))))
    return ll, st


def bench_mfma_f64(device: int = 0, iters: int = 20000) -> float:
    out = C.c_double(0.0)
    _check(load_library().boss_bench_mfma_f64(device, iters, C.byref(out)))
    return out.value


def prof_enable(device: int, on: bool):
    _check(load_library().boss_prof_enable(device, 1 if on else 0))


def prof_reset(device: int):
    _check(load_library().boss_prof_reset(device))


def prof_get(device: int, kernel_class: str):
    ms = C.c_double(0.0)
    n = C.c_long(0)
    _check(load_library().boss_prof_get(device, kernel_class.encode(), C.byref(ms), C.byref(n)))
    return ms.value, n.value

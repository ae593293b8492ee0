"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/liboracle.so (the plain-C
restatement in nmrfit_oracle.c).  Builds it with `make -C oracle` if missing.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_objective_batch.argtypes = [ctypes.c_int64, _dp, _dp, _dp, _dp, ctypes.c_int64,
                                             ctypes.c_int, _dp, _dp, ctypes.c_int]
        L.oracle_objective_batch.restype = ctypes.c_int
        L.oracle_residual_batch.argtypes = [ctypes.c_int64, _dp, _dp, _dp, _dp, ctypes.c_int64,
                                            ctypes.c_int, _dp, _dp, _dp, ctypes.c_int]
        L.oracle_residual_batch.restype = ctypes.c_int
        L.oracle_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def objective_batch(X, w, u, v, weights, threads=1):
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[None, :]
    S, D = X.shape
    P = (D - 4) // 3
    assert D == 4 + 3 * P
    w, pw = _c(w); u, pu = _c(u); v, pv = _c(v); weights, pwt = _c(weights)
    f = np.empty(S)
    lib().oracle_objective_batch(w.size, pw, pu, pv, pwt, S, P, X.ctypes.data_as(_dp),
                                 f.ctypes.data_as(_dp), threads)
    return f


def residual_batch(X, w, u, v, weights, threads=1):
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[None, :]
    B, D = X.shape
    P = (D - 4) // 3
    w, pw = _c(w); u, pu = _c(u); v, pv = _c(v); weights, pwt = _c(weights)
    R = np.empty((B, w.size))
    f = np.empty(B)
    lib().oracle_residual_batch(w.size, pw, pu, pv, pwt, B, P, X.ctypes.data_as(_dp),
                                R.ctypes.data_as(_dp), f.ctypes.data_as(_dp), threads)
    return R, f


def max_threads():
    return lib().oracle_max_threads()

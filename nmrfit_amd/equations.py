"""
Host-side mirror of the reference's operator interface for the hot path
(nmrfit/equations.py): same names and argument meaning, evaluated on the MI355X through
libnmrfit_amd.so.

    objective(x, w, u, v, weights, fit_im=False) -> float      nmrfit/equations.py:152
    Evaluator(w, u, v, weights).objective_batch(X) -> f[S]     the batched form of the above
    Evaluator(...).residual_batch(X) -> R[B, N]                weights*(V_data - V_fit), eq.py:202
    laplace1d(x, n=10, omega=0.33333333)                       nmrfit/equations.py:215-238

    Evaluator(...).contributions(x, w) -> real[P, N], imag[P, N]   voigt + Kramers-Kronig per peak
    ps2(u, v, p0, p1, inv) / voigt(w, ...) / kk_relation_vectorized(w, ...)   host utilities

``fit_im=True`` is supported: the imaginary line shape the reference obtains by adaptive
quadrature per grid point (equations.py:9-80) is evaluated in closed form on the GPU.

There is no CPU fallback in this module: without the HIP library and a gfx950 device every
evaluation raises ``NmrfitError``.
"""
import collections
import ctypes
import hashlib
import weakref

import numpy as np

from . import _cabi
from ._cabi import NmrfitError  # noqa: F401  (re-export)


def fit_im_mode(fit_im):
    """Map the reference's ``fit_im`` argument to the library's mode: False/0 -> real part only;
    True/1 -> exactly what the reference computes (equations.py:197-209: the imaginary model is
    the LAST peak only, since I_fit is assigned, not accumulated); "sum"/2 -> all peaks."""
    if fit_im is False or fit_im is None or fit_im == 0:
        return _cabi.FIT_IM_OFF
    if fit_im is True or fit_im == 1 or fit_im == "reference":
        return _cabi.FIT_IM_REFERENCE
    if fit_im == 2 or fit_im == "sum":
        return _cabi.FIT_IM_SUM
    raise ValueError("fit_im must be False, True, 'reference' or 'sum'")


class Evaluator:
    """GPU-resident (w, u, v, weights) + the batched objective.  One per device/process.

    Replaces the ``args=(w, u, v, weights, fit_im)`` tuple FitUtility.fit passes to the
    optimiser for every call (nmrfit/utils.py:176)."""

    def __init__(self, w, u, v, weights, device=0):
        self._lib = _cabi.lib()
        self._ctx = ctypes.c_void_p()
        w, u, v, weights = (_cabi.f64(a) for a in (w, u, v, weights))
        if not (w.ndim == u.ndim == v.ndim == weights.ndim == 1 and w.size == u.size == v.size == weights.size):
            raise ValueError("w, u, v, weights must be 1-D arrays of equal length")
        self.N = int(w.size)
        self.device = device
        self._children = weakref.WeakSet()     # swarms built on this context: closed before it
        _cabi.check(self._lib.nmrfit_ctx_create(device, self.N, _cabi.ptr(w), _cabi.ptr(u), _cabi.ptr(v),
                                                _cabi.ptr(weights), ctypes.byref(self._ctx)))

    # -- life cycle ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            # swarms and communicators hold a pointer to this context; swarms go first (a swarm may
            # still have a communicator attached, and nmrfit_comm_destroy refuses while one does)
            for child in sorted(getattr(self, "_children", ()), key=lambda c: 0 if hasattr(c, "set_comm") else 1):
                child.close()
            self._lib.nmrfit_ctx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._ctx

    # -- configuration ------------------------------------------------------------------
    def set_weights(self, weights):
        weights = _cabi.f64(weights)
        if weights.shape != (self.N,):
            raise ValueError("weights must have length N")
        _cabi.check(self._lib.nmrfit_ctx_set_weights(self._ctx, _cabi.ptr(weights)))

    def set_variant(self, variant):
        _cabi.check(self._lib.nmrfit_ctx_set_variant(self._ctx, int(variant)))

    def set_stream(self, stream_handle):
        """Run on an externally owned HIP stream (e.g. torch.cuda.current_stream().cuda_stream)
        so that launches order with a collective library's work without host syncs."""
        _cabi.check(self._lib.nmrfit_ctx_set_stream(self._ctx, ctypes.c_void_p(stream_handle) if stream_handle
                                                    else None))

    def synchronize(self):
        _cabi.check(self._lib.nmrfit_ctx_synchronize(self._ctx))

    # -- the hot path, host pointers ------------------------------------------------------
    @staticmethod
    def _as_batch(X):
        X = _cabi.f64(X)
        if X.ndim == 1:
            X = X[None, :]
        if X.ndim != 2 or X.shape[1] < 4 or (X.shape[1] - 4) % 3:
            raise ValueError("parameter matrix must be S x (4 + 3P)")
        return X, (X.shape[1] - 4) // 3

    def objective_batch(self, X, fit_im=False):
        X, P = self._as_batch(X)
        f = np.empty(X.shape[0], dtype=np.float64)
        _cabi.check(self._lib.nmrfit_objective_batch(self._ctx, X.shape[0], P, _cabi.ptr(X), fit_im_mode(fit_im),
                                                     _cabi.ptr(f)))
        return f

    def set_fit_im(self, fit_im):
        """Imaginary-part mode for the device-resident calls and the swarm (see fit_im_mode)."""
        _cabi.check(self._lib.nmrfit_ctx_set_fit_im(self._ctx, fit_im_mode(fit_im)))

    def contributions(self, x, w=None):
        """Per-peak (real, imag) contributions [P, Nout] of one parameter vector: voigt per peak
        and its Kramers-Kronig partner in closed form (FitUtility.generate_result,
        nmrfit/utils.py:262-281).  ``w`` None -> the context's grid."""
        x = _cabi.f64(x)
        if x.ndim != 1 or x.size < 4 or (x.size - 4) % 3:
            raise ValueError("parameter vector must have 4 + 3P entries")
        P = (x.size - 4) // 3
        wout = None if w is None else _cabi.f64(w)
        n = self.N if wout is None else wout.size
        real = np.empty((P, n), dtype=np.float64)
        imag = np.empty((P, n), dtype=np.float64)
        _cabi.check(self._lib.nmrfit_contributions(self._ctx, P, _cabi.ptr(x), n, _cabi.ptr(wout),
                                                   _cabi.ptr(real), _cabi.ptr(imag)))
        return real, imag

    def generate_result(self, x, w=None):
        """FitUtility.generate_result's arithmetic for one parameter vector in one launch (nmrfit/utils.py:226-295):
        ``(real[P, Nout], imag[P, Nout], fit[4, Nout], data[2, N])`` -- the per-peak contributions, (V_fit, I_fit, u_fit,
        v_fit) and the spectrum rotated by the fitted phase (V, I).  ``w`` None -> the context's grid."""
        x = _cabi.f64(x)
        if x.ndim != 1 or x.size < 4 or (x.size - 4) % 3:
            raise ValueError("parameter vector must have 4 + 3P entries")
        P = (x.size - 4) // 3
        wout = None if w is None else _cabi.f64(w)
        n = self.N if wout is None else wout.size
        real = np.empty((P, n), dtype=np.float64)
        imag = np.empty((P, n), dtype=np.float64)
        fit = np.empty((4, n), dtype=np.float64)
        data = np.empty((2, self.N), dtype=np.float64)
        _cabi.check(self._lib.nmrfit_generate_result(self._ctx, P, _cabi.ptr(x), n, _cabi.ptr(wout), _cabi.ptr(real),
                                                     _cabi.ptr(imag), _cabi.ptr(fit), _cabi.ptr(data)))
        return real, imag, fit, data

    def residual_batch(self, X, return_f=False):
        X, P = self._as_batch(X)
        R = np.empty((X.shape[0], self.N), dtype=np.float64)
        f = np.empty(X.shape[0], dtype=np.float64)
        _cabi.check(self._lib.nmrfit_residual_batch(self._ctx, X.shape[0], P, _cabi.ptr(X), _cabi.ptr(R),
                                                    _cabi.ptr(f)))
        return (R, f) if return_f else R

    # -- device-resident helpers (bench / swarm) ------------------------------------------
    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        _cabi.check(self._lib.nmrfit_dev_alloc(self._ctx, int(nbytes), ctypes.byref(p)))
        return p

    def dev_free(self, dptr):
        _cabi.check(self._lib.nmrfit_dev_free(self._ctx, dptr))

    def upload(self, dptr, host):
        host = np.ascontiguousarray(host)
        _cabi.check(self._lib.nmrfit_memcpy_h2d(self._ctx, dptr, _cabi.ptr(host), host.nbytes))

    def download(self, dptr, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        _cabi.check(self._lib.nmrfit_memcpy_d2h(self._ctx, _cabi.ptr(out), dptr, out.nbytes))
        return out

    def objective_batch_dev(self, S, P, dX, df):
        _cabi.check(self._lib.nmrfit_objective_batch_dev(self._ctx, S, P, dX, df))

    def residual_batch_dev(self, B, P, dX, dR, df=None):
        _cabi.check(self._lib.nmrfit_residual_batch_dev(self._ctx, B, P, dX, dR, df))

    def timer_begin(self):
        _cabi.check(self._lib.nmrfit_timer_begin(self._ctx))

    def timer_end(self):
        ms = ctypes.c_double(0.0)
        _cabi.check(self._lib.nmrfit_timer_end(self._ctx, ctypes.byref(ms)))
        return ms.value

    def prof_enable(self, capacity):
        """Bracket every objective kernel launch with HIP events (nmrfit_prof_*); 0 disables."""
        _cabi.check(self._lib.nmrfit_prof_enable(self._ctx, int(capacity)))
        self._prof_cap = int(capacity)

    def prof_mark(self):
        _cabi.check(self._lib.nmrfit_prof_mark(self._ctx))

    def prof_read(self):
        """(kernel_ms[], step_ms[], clock_mhz): per-launch durations of the objective kernel, the
        durations between consecutive marks, and the shader clock seen by the last profiled
        launch (0 if unknown).  Synchronizes; rewinds the recording."""
        cap = getattr(self, "_prof_cap", 0)
        k = np.zeros(max(cap, 1))
        st = np.zeros(max(cap, 1))
        nk, ns, mhz = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_double(0.0)
        _cabi.check(self._lib.nmrfit_prof_read(self._ctx, _cabi.ptr(k), cap, ctypes.byref(nk), _cabi.ptr(st), cap,
                                               ctypes.byref(ns), ctypes.byref(mhz)))
        return k[:nk.value].copy(), st[:ns.value].copy(), mhz.value

    def last_launch(self):
        waves, nseg, seg_len = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int64(0)
        _cabi.check(self._lib.nmrfit_last_launch(self._ctx, ctypes.byref(waves), ctypes.byref(nseg),
                                                 ctypes.byref(seg_len)))
        wpw = ctypes.c_int32(0)
        _cabi.check(self._lib.nmrfit_last_launch_workgroup(self._ctx, ctypes.byref(wpw)))
        return dict(waves=waves.value, segments=nseg.value, segment_len=seg_len.value, waves_per_workgroup=wpw.value)


# ---- scalar shim with the reference signature ---------------------------------------------
_shim_cache = collections.OrderedDict()     # least recently used first
_SHIM_CACHE_MAX = 4


def _key(*arrays):
    """Content key of the constant arrays: a 128-bit digest each (a collision would silently
    evaluate against another spectrum's data, so no 32-bit checksums here)."""
    return tuple((a.size, hashlib.blake2b(a.view(np.uint8), digest_size=16).digest()) for a in arrays)


def _cached_evaluator(k, make):
    ev = _shim_cache.get(k)
    if ev is None:
        while len(_shim_cache) >= _SHIM_CACHE_MAX:
            _, old = _shim_cache.popitem(last=False)      # evict the least recently used
            old.close()
        ev = make()
        _shim_cache[k] = ev
    else:
        _shim_cache.move_to_end(k)
    return ev


def objective(x, w, u, v, weights, fit_im=False):
    """Drop-in for nmrfit.equations.objective (equations.py:152): one particle, one float.
    The constant arrays are cached on the GPU between calls (keyed by content), so a
    third-party optimiser that calls this per particle still avoids re-uploading them; use
    ``Evaluator.objective_batch`` to evaluate a whole swarm per launch."""
    arrays = tuple(_cabi.f64(a) for a in (w, u, v, weights))
    ev = _cached_evaluator(_key(*arrays), lambda: Evaluator(*arrays))
    return float(ev.objective_batch(np.asarray(x, dtype=np.float64), fit_im=fit_im)[0])


def laplace1d(x, n=10, omega=0.33333333):
    """In-place 1-D Laplacian smoothing, end points fixed (equations.py:215-238).  Host code
    in the reference too: it runs once per fit to build ``weights``."""
    for _ in range(n):
        x[1:-1] = (1. - omega) * x[1:-1] + omega * 0.5 * (x[2:] + x[:-2])
    return x


# ---- single-line-shape utilities with the reference's names --------------------------------
def _grid_evaluator(w):
    """An Evaluator that only carries a grid (for voigt / kk_relation_vectorized)."""
    w = _cabi.f64(w)
    z = np.zeros_like(w)
    return _cached_evaluator(("grid",) + _key(w), lambda: Evaluator(w, z, z, np.ones_like(w)))


def voigt(w, r, yoff, width, loc, a):
    """nmrfit.equations.voigt (equations.py:115-149) on the GPU: yoff + a*(r*L + (1-r)*G)."""
    x = np.array([0.0, 0.0, r, yoff, width, loc, a], dtype=np.float64)
    return _grid_evaluator(w).contributions(x)[0][0]


def kk_relation_vectorized(w, r, yoff, width, loc, a):
    """nmrfit.equations.kk_relation_vectorized (equations.py:52-80, 242): the Kramers-Kronig
    partner of ``voigt`` over w.  The reference integrates numerically for every point
    (scipy.integrate.quad, ~4 ms per point); this is the closed form (Lorentzian dispersion +
    Dawson's integral), which the quadrature approximates to ~1e-12."""
    x = np.array([0.0, 0.0, r, yoff, width, loc, a], dtype=np.float64)
    return _grid_evaluator(w).contributions(x)[1][0]


def kk_relation(w, r, yoff, width, loc, a):
    """nmrfit.equations.kk_relation (equations.py:52-80): the transform at ONE frequency w."""
    return float(kk_relation_vectorized(np.array([w], dtype=np.float64), r, yoff, width, loc, a)[0])


def kk_relation_parallel(w, r, yoff, width, loc, a, pool=None):
    """nmrfit.equations.kk_relation_parallel (equations.py:83-112).  ``pool`` is accepted for
    signature compatibility and ignored: the whole array is one GPU launch."""
    return kk_relation_vectorized(w, r, yoff, width, loc, a)

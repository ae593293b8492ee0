"""
Device-batched fits: K independent spectra fitted together, one kernel launch per swarm generation for all of them
(``nmrfit_batch_*`` of libnmrfit_amd.so, csrc/batch.hip).

The reference's users call ``nmrfit.fit`` once per spectrum (nmrfit/core.py:64, README.md:64-66), each fit a
204-particle swarm (nmrfit/utils.py:177) -- a fraction of an MI355X.  ``FitBatch`` holds K spectra of equal length and
K swarms of equal size on the device; every fit has its own peak count, box, seed and stopping rule, follows exactly
the trajectory a lone ``nmrfit_amd.fit`` gives it (bit-identical ``params`` / ``error`` for the same seed) and stops on
its own.  ``nmrfit_amd.fit_many`` builds these batches from a list of jobs.
"""
import ctypes

import numpy as np

from . import _cabi, equations, pso


class FitBatch:
    """K fits on one GPU.

    spectra : K tuples ``(w, u, v, weights)`` of equal length N (what FitUtility.fit passes as ``args``,
              nmrfit/utils.py:176)
    lowers, uppers : K parameter boxes, 4 + 3 P_k floats each (nmrfit/containers.py:193-217)
    swarmsize : particles per fit (the same for every fit of a batch)
    seeds : K integers (the swarm's random stream, like options['seed'] of ``fit``)
    omega, phip, phig, minstep, minfunc : scalars or length-K sequences
    variant : "default" or "farfield" (what ``fit`` would select for these shapes)
    fit_im : False, True (the reference's imaginary term) or "sum" (every peak; "default" kernel), for the whole batch
    """

    def __init__(self, spectra, lowers, uppers, swarmsize=pso.DEFAULTS["swarmsize"], seeds=None,
                 omega=pso.DEFAULTS["omega"], phip=pso.DEFAULTS["phip"], phig=pso.DEFAULTS["phig"],
                 minstep=pso.DEFAULTS["minstep"], minfunc=pso.DEFAULTS["minfunc"], variant="default", fit_im=False,
                 device=0):
        self._lib = _cabi.lib()
        self._h = ctypes.c_void_p()
        K = len(spectra)
        if K == 0 or len(lowers) != K or len(uppers) != K:
            raise ValueError("FitBatch: as many boxes as spectra, at least one")
        N = len(spectra[0][0])
        planes = [np.empty((K, N)) for _ in range(4)]
        self._w_range = [(np.min(sp[0]), np.max(sp[0])) for sp in spectra]      # (generate: the upsampled grids' ends)
        for k, sp in enumerate(spectra):
            if any(len(a) != N for a in sp):
                raise ValueError("FitBatch: every spectrum of a batch has the same length (fit %d differs)" % k)
            for a, plane in zip(sp, planes):
                plane[k] = a            # (contiguous float64 rows: the reference hands out reversed views, core.py:60)
        lbs = [_cabi.f64(lo) for lo in lowers]
        ubs = [_cabi.f64(up) for up in uppers]
        self.D = []
        for k, (lo, up) in enumerate(zip(lbs, ubs)):
            assert len(lo) == len(up), 'Lower- and upper-bounds must be the same length'
            assert np.all(up > lo), 'All upper-bound values must be greater than lower-bound values'
            if lo.size < 4 or (lo.size - 4) % 3:
                raise ValueError("bounds must have 4 + 3P entries (fit %d)" % k)
            self.D.append(int(lo.size))
        self.K, self.N, self.S = K, N, int(swarmsize)
        self.P = np.array([(d - 4) // 3 for d in self.D], dtype=np.int32)
        self.offsets = np.concatenate(([0], np.cumsum(self.D)))
        lower = np.concatenate(lbs)
        upper = np.concatenate(ubs)
        if seeds is None:
            seeds = np.random.SeedSequence().generate_state(K, dtype=np.uint64)

        def per_fit(x):
            a = np.broadcast_to(np.asarray(x, dtype=np.float64), (K,))
            return a
        om, pp, pg, ms, mf = (per_fit(x) for x in (omega, phip, phig, minstep, minfunc))
        self.minstep, self.minfunc = ms.copy(), mf.copy()
        self.seeds = [int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds]
        prm = (_cabi.PsoParams * K)()
        for k in range(K):
            prm[k] = _cabi.PsoParams(om[k], pp[k], pg[k], ms[k], mf[k], self.seeds[k])
        _cabi.check(self._lib.nmrfit_batch_create(int(device), K, N, _cabi.ptr(planes[0]), _cabi.ptr(planes[1]),
                                                  _cabi.ptr(planes[2]), _cabi.ptr(planes[3]), _cabi.ptr(self.P),
                                                  _cabi.ptr(lower), _cabi.ptr(upper), self.S, prm,
                                                  _cabi.variant_id(variant), equations.fit_im_mode(fit_im),
                                                  ctypes.byref(self._h)))

    # -- life cycle ----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nmrfit_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the product calls -------------------------------------------------------------------------
    def run(self, maxiter=pso.DEFAULTS["maxiter"], check_every=64):
        """Generation 0 (if needed) + up to ``maxiter`` generations of every fit; returns when all have stopped."""
        _cabi.check(self._lib.nmrfit_batch_run(self._h, int(maxiter), int(check_every)))

    def status(self):
        """Per fit: completed generations, stop code (0 running, 1 minfunc, 2 minstep), current fg."""
        it = np.zeros(self.K, dtype=np.int64)
        stop = np.zeros(self.K, dtype=np.int32)
        fg = np.zeros(self.K)
        _cabi.check(self._lib.nmrfit_batch_status(self._h, _cabi.ptr(it), _cabi.ptr(stop), _cabi.ptr(fg)))
        return [dict(iteration=int(i), stop=int(s), fg=float(f)) for i, s, f in zip(it, stop, fg)]

    def best(self):
        """Per fit ``(x_best, f_best)``: after a stop these are pyswarm's return values."""
        x = np.empty(int(self.offsets[-1]))
        f = np.empty(self.K)
        _cabi.check(self._lib.nmrfit_batch_best(self._h, _cabi.ptr(x), _cabi.ptr(f)))
        return [(x[self.offsets[k]:self.offsets[k + 1]].copy(), float(f[k])) for k in range(self.K)]

    def generate(self, scale=1, grids=None):
        """FitUtility.generate_result for every fit of the batch at its best position (nmrfit/utils.py:226-295), ONE launch
        from the batch's resident spectra.  ``scale`` == 1: on each fit's own grid; else on
        ``np.linspace(w.min(), w.max(), int(scale * N))`` per fit (utils.py:236; ``grids``: the K output grids, K x Nout,
        if the caller has them).  Returns per fit a dict with ``w`` (None when scale == 1: the fit's own grid), ``real``,
        ``imag`` ([P_k, Nout] each), ``V, I, u, v`` (the fit: utils.py:276-284) and ``data_V, data_I`` (the spectrum rotated
        by the fitted phase, what data.shift_phase(method='manual') stores, utils.py:251) -- all views of four arrays."""
        K, N = self.K, self.N
        if scale == 1.0 and grids is None:
            wout, n = None, N
        else:
            if grids is None:
                n = int(scale * N)
                grids = np.stack([np.linspace(lo, hi, n) for lo, hi in self._w_range])
            wout = _cabi.f64(grids)
            if wout.ndim != 2 or wout.shape[0] != K:
                raise ValueError("FitBatch.generate: grids must be K x Nout")
            n = wout.shape[1]
        rows = int(self.P.sum())
        real = np.empty((rows, n))
        imag = np.empty((rows, n))
        fit = np.empty((K, 4, n))
        data = np.empty((K, 2, N))
        _cabi.check(self._lib.nmrfit_batch_contributions(self._h, n, _cabi.ptr(wout), _cabi.ptr(real), _cabi.ptr(imag),
                                                         _cabi.ptr(fit), _cabi.ptr(data)))
        out, row = [], 0
        for k in range(K):
            p = int(self.P[k])
            out.append(dict(w=None if wout is None else wout[k], real=real[row:row + p], imag=imag[row:row + p],
                            V=fit[k, 0], I=fit[k, 1], u=fit[k, 2], v=fit[k, 3], data_V=data[k, 0], data_I=data[k, 1]))
            row += p
        return out

    # -- diagnostics (include/nmrfit_amd_diag.h) ----------------------------------------------------
    def step(self):
        """Generation 0 on the first call, then one generation per call (asynchronous)."""
        _cabi.check(self._lib.nmrfit_batch_step(self._h))

    def synchronize(self):
        _cabi.check(self._lib.nmrfit_batch_synchronize(self._h))

    def set_geometry(self, mode):
        """"workgroup" (a workgroup per particle) or "wave" (a wave per particle); bit-identical results."""
        if isinstance(mode, str):
            mode = {"workgroup": 0, "wave": 1}[mode.lower()]
        _cabi.check(self._lib.nmrfit_batch_set_geometry(self._h, int(mode)))

    def geometry(self):
        m, w, s = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        n = ctypes.c_int64(0)
        _cabi.check(self._lib.nmrfit_batch_geometry(self._h, ctypes.byref(m), ctypes.byref(w), ctypes.byref(s),
                                                    ctypes.byref(n)))
        return dict(mode=("workgroup", "wave")[m.value], waves_per_workgroup=w.value, segments=s.value,
                    workgroups=n.value)

    def state(self, k):
        D = self.D[k]
        x = np.empty((self.S, D)); v = np.empty_like(x); p = np.empty_like(x)
        fx = np.empty(self.S); fp = np.empty(self.S)
        _cabi.check(self._lib.nmrfit_batch_get_state(self._h, int(k), _cabi.ptr(x), _cabi.ptr(v), _cabi.ptr(p),
                                                     _cabi.ptr(fx), _cabi.ptr(fp)))
        return dict(x=x, v=v, p=p, fx=fx, fp=fp)

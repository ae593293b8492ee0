"""
Device-batched fits: K independent spectra fitted together, one kernel launch per swarm generation for all of them
(``nmrfit_batch_*`` of libnmrfit_amd.so, csrc/batch.hip).

The reference's users call ``nmrfit.fit`` once per spectrum (nmrfit/core.py:64, README.md:64-66), each fit a
204-particle swarm (nmrfit/utils.py:177) -- a fraction of an MI355X.  ``FitBatch`` holds K spectra -- of any lengths:
every dataset is cropped to its own region, nmrfit/containers.py:112-130 -- and K swarms (of any sizes) on the device;
every fit has its own grid, peak count, box, seed and stopping rule, follows exactly the trajectory a lone
``nmrfit_amd.fit`` gives it (bit-identical ``params`` / ``error`` for the same seed) and stops on its own.
``nmrfit_amd.fit_many`` builds these batches from a list of jobs.
"""
import ctypes

import numpy as np

from . import _cabi, equations, pso


class FitBatch:
    """K fits on one GPU.

    spectra : K tuples ``(w, u, v, weights)`` (what FitUtility.fit passes as ``args``, nmrfit/utils.py:176); the K
              spectra may differ in length
    lowers, uppers : K parameter boxes, 4 + 3 P_k floats each (nmrfit/containers.py:193-217)
    swarmsize : particles per fit: one number, or K of them (``options['swarmsize']`` of each fit, nmrfit/utils.py:177)
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
        Ns = np.array([len(sp[0]) for sp in spectra], dtype=np.int64)
        noff = np.concatenate(([0], np.cumsum(Ns)))
        planes = [np.empty(int(noff[-1])) for _ in range(4)]
        for k, sp in enumerate(spectra):
            if len(sp) != 4 or any(len(a) != Ns[k] for a in sp) or Ns[k] == 0:
                raise ValueError("FitBatch: w, u, v, weights of a spectrum have the same non-zero length (fit %d)" % k)
            for a, plane in zip(sp, planes):
                plane[noff[k]:noff[k + 1]] = a      # (contiguous float64: the reference hands out reversed views, core.py:60)
        self._w_range = [(np.min(sp[0]), np.max(sp[0])) for sp in spectra]      # (generate: the upsampled grids' ends)
        lbs = [_cabi.f64(lo) for lo in lowers]
        ubs = [_cabi.f64(up) for up in uppers]
        self.D = []
        for k, (lo, up) in enumerate(zip(lbs, ubs)):
            assert len(lo) == len(up), 'Lower- and upper-bounds must be the same length'
            assert np.all(up > lo), 'All upper-bound values must be greater than lower-bound values'
            if lo.size < 4 or (lo.size - 4) % 3:
                raise ValueError("bounds must have 4 + 3P entries (fit %d)" % k)
            self.D.append(int(lo.size))
        # N: the common grid length, or None when the spectra differ in length (Ns, noff: per fit)
        # S: the common swarm size, or None when the swarms differ in size (Ss: per fit)
        self.Ss = np.ascontiguousarray(np.broadcast_to(np.asarray(swarmsize, dtype=np.int64), (K,)))
        self.K, self.Ns, self.noff = K, Ns, noff
        self.N = int(Ns[0]) if np.all(Ns == Ns[0]) else None
        self.S = int(self.Ss[0]) if np.all(self.Ss == self.Ss[0]) else None
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
        _cabi.check(self._lib.nmrfit_batch_create_ragged(int(device), K, _cabi.ptr(Ns), _cabi.ptr(planes[0]),
                                                         _cabi.ptr(planes[1]), _cabi.ptr(planes[2]), _cabi.ptr(planes[3]),
                                                         _cabi.ptr(self.P), _cabi.ptr(lower), _cabi.ptr(upper), _cabi.ptr(self.Ss), prm,
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
        ``np.linspace(w.min(), w.max(), int(scale * N_k))`` per fit (utils.py:236; ``grids``: the K output grids, if the
        caller has them).  Returns per fit a dict with ``w`` (None when scale == 1: the fit's own grid), ``real``,
        ``imag`` ([P_k, Nout] each), ``V, I, u, v`` (the fit: utils.py:276-284) and ``data_V, data_I`` (the spectrum rotated
        by the fitted phase, what data.shift_phase(method='manual') stores, utils.py:251) -- all views of four arrays."""
        K, Ns = self.K, self.Ns
        if scale == 1.0 and grids is None:
            wout = nout = None
            n = Ns
        else:
            if grids is None:
                grids = [np.linspace(lo, hi, int(scale * Ns[k])) for k, (lo, hi) in enumerate(self._w_range)]
            if len(grids) != K:
                raise ValueError("FitBatch.generate: one output grid per fit")
            n = np.array([len(g) for g in grids], dtype=np.int64)
            nout = n
            wout = np.concatenate([_cabi.f64(g) for g in grids]) if K else np.empty(0)
        P = self.P.astype(np.int64)
        roff = np.concatenate(([0], np.cumsum(P * n)))          # contributions: P_k rows of n_k per fit
        foff = np.concatenate(([0], np.cumsum(4 * n)))
        woff = np.concatenate(([0], np.cumsum(n)))
        real = np.empty(int(roff[-1]))
        imag = np.empty(int(roff[-1]))
        fit = np.empty(int(foff[-1]))
        data = np.empty(2 * int(self.noff[-1]))
        _cabi.check(self._lib.nmrfit_batch_contributions(self._h, _cabi.ptr(nout), _cabi.ptr(wout), _cabi.ptr(real),
                                                         _cabi.ptr(imag), _cabi.ptr(fit), _cabi.ptr(data)))
        out = []
        for k in range(K):
            nk, Nk = int(n[k]), int(Ns[k])
            f4 = fit[foff[k]:foff[k + 1]].reshape(4, nk)
            d2 = data[2 * self.noff[k]:2 * self.noff[k + 1]].reshape(2, Nk)
            out.append(dict(w=None if wout is None else wout[woff[k]:woff[k + 1]],
                            real=real[roff[k]:roff[k + 1]].reshape(int(P[k]), nk),
                            imag=imag[roff[k]:roff[k + 1]].reshape(int(P[k]), nk),
                            V=f4[0], I=f4[1], u=f4[2], v=f4[3], data_V=d2[0], data_I=d2[1]))
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
        D, S = self.D[k], int(self.Ss[k])
        x = np.empty((S, D)); v = np.empty_like(x); p = np.empty_like(x)
        fx = np.empty(S); fp = np.empty(S)
        _cabi.check(self._lib.nmrfit_batch_get_state(self._h, int(k), _cabi.ptr(x), _cabi.ptr(v), _cabi.ptr(p),
                                                     _cabi.ptr(fx), _cabi.ptr(fp)))
        return dict(x=x, v=v, p=p, fx=fx, fp=fp)

#!/usr/bin/env python3
"""Kernel time of the objective with the imaginary channel at BASELINE's C3 shape
(4096 x 65536 x 24): fit_im False / True (reference: last peak only) / "sum" (all peaks), DEFAULT
and FARFIELD, HIP events around the kernel alone after 0.3 s of the same launches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import synth, _cabi
from nmrfit_amd.equations import Evaluator

if len(sys.argv) > 1:      # another build of the library (A/B): tools/fit_im_timing.py nmrfit_amd/lib/libab_x.so
    import ctypes
    L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
    for name, argtypes in _cabi.ALL_SIGNATURES.items():
        if not hasattr(L, name): continue      # (an older build of the library: entry points added since)
        fn = getattr(L, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
    L.nmrfit_last_error.argtypes = []; L.nmrfit_last_error.restype = ctypes.c_char_p
    _cabi._LIB = L
    print("library:", sys.argv[1])

sp, X = synth.make_workload("C3")
S, D = X.shape
P = (D - 4) // 3
with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
    dX, df = ev.dev_alloc(X.nbytes), ev.dev_alloc(8 * S)
    ev.upload(dX, X)
    for vname in ("default", "farfield"):
        ev.set_variant(_cabi.variant_id(vname))
        for mode in (False, True, "sum"):
            ev.set_fit_im(mode)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                ev.objective_batch_dev(S, P, dX, df)
                ev.synchronize()
            ev.prof_enable(10)
            for _ in range(10):
                ev.objective_batch_dev(S, P, dX, df)
            k = ev.prof_read()[0]
            ev.prof_enable(0)
            f = ev.download(df, (S,))
            print("%-8s fit_im=%-5s: %.3f ms per launch (min %.3f)   f[:3] = %s" % (
                vname, mode, np.mean(k), np.min(k), " ".join(float(v).hex() for v in f[:3])))

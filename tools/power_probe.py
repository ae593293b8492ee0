#!/usr/bin/env python3
"""Kernel time and shader clock of the C3 objective launch over ~12 s of back-to-back launches, one line a second --
to be read next to a `rocm-smi --showpower --showclocks` loop started beforehand by the calling shell (the chip's fp64
clock under this load is power-managed and differs from box to box).
    python tools/power_probe.py [variant] [seconds] [library: another build, e.g. a -DNMRFIT_DIAG_NOLOAD=2 one]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import synth, _cabi
from nmrfit_amd.equations import Evaluator

variant = sys.argv[1] if len(sys.argv) > 1 else "default"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
if len(sys.argv) > 3:
    import ctypes
    L = ctypes.CDLL(os.path.abspath(sys.argv[3]))
    for name, argtypes in _cabi.ALL_SIGNATURES.items():
        if not hasattr(L, name): continue
        fn = getattr(L, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
    L.nmrfit_last_error.argtypes = []; L.nmrfit_last_error.restype = ctypes.c_char_p
    _cabi._LIB = L
    print("library:", sys.argv[3])
sp, X = synth.make_workload("C3")
S, D = X.shape
P = (D - 4) // 3
with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
    ev.set_variant(_cabi.variant_id(variant))
    dX, df = ev.dev_alloc(X.nbytes), ev.dev_alloc(8 * S)
    ev.upload(dX, X)
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < seconds:
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 0.8:
            for _ in range(16):
                ev.objective_batch_dev(S, P, dX, df)
            ev.synchronize()
            n += 16
        ev.prof_enable(16)
        for _ in range(16):
            ev.objective_batch_dev(S, P, dX, df)
        k, _, mhz = ev.prof_read()
        ev.prof_enable(0)
        print("t=%5.1f s  %s kernel %.4f ms (min %.4f)  shader clock %.0f MHz  (%d launches in the last 0.8 s)" % (
            time.perf_counter() - t_start, variant, np.mean(k), np.min(k), mhz, n), flush=True)

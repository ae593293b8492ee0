#!/usr/bin/env python3
"""Per-generation wall time of nmrfit_pso_run (the device-resident loop) for small swarms --
the regime of the reference's defaults (204 particles) where launches, not arithmetic, set the
pace.  Stopping tests disabled so every generation runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import synth, pso
from nmrfit_amd.equations import Evaluator

if len(sys.argv) > 1:      # another build of the library (A/B): tools/small_swarm_timing.py nmrfit_amd/lib/libab_x.so
    import ctypes
    from nmrfit_amd import _cabi
    L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
    for name, argtypes in _cabi.ALL_SIGNATURES.items():
        if not hasattr(L, name): continue
        fn = getattr(L, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
    L.nmrfit_last_error.argtypes = []; L.nmrfit_last_error.restype = ctypes.c_char_p
    _cabi._LIB = L
    print("library:", sys.argv[1])
SHAPES = [(50, 4096, 6), (204, 4096, 6), (204, 16384, 12), (512, 4096, 6), (1024, 4096, 6), (204, 65536, 24), (4096, 65536, 24)]
for (S, N, P) in SHAPES:
    sp = synth.make_spectrum(N, P, seed=1)
    with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
        if os.environ.get("NMRFIT_HANDOVER"):               # fast (default) / fenced / two_launch
            sw.set_handover(os.environ["NMRFIT_HANDOVER"])
        sw.run(50, check_every=50)          # warm-up
        gens = 2000 if S * N * P < 1e9 else 200
        t0 = time.perf_counter()
        sw.run(gens, check_every=100)
        dt = time.perf_counter() - t0
        st = sw.status()
        sw.close()
    print("S=%5d N=%6d P=%3d: %7.2f us per generation (%d generations, fg=%.6g)  -> %.3g units/s" % (
        S, N, P, dt / gens * 1e6, st["iteration"], st["fg"], S * N * P * gens / dt))

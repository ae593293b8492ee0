#!/usr/bin/env python3
"""Per-generation wall time of nmrfit_pso_run (the device-resident loop) for small swarms --
the regime of the reference's defaults (204 particles) where launches, not arithmetic, set the
pace.  Stopping tests disabled so every generation runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import synth, pso
from nmrfit_amd.equations import Evaluator

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

#!/usr/bin/env python3
"""Where does the far-field kernel start to pay?  (nmrfit_amd.utils.default_variant's
grid x peaks >= 1e5 threshold had been measured at 204 particles only.)

Per-generation wall time of nmrfit_pso_run (what fit() runs; stopping tests off) with the DEFAULT and
the FARFIELD objective kernel, interleaved A/B/A/B in one process on one device, for swarms of 204,
1024 and 4096 particles over a ladder of grid x peaks products either side of 1e5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrfit_amd import _cabi, pso, synth
from nmrfit_amd.equations import Evaluator
from nmrfit_amd.utils import default_variant

LADDER = [(4096, 6), (8192, 6), (4096, 12), (4096, 24), (8192, 12), (16384, 6), (16384, 12), (32768, 6), (8192, 24),
          (32768, 12), (16384, 24), (65536, 24)]
print("%6s %7s %4s %9s | %12s %12s  %s" % ("S", "N", "P", "N*P", "default us", "farfield us", "farfield/default   (fit() picks)"))
for S in (204, 1024, 4096):
    for N, P in LADDER:
        if S * N * P > 1.7e9 and (N, P) != (65536, 24):
            continue
        sp = synth.make_spectrum(N, P, seed=1)
        gens = max(30, min(1500, int(4e9 / (S * N * P))))
        t = {"default": [], "farfield": []}
        with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
            for rep in range(3):
                for name in ("default", "farfield"):
                    ev.set_variant(_cabi.variant_id(name))
                    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
                    sw.run(max(10, gens // 10), check_every=1000)
                    t0 = time.perf_counter()
                    sw.run(gens, check_every=1000)
                    t[name].append((time.perf_counter() - t0) / gens * 1e6)
                    sw.close()
        d, f = min(t["default"]), min(t["farfield"])
        print("%6d %7d %4d %9d | %12.2f %12.2f  %.3f   (%s)" % (S, N, P, N * P, d, f, f / d, default_variant(N, P)), flush=True)

#!/usr/bin/env python3
"""Where does the far-field kernel start to pay WITH the imaginary channel (fit_im=True, "sum")?  Round 6: the far-field
kernel with the all-peak imaginary model now fits three waves per SIMD (pair expansions + quarter-interval Dawson table).
Per-generation wall time of nmrfit_pso_run with the DEFAULT and the FARFIELD kernel, interleaved, for 204 and 1024
particles over the ladder of tools/archive/variant_threshold.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import _cabi, pso, synth
from nmrfit_amd.equations import Evaluator

LADDER = [(4096, 6), (4096, 24), (8192, 12), (16384, 6), (16384, 12), (8192, 24), (32768, 12), (16384, 24), (65536, 24)]
for mode in (True, "sum"):
    print("fit_im = %r" % (mode,))
    print("%6s %7s %4s %9s | %12s %12s  %s" % ("S", "N", "P", "N*P", "default us", "farfield us", "farfield/default"))
    for S in (204, 1024):
        for N, P in LADDER:
            sp = synth.make_spectrum(N, P, seed=1, physical=True)
            gens = max(20, min(600, int(1.5e9 / (S * N * P))))
            t = {"default": [], "farfield": []}
            with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
                ev.set_fit_im(mode)
                for rep in range(2):
                    for name in ("default", "farfield"):
                        ev.set_variant(_cabi.variant_id(name))
                        sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
                        sw.run(max(5, gens // 10), check_every=1000)
                        t0 = time.perf_counter()
                        sw.run(gens, check_every=1000)
                        t[name].append((time.perf_counter() - t0) / gens * 1e6)
                        sw.close()
            d, f = min(t["default"]), min(t["farfield"])
            print("%6d %7d %4d %9d | %12.2f %12.2f  %.3f" % (S, N, P, N * P, d, f, f / d), flush=True)

#!/usr/bin/env python3
"""Which shader clock does a latency-bound small-swarm run see?  The objective kernel's in-kernel
clock probe (nmrfit_prof_*: s_memtime / s_memrealtime of workgroup 0) during launch-per-phase
generations of small swarms, against the C3 shape."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrfit_amd import synth, pso
from nmrfit_amd.equations import Evaluator

for (S, N, P, gens) in [(204, 4096, 6, 3000), (1024, 4096, 6, 3000), (204, 65536, 24, 1000), (4096, 65536, 24, 300)]:
    sp = synth.make_spectrum(N, P, seed=1)
    with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
        sw.run(gens, check_every=gens)              # settle the clocks on this workload
        ev.prof_enable(64)
        sw.run(64, check_every=64)
        k, _, mhz = ev.prof_read()
        ev.prof_enable(0)
        sw.close()
    print("S=%5d N=%6d P=%3d: objective kernel %.2f us (median of 64), shader clock %.0f MHz" % (S, N, P, np.median(k) * 1e3, mhz))

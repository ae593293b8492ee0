#!/usr/bin/env python3
"""Time to solution of nmrfit_amd.fit with the reference's defaults (swarmsize 204, maxiter 2000,
utils.py:177-178) on a synthetic 6-peak, 4096-point spectrum: wall time of the whole call
(weights, context, swarm, summary off), with pyswarm's stopping rule armed and with it disabled
(all 2000 generations).  For scale: the reference evaluates 204 x 2001 objectives at ~0.5 ms each
on one core = ~200 s for the full 2000 generations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmrfit_amd
from nmrfit_amd import synth

sp = synth.make_spectrum(4096, 6, seed=1)
data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False, options={"maxiter": 5, "seed": 1})   # load the library
for label, extra in (("stopping rule armed (minstep = minfunc = 1e-8)", {}),
                     ("stopping rule off: all 2000 generations", {"minstep": -1.0, "minfunc": -1.0})):
    t0 = time.perf_counter()
    r = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False, options=dict({"seed": 7}, **extra))
    dt = time.perf_counter() - t0
    print("%-50s %.1f ms, error %.6g" % (label, dt * 1e3, r.error))

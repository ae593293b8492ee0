#!/usr/bin/env python3
"""Per-generation time of nmrfit_pso_run against the number of segments per particle (forced with
NMRFIT_TARGET_WAVES, read at context creation), interleaved A/B/A/B in one process on one device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrfit_amd import synth, pso
from nmrfit_amd.equations import Evaluator

for (S, N, P, segs) in [(1024, 4096, 6, (0, 2, 4, 8)), (512, 4096, 6, (0, 2, 4, 8)), (204, 4096, 6, (0, 4, 8)),
                        (204, 16384, 12, (0, 4, 8, 16)), (2048, 4096, 6, (0, 1, 2, 4))]:
    sp = synth.make_spectrum(N, P, seed=1)
    res = {}
    for rep in range(3):
        for nseg in segs:
            if nseg:
                os.environ["NMRFIT_TARGET_WAVES"] = str(S * nseg)
            else:
                os.environ.pop("NMRFIT_TARGET_WAVES", None)
            with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
                sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
                sw.run(100, check_every=100)
                t0 = time.perf_counter()
                sw.run(1500, check_every=500)
                dt = (time.perf_counter() - t0) / 1500 * 1e6
                geom = ev.last_launch()["segments"]
                sw.close()
            res.setdefault((nseg, geom), []).append(dt)
    print("S=%5d N=%6d P=%3d  " % (S, N, P) + "  ".join("%s->%d: %s us" % ("auto" if k[0] == 0 else k[0], k[1], "/".join("%.1f" % v for v in vals)) for k, vals in res.items()))

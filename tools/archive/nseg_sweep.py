#!/usr/bin/env python3
"""Objective kernel time (HIP events around the kernel alone) against the number of segments per
particle, forced through NMRFIT_TARGET_WAVES (read at context creation)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrfit_amd import synth
from nmrfit_amd.equations import Evaluator

shapes = [(204, 4096, 6), (512, 4096, 6), (1024, 4096, 6), (2048, 4096, 6), (4096, 4096, 6), (204, 16384, 12), (1024, 16384, 12), (2048, 16384, 12),
          (512, 65536, 24), (1024, 65536, 24), (2048, 65536, 24), (4096, 65536, 24)]
for (S, N, P) in shapes:
    sp = synth.make_spectrum(N, P, seed=1)
    X = synth.make_swarm(sp["lower"], sp["upper"], S, seed=2, x_true=sp["x_true"])
    row = []
    for nseg in (0, 1, 2, 4, 8, 16):
        if nseg:
            os.environ["NMRFIT_TARGET_WAVES"] = str(S * nseg)
        else:
            os.environ.pop("NMRFIT_TARGET_WAVES", None)
        with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
            dX, df = ev.dev_alloc(X.nbytes), ev.dev_alloc(8 * S)
            ev.upload(dX, X)
            reps = 200 if S * N * P < 1e9 else 20
            for _ in range(reps):
                ev.objective_batch_dev(S, P, dX, df)
            ev.prof_enable(reps)
            for _ in range(reps):
                ev.objective_batch_dev(S, P, dX, df)
            k = ev.prof_read()[0]
            ev.prof_enable(0)
            geom = ev.last_launch()
            ev.dev_free(dX); ev.dev_free(df)
        row.append("%s->%d: %.2f us" % ("auto" if nseg == 0 else str(nseg), geom["segments"], np.median(k) * 1e3))
    print("S=%5d N=%6d P=%3d  " % (S, N, P) + "  ".join(row))

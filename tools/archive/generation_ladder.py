#!/usr/bin/env python3
"""Per-generation wall time of nmrfit_pso_run (stopping tests off) and kernel-only time of objective_batch over a ladder
of swarm x grid x peaks shapes, with the launch geometry the library picks (segments per particle, waves per workgroup).
    python tools/generation_ladder.py            (NMRFIT_SEG_RULE=3: the round-3 segment rule, for A/B)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import synth, pso
from nmrfit_amd.equations import Evaluator

SHAPES = [(204, 4096, 6), (512, 4096, 6), (1024, 4096, 6), (2048, 4096, 6), (4096, 4096, 6), (8192, 4096, 6), (204, 16384, 12),
          (512, 16384, 12), (1024, 16384, 12), (2048, 16384, 12), (4096, 16384, 12), (204, 65536, 24), (512, 65536, 24),
          (1024, 65536, 24), (2048, 65536, 24), (4096, 65536, 24)]
for (S, N, P) in SHAPES:
    sp = synth.make_spectrum(N, P, seed=1)
    X = synth.make_swarm(sp["lower"], sp["upper"], S, seed=2, x_true=sp["x_true"])
    with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        dX, df = ev.dev_alloc(X.nbytes), ev.dev_alloc(8 * S)
        ev.upload(dX, X)
        reps = 100 if S * N * P < 1e9 else 15
        for _ in range(reps):
            ev.objective_batch_dev(S, P, dX, df)
        ev.prof_enable(reps)
        for _ in range(reps):
            ev.objective_batch_dev(S, P, dX, df)
        k = np.median(ev.prof_read()[0]) * 1e3
        ev.prof_enable(0)
        geom = ev.last_launch()
        ev.dev_free(dX); ev.dev_free(df)
        sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
        gens = 1000 if S * N * P < 1e9 else 100
        sw.run(max(20, gens // 10), check_every=1000)
        t0 = time.perf_counter()
        sw.run(gens, check_every=1000)
        g = (time.perf_counter() - t0) / gens * 1e6
        n_launch = sw.last_launches()
        sw.close()
    print("S=%5d N=%6d P=%3d: %2d segments, %d waves per workgroup | objective kernel %8.2f us | generation %8.2f us (%d launches)"
          % (S, N, P, geom["segments"], geom["waves_per_workgroup"], k, g, n_launch), flush=True)

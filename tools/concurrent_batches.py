#!/usr/bin/env python3
"""Do the tails of device batches overlap?  Four batches of 50 default fits with pyswarm's rule, run one after the other
against two (or three) at a time from runner threads (batch creation outside the timed part).
    python tools/concurrent_batches.py"""
import os, sys, time, threading
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nmrfit_amd import synth
from nmrfit_amd.batch import FitBatch

specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(8)]
def make(first, K):
    sps = [specs[(first + k) % 8] for k in range(K)]
    return FitBatch([(q["w"], q["u"], q["v"], q["weights"]) for q in sps], [q["lower"] for q in sps], [q["upper"] for q in sps],
                    swarmsize=204, seeds=[7 + first + k for k in range(K)])
make(0, 8).run(5, 5)
for K, nb in ((50, 4), (50, 8), (100, 4), (25, 8)):
    for runners in (1, 2, 3, 1, 2, 3):
        fbs = [make(K * b, K) for b in range(nb)]
        t0 = time.perf_counter()
        if runners == 1:
            for fb in fbs:
                fb.run(2000, 64)
        else:
            with ThreadPoolExecutor(max_workers=runners) as pool:
                list(pool.map(lambda fb: fb.run(2000, 64), fbs))
        dt = time.perf_counter() - t0
        gens = [st["iteration"] for fb in fbs for st in fb.status()]
        for fb in fbs:
            fb.close()
        print("%d batches of %3d fits, %d at a time: %7.1f ms = %7.1f fits/s (generations mean %.0f max %d)" % (nb, K, runners, dt * 1e3, nb * K / dt, np.mean(gens), max(gens)), flush=True)

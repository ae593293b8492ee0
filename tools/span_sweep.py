#!/usr/bin/env python3
"""fit_many on 200 (and 1000) default jobs with pyswarm's rule: jobs per device batch (core.BATCH_JOBS) x check_every.
    python tools/span_sweep.py"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmrfit_amd
from nmrfit_amd import core, synth

specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(8)]
def jobs(n, ce):
    return [dict(data=synth.SynthData(specs[k % 8]["w"], specs[k % 8]["u"], specs[k % 8]["v"], specs[k % 8]["peaks"]),
                 lower=list(specs[k % 8]["lower"]), upper=list(specs[k % 8]["upper"]), options={"seed": 7 + k, "check_every": ce}) for k in range(n)]
with contextlib.redirect_stdout(io.StringIO()):
    nmrfit_amd.fit_many(jobs(8, 64), generate=True)
for n in (200, 1000):
    for bj in (32, 40, 50, 64, 100, 200):
        for ce in (32, 64):
            core.BATCH_JOBS = bj
            best = {}
            for gen in (False, True):
                ts = []
                for rep in range(3):
                    with contextlib.redirect_stdout(io.StringIO()):
                        t0 = time.perf_counter()
                        nmrfit_amd.fit_many(jobs(n, ce), generate=gen)
                        ts.append(time.perf_counter() - t0)
                best[gen] = min(ts)
            print("jobs %4d  BATCH_JOBS %3d  check_every %2d: fit only %7.1f fits/s, with generate %7.1f (%.2f)" % (n, bj, ce, n / best[False], n / best[True], best[False] / best[True]), flush=True)

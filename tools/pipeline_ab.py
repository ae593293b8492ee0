#!/usr/bin/env python3
"""fit_many on 200 / 1000 default jobs with pyswarm's rule: batches driven at once (core.RUN_AT_ONCE) x where the read-back
runs (core.READ_ON_RUNNER), interleaved on one device, best of three each.   python tools/pipeline_ab.py"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmrfit_amd
from nmrfit_amd import core, synth

specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(8)]
def jobs(n):
    return [dict(data=synth.SynthData(specs[k % 8]["w"], specs[k % 8]["u"], specs[k % 8]["v"], specs[k % 8]["peaks"]),
                 lower=list(specs[k % 8]["lower"]), upper=list(specs[k % 8]["upper"]), options={"seed": 7 + k}) for k in range(n)]
with contextlib.redirect_stdout(io.StringIO()):
    nmrfit_amd.fit_many(jobs(8), generate=True)
configs = [(1, False), (2, False), (2, True), (3, True)]
for n in (200, 1000):
    best = {(c, g): 1e9 for c in configs for g in (False, True)}
    for rep in range(3):
        for c in configs:
            core.RUN_AT_ONCE, core.READ_ON_RUNNER = c
            for g in (False, True):
                with contextlib.redirect_stdout(io.StringIO()):
                    t0 = time.perf_counter()
                    nmrfit_amd.fit_many(jobs(n), generate=g)
                    best[(c, g)] = min(best[(c, g)], time.perf_counter() - t0)
    for c in configs:
        print("jobs %4d  at once %d  read on %-6s: fit only %7.1f fits/s, with generate %7.1f (%.2f)" % (n, c[0], "runner" if c[1] else "store", n / best[(c, False)], n / best[(c, True)], best[(c, False)] / best[(c, True)]), flush=True)

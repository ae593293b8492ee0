#!/usr/bin/env python3
"""Repeat nmrfit_amd.fit_many(generate=True) on a mixed job list (lengths 3000 ... 6000, swarm sizes 100 ... 204, 2 ... 8 peaks;
three threads in flight per call) and compare every array of every run with the first run's, bit for bit.
    python tools/pipeline_stress.py [runs] [jobs]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmrfit_amd
from nmrfit_amd import synth

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(5)
lengths = rng.integers(3000, 6001, 24)
specs = [synth.make_spectrum(int(m), 2 + k % 7, seed=300 + k, physical=True) for k, m in enumerate(lengths)]


def jobs():
    return [dict(data=synth.SynthData(specs[k % 24]["w"], specs[k % 24]["u"], specs[k % 24]["v"], specs[k % 24]["peaks"]),
                 lower=list(specs[k % 24]["lower"]), upper=list(specs[k % 24]["upper"]),
                 options={"seed": 11 + k, "swarmsize": 100 + 13 * (k % 9)}) for k in range(n)]


first, t0 = None, time.perf_counter()
for r in range(runs):
    with contextlib.redirect_stdout(io.StringIO()):
        res = nmrfit_amd.fit_many(jobs(), generate=True if r % 2 == 0 else 1.5)
    key = [(f.params.copy(), f.error, f.u.copy(), f.imag_contribs[-1].copy(), f.data.V.copy()) for f in res]
    if r < 2:
        first = first or {}
        first[r % 2] = key
    else:
        for k, (a, b) in enumerate(zip(key, first[r % 2])):
            assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]), (r, k)
    print("run %d ok (%.1f s)" % (r, time.perf_counter() - t0), flush=True)
print("all %d runs identical" % runs)

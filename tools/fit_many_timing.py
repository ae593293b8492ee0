#!/usr/bin/env python3
"""End-to-end wall time of nmrfit_amd.fit_many on default-shape jobs (204 particles x 4096 points x 6 peaks): host
preparation (weights, plans), device batch, results into FitUtility objects -- against a plain loop over nmrfit_amd.fit.
    python tools/fit_many_timing.py [jobs]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmrfit_amd
from nmrfit_amd import synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
jobs = []
for k in range(K):
    sp = synth.make_spectrum(4096, 6, seed=100 + k % 8)
    jobs.append((synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), list(sp["lower"]), list(sp["upper"])))
with contextlib.redirect_stdout(io.StringIO()):
    nmrfit_amd.fit(*jobs[0], summary=False, options={"seed": 1, "maxiter": 5})     # load the library, warm the device
for name, opts in (("stopping rule off (2000 generations each)", {"seed": 7, "minstep": -1.0, "minfunc": -1.0}),
                   ("pyswarm's stopping rule (defaults)", {"seed": 7})):
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        many = nmrfit_amd.fit_many(jobs, options=opts)
        t1 = time.perf_counter()
        loop = [nmrfit_amd.fit(*j, summary=False, options=opts) for j in jobs[:8]]
        t2 = time.perf_counter()
    assert all((a.params == b.params).all() and a.error == b.error for a, b in zip(many, loop))
    print("%-42s fit_many(%d jobs): %7.1f ms = %6.1f fits/s;  plain loop: %6.1f ms per fit = %5.1f fits/s  (results identical)"
          % (name, K, (t1 - t0) * 1e3, K / (t1 - t0), (t2 - t1) / 8 * 1e3, 8 / (t2 - t1)), flush=True)

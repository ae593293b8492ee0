#!/usr/bin/env python3
"""Device batches of spectra of DIFFERENT lengths (cropped per dataset, nmrfit/containers.py:112-130) against batches of
equal length: K default-size swarms (204 particles, 6 peaks), lengths drawn from 3000 ... 6000 (mean 4500), against K
fits of 4096 and of 4608 points (the 512-multiple next to the mean).  fits/s and units/s, stopping rule off and on.
    python tools/ragged_batch_timing.py [fits]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nmrfit_amd import synth
from nmrfit_amd.batch import FitBatch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(11)
ragged = [int(n) for n in rng.integers(3000, 6001, K)]
cases = (("equal 4096", [4096] * K), ("equal 4608", [4608] * K), ("ragged 3000..6000 (mean %d)" % np.mean(ragged), ragged),
         ("sorted ragged", sorted(ragged)))
cache = {}


def spec(n, k):
    if (n, k % 8) not in cache:
        cache[(n, k % 8)] = synth.make_spectrum(n, 6, seed=100 + k % 8)
    return cache[(n, k % 8)]


def run(lengths, rule, maxiter=2000):
    sps = [spec(n, k) for k, n in enumerate(lengths)]
    t0 = time.perf_counter()
    with FitBatch([(q["w"], q["u"], q["v"], q["weights"]) for q in sps], [q["lower"] for q in sps], [q["upper"] for q in sps],
                  swarmsize=204, seeds=list(range(7, 7 + len(sps))), **rule) as fb:
        t1 = time.perf_counter()
        fb.run(maxiter, 64)
        st = fb.status()
        t2 = time.perf_counter()
    gens = np.array([q["iteration"] for q in st], dtype=float)
    units = float(np.sum(204.0 * np.array(lengths) * 6 * (gens + 1)))
    return t2 - t0, t2 - t1, units


run([4096] * 8, {}, 5)
run(ragged[:8], {}, 5)
for rule_name, rule in (("stopping rule off", dict(minstep=-1.0, minfunc=-1.0)), ("pyswarm's rule", {})):
    print(rule_name + ", %d fits:" % K)
    for name, lengths in cases:
        best = min((run(lengths, rule) for _ in range(2)), key=lambda r: r[0])
        print("    %-34s %8.1f ms (run %8.1f) = %7.1f fits/s, %.3e units/s" % (name, best[0] * 1e3, best[1] * 1e3, K / best[0], best[2] / best[1]), flush=True)

#!/usr/bin/env python3
"""The reference's per-spectrum script end to end -- fit -> generate_result -> calculate_area_fraction (README.md:64-72;
nmrfit/utils.py:164-189, 226-295, 297-322) -- on default-shape jobs (204 particles x 4096 points x 6 peaks): the plain
loop over nmrfit_amd.fit against nmrfit_amd.fit_many(jobs, generate=True), and fit_many without the reconstruction.
    python tools/readme_pipeline.py [jobs] [loop_jobs]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nmrfit_amd
from nmrfit_amd import synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(8)]


def jobs(n):
    return [dict(data=synth.SynthData(specs[k % 8]["w"], specs[k % 8]["u"], specs[k % 8]["v"], specs[k % 8]["peaks"]),
                 lower=list(specs[k % 8]["lower"]), upper=list(specs[k % 8]["upper"])) for k in range(n)]


def with_seed(js, opts):
    return [dict(j, options=dict(opts, seed=7 + k)) for k, j in enumerate(js)]


with contextlib.redirect_stdout(io.StringIO()):
    nmrfit_amd.fit_many(with_seed(jobs(8), {"maxiter": 5}), generate=True)     # load the library and the kernels, warm the device
for name, opts in (("stopping rule off (2000 generations each)", {"minstep": -1.0, "minfunc": -1.0}),
                   ("pyswarm's stopping rule (defaults)", {})):
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        fit_only = nmrfit_amd.fit_many(with_seed(jobs(K), opts))
        t1 = time.perf_counter()
        full = nmrfit_amd.fit_many(with_seed(jobs(K), opts), generate=True)
        fractions = [f.calculate_area_fraction() for f in full]
        t2 = time.perf_counter()
        loop = []
        for j in with_seed(jobs(L), opts):
            f = nmrfit_amd.fit(j["data"], j["lower"], j["upper"], summary=False, options=j["options"])
            f.generate_result()
            loop.append((f, f.calculate_area_fraction()))
        t3 = time.perf_counter()
        loop_fit = [nmrfit_amd.fit(j["data"], j["lower"], j["upper"], summary=False, options=j["options"])
                    for j in with_seed(jobs(L), opts)]
        t4 = time.perf_counter()
    for a, (b, frac), c in zip(full, loop, fractions):
        assert np.array_equal(a.params, b.params) and a.error == b.error and frac == c
        assert np.array_equal(a.V, b.V) and np.array_equal(a.u, b.u) and np.array_equal(a.imag_contribs[-1], b.imag_contribs[-1])
    print("%s, %d jobs:" % (name, K))
    print("    fit_many, fit only                        %8.1f ms = %7.1f fits/s" % ((t1 - t0) * 1e3, K / (t1 - t0)))
    print("    fit_many(generate=True) + area fractions  %8.1f ms = %7.1f fits/s  (%.2f of the fit-only rate)"
          % ((t2 - t1) * 1e3, K / (t2 - t1), (t1 - t0) / (t2 - t1)))
    print("    plain loop fit -> generate_result -> areas %7.2f ms per spectrum = %6.1f fits/s (fit alone: %.2f ms); "
          "results identical" % ((t3 - t2) / L * 1e3, L / (t3 - t2), (t4 - t3) / L * 1e3), flush=True)

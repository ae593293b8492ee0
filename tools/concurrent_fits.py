#!/usr/bin/env python3
"""Fits per second when several host threads each run nmrfit_amd.fit() on their own spectra (one context per fit, its
own HIP stream; the library releases the GIL inside its calls): a 204-particle swarm fills a fraction of the chip, so
independent fits overlap on it.  Reference defaults (204 particles, 2000 generations, stopping rule off so that every
fit does the same work), 6-peak 4096-point synthetic spectra.
    python tools/concurrent_fits.py [fits per thread]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmrfit_amd
from nmrfit_amd import synth

per_thread = int(sys.argv[1]) if len(sys.argv) > 1 else 6
specs = []
for k in range(8):
    sp = synth.make_spectrum(4096, 6, seed=100 + k)
    specs.append((synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), list(sp["lower"]), list(sp["upper"])))
opts = {"seed": 7, "minstep": -1.0, "minfunc": -1.0}
nmrfit_amd.fit(*specs[0], summary=False, options=dict(opts, maxiter=5))     # load the library, warm the device
single = None
for nthreads in (1, 2, 3, 4, 6, 8):
    errs, results = [], [None] * nthreads
    start = threading.Barrier(nthreads + 1)

    def work(t):
        try:
            start.wait()
            for i in range(per_thread):
                r = nmrfit_amd.fit(*specs[t], summary=False, options=opts)
            results[t] = r.error
        except BaseException as e:
            errs.append(e)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for th in ths:
        th.start()
    start.wait()
    t0 = time.perf_counter()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    assert not errs, errs[0]
    rate = nthreads * per_thread / dt
    single = single or rate
    print("%d thread%s: %5.1f fits/s (%.1f ms per fit per thread, %.2fx of one thread)  errors %s" % (
        nthreads, " " if nthreads == 1 else "s", rate, dt / per_thread * 1e3, rate / single,
        " ".join("%.6g" % e for e in results[:3])), flush=True)

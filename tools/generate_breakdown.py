#!/usr/bin/env python3
"""Where the time of a batch's reconstruction goes (FitBatch.generate after a run): the C call (launch + copies back)
against fresh and against already-touched host arrays, and the Python around it.
    python tools/generate_breakdown.py [fits]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nmrfit_amd import _cabi, synth
from nmrfit_amd.batch import FitBatch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(K)]
with FitBatch([(q["w"], q["u"], q["v"], q["weights"]) for q in specs], [q["lower"] for q in specs], [q["upper"] for q in specs],
              swarmsize=204, seeds=list(range(K))) as fb:
    fb.run(100, 64)
    fb.generate()
    for rep in range(3):
        t0 = time.perf_counter()
        st = fb.status(); best = fb.best()
        t1 = time.perf_counter()
        res = fb.generate()
        t2 = time.perf_counter()
        print("status+best %.2f ms, generate() %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    N, rows = 4096, int(fb.P.sum())
    for touched in (False, True, True):
        t0 = time.perf_counter()
        real = np.empty((rows, N)); imag = np.empty((rows, N)); fit = np.empty((K, 4, N)); data = np.empty((K, 2, N))
        if touched:
            for a in (real, imag, fit, data):
                a.fill(0.0)
        t1 = time.perf_counter()
        _cabi.check(fb._lib.nmrfit_batch_contributions(fb._h, N, None, _cabi.ptr(real), _cabi.ptr(imag), _cabi.ptr(fit), _cabi.ptr(data)))
        t2 = time.perf_counter()
        mb = (real.nbytes + imag.nbytes + fit.nbytes + data.nbytes) / 1e6
        print("arrays %s: alloc%s %.2f ms, C call %.2f ms for %.1f MB = %.1f GB/s" % ("touched" if touched else "fresh", "+fill" if touched else "", (t1 - t0) * 1e3, (t2 - t1) * 1e3, mb, mb / (t2 - t1) / 1e3))
    # only the small outputs
    t1 = time.perf_counter()
    _cabi.check(fb._lib.nmrfit_batch_contributions(fb._h, N, None, None, None, _cabi.ptr(fit), None))
    t2 = time.perf_counter()
    print("fit_out only (%.1f MB): %.2f ms" % (fit.nbytes / 1e6, (t2 - t1) * 1e3))

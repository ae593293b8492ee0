#!/usr/bin/env python3
"""Default fits per second through the device-batched path (nmrfit_amd.batch.FitBatch): K fits of the reference's
default shape (204 particles x 4096 points x 6 peaks, 2000 generations, stopping rule off so that every fit does the
same work) in one batch, one launch per generation for all of them -- by K and launch geometry.
    python tools/batch_fits.py [generations] [K list] [geometries]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nmrfit_amd import synth
from nmrfit_amd.batch import FitBatch

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
Ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 12, 16, 20, 32, 40, 64]
geoms = sys.argv[3].split(",") if len(sys.argv) > 3 else ["workgroup", "wave"]
N, P, S = (int(os.environ.get(k, d)) for k, d in (("BF_N", 4096), ("BF_P", 6), ("BF_S", 204)))
specs = []
for k in range(max(Ks)):
    sp = synth.make_spectrum(N, P, seed=100 + k % 8)
    specs.append(sp)
print("# %d particles x %d points x %d peaks, %d generations, stopping rule off" % (S, N, P, gens))
for K in Ks:
    for geom in geoms:
        spectra = [(sp["w"], sp["u"], sp["v"], sp["weights"]) for sp in specs[:K]]
        t0 = time.perf_counter()
        fb = FitBatch(spectra, [sp["lower"] for sp in specs[:K]], [sp["upper"] for sp in specs[:K]], swarmsize=S,
                      seeds=list(range(7, 7 + K)), minstep=-1.0, minfunc=-1.0)
        try:
            fb.set_geometry(geom)
        except Exception as e:
            print("K=%3d %-9s not available (%s)" % (K, geom, e)); fb.close(); continue
        t1 = time.perf_counter()
        fb.run(50, 50)                    # warm
        t2 = time.perf_counter()
        fb.run(gens, 64)
        best = fb.best()
        t3 = time.perf_counter()
        g = fb.geometry()
        fb.close()
        per_gen = (t3 - t2) / gens
        print("K=%3d %-9s %6d workgroups x %d waves: %7.2f us per generation = %5.2f us per fit-generation -> %6.1f fits/s "
              "(create %.1f ms; %.3g units/s; err0 %.6g)" % (K, geom, g["workgroups"], g["waves_per_workgroup"], per_gen * 1e6,
              per_gen * 1e6 / K, K / (t3 - t2 + (t1 - t0)) * (gens / 2000.0) ** 0 if gens == 2000 else K / (per_gen * 2000 + (t1 - t0)),
              (t1 - t0) * 1e3, K * S * N * P / per_gen, best[0][1]), flush=True)

#!/usr/bin/env python3
"""Largest relative difference of f from the REFERENCE's own values (tests/golden/, produced by
oracle/make_golden.py from the reference code) for every kernel variant, on the committed golden
sets.  Run on the GPU box; prints a small table for DESIGN.md."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nmrfit_amd import _cabi, synth
from nmrfit_amd.equations import Evaluator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {v: k for k, v in _cabi._VARIANT_NAMES.items()}
sets = [("objective_P6_N4096.npz", 4096, 6), ("objective_P12_N16384.npz", 16384, 12), ("objective_P24_N65536.npz", 65536, 24)]
print("%-10s" % "variant" + "".join("%22s" % s[0][10:-4] for s in sets))
for variant in sorted(_cabi.available_variants()):      # (the A/B forms: with NMRFIT_LIB=.../libnmrfit_amd_ab.so)
    row = "%-10s" % names[variant]
    for fn, N, P in sets:
        g = np.load(os.path.join(ROOT, "tests", "golden", fn))
        sp = synth.make_spectrum(N, P, seed=int(g["seed"]))
        with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
            ev.set_variant(variant)
            f = ev.objective_batch(g["X"])
        row += "%22.2e" % np.max(np.abs(f - g["f"]) / np.maximum(np.abs(g["f"]), 1e-6))
    print(row)
# ... and the far-field kernel against the direct one on spectra where nothing is sparse (broad overlapping lines)
sp, X = synth.make_workload("C3")
Xd = synth.make_dense_swarm(256, 24, seed=5, w_lo=float(sp["w"].min()), w_hi=float(sp["w"].max()))
with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
    out = {}
    for v in (_cabi.VARIANT_DEFAULT, _cabi.VARIANT_FARFIELD):
        ev.set_variant(v)
        out[v] = (ev.objective_batch(X[:512]), ev.objective_batch(Xd))
    for k, what in ((0, "C3 sparse swarm (512 particles)"), (1, "dense swarm, C3 shape (256 particles)")):
        a, b = out[_cabi.VARIANT_DEFAULT][k], out[_cabi.VARIANT_FARFIELD][k]
        print("farfield vs default, %s: max rel diff %.2e" % (what, np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-6))))

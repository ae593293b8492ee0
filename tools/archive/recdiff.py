#!/usr/bin/env python3
"""Point-by-point accuracy of the kernel variants on C3: residual rows and f of DEFAULT (0), NOREC (7)
and FARFIELD (6) against BASELINE (IEEE divide + libdevice exp2).  Run on the GPU box."""
import sys; sys.path.insert(0, ".")
import numpy as np
from nmrfit_amd import _cabi, synth, equations as eq
sp, X = synth.make_workload("C3")
with eq.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
    ev.set_variant(_cabi.VARIANT_BASELINE); Rb = ev.residual_batch(X[:8]); fb = ev.objective_batch(X[:64])
    for v in (0, 7, 6):
        ev.set_variant(v); R = ev.residual_batch(X[:8]); f = ev.objective_batch(X[:64])
        print("variant", v, "max |R - R_baseline| / max|R| = %.2e" % (np.abs(R - Rb).max() / np.abs(Rb).max()),
              " max rel f diff %.2e" % np.max(np.abs(f - fb) / fb))

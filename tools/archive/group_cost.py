#!/usr/bin/env python3
"""Dynamic cost of the two Lorentzian group forms: P = 8 (one group), pure Lorentzian lines on a
non-uniform grid (no Gaussian recurrence), variant 7 (general pair form) and variant 0 (two-operation
pair form) launched alternately.  Run under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU` and read
the per-dispatch counter: dispatches alternate 7, 0, 7, 0, ...

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d out -- python3 tools/group_cost.py [P]

P is an argument (default 8), so that python3 is the program directly after `--`: under a profiler
never go through `env VAR=... python3` (the profiler's preload has initialised the GPU, and
replacing such a process is refused on this pool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nmrfit_amd import synth
from nmrfit_amd.equations import Evaluator

S, N, P = 4096, 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 8
sp = synth.make_spectrum(N, P, seed=1)
rng = np.random.default_rng(0)
w = np.sort(sp["w"] + 1e-7 * rng.standard_normal(N))        # non-uniform: the recurrence is off
X = synth.make_swarm(sp["lower"], sp["upper"], S, seed=2)
X[:, 2] = 1.0                                                 # pure Lorentzian
with Evaluator(w, sp["u"], sp["v"], sp["weights"]) as ev:
    dX = ev.dev_alloc(X.nbytes); df = ev.dev_alloc(8 * S); ev.upload(dX, X)
    for rep in range(4):
        for variant in (7, 0):
            ev.set_variant(variant)
            ev.timer_begin()
            ev.objective_batch_dev(S, P, dX, df)
            ms = ev.timer_end()
            print("variant %d: %.4f ms" % (variant, ms))

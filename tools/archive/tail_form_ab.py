#!/usr/bin/env python3
"""A/B of the ways a single-rank swarm generation ends (csrc/pso_update.h, PsoFused::tail), interleaved on one
device: `two` = a single-workgroup launch after the objective launch (argmin over fp, candidate record, fold);
`deferred` = the NEXT objective launch's prologue does it, every workgroup for itself.  Per-generation wall time of
nmrfit_pso_run with the stopping tests off, and whether both end in bit-identical states.  (A third form, `ticket`
-- the workgroup that draws the last ticket of the objective launch finishes the generation -- was measured with this
script before it was removed: profiles/r04/generation_tail_forms_ab.txt.)
    python tools/tail_form_ab.py [rounds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import synth, pso
from nmrfit_amd.equations import Evaluator

SHAPES = [(50, 4096, 6), (204, 4096, 6), (256, 4096, 6), (512, 4096, 6), (1024, 4096, 6), (2048, 4096, 6), (204, 16384, 12),
          (1024, 16384, 12), (4096, 65536, 24)]
FORMS = ("two", "deferred")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def make(ev, sp, S, form):
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
    sw.set_fused_tail(form != "two")
    return sw


for (S, N, P) in SHAPES:
    sp = synth.make_spectrum(N, P, seed=1)
    with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        gens = 2000 if S * N * P < 1e9 else 200
        times = {f: [] for f in FORMS}
        launches, states = {}, {}
        for r in range(rounds):
            for form in FORMS:
                sw = make(ev, sp, S, form)
                sw.run(50, check_every=1000)
                ev.synchronize()
                t0 = time.perf_counter()
                sw.run(gens, check_every=1000)
                times[form].append((time.perf_counter() - t0) / gens * 1e6)
                launches[form] = sw.last_launches()
                if r == 0:
                    st = sw.state()
                    x, f = sw.best()
                    states[form] = (st["x"], st["v"], st["p"], st["fp"], x, f, sw.status()["iteration"], sw.candidate())
                sw.close()
        same = all(all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(states["two"], states[f])) for f in FORMS[1:])
        print("S=%5d N=%6d P=%3d: " % (S, N, P) + "  ".join(
            "%s %7.2f us (%d launch%s)" % (f, np.median(times[f]), launches[f], "" if launches[f] == 1 else "es") for f in FORMS)
              + "   states %s" % ("bit-identical" if same else "DIFFER"), flush=True)

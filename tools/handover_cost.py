#!/usr/bin/env python3
"""What the cross-workgroup hand-over of the personal-best / argmin kernel costs per generation in its three forms
(nmrfit_pso_set_handover: fast = fence-free agent-scope stores, fenced = release / acquire, two_launch), on the shapes
that still hand over inside a launch in round 5: swarms of up to 1024 particles whose objective launch does not make a
workgroup the particle -- the imaginary channel (no eight-wave form), or fused personal bests switched off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import synth, pso
from nmrfit_amd.equations import Evaluator

SHAPES = [(204, 4096, 6, True, True), (1024, 4096, 6, True, True), (204, 4096, 6, "sum", True), (204, 4096, 6, False, False),
          (1024, 4096, 6, False, False)]
for (S, N, P, fit_im, fused) in SHAPES:
    sp = synth.make_spectrum(N, P, seed=1)
    row = []
    for mode in ("fast", "fenced", "two_launch"):
        with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
            ev.set_fit_im(fit_im)
            sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
            sw.set_handover(mode)
            sw.set_fused_pbest(fused)
            sw.run(50, check_every=50)
            gens = 2000
            t0 = time.perf_counter()
            sw.run(gens, check_every=100)
            dt = time.perf_counter() - t0
            row.append((mode, dt / gens * 1e6, sw.last_launches(), sw.status()["fg"]))
            sw.close()
    assert len({r[3] for r in row}) == 1, row
    print("S=%5d N=%5d P=%2d fit_im=%-5s fused_pbest=%-5s: " % (S, N, P, fit_im, fused) +
          "  ".join("%s %6.2f us (%d launches)" % (m, us, nl) for m, us, nl, _ in row), flush=True)

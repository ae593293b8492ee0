#!/usr/bin/env python3
"""Kernel time vs peak count / swarm size on the GPU box: separates the per-point overhead
(phase rotation, residual, loads) from the per-(point, peak) cost.  HIP-event timing of
objective-only launches on resident inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nmrfit_amd import synth
from nmrfit_amd.equations import Evaluator


def time_case(S, N, P, variant=0, reps=10, residual=False, fit_im=0):
    sp = synth.make_spectrum(N, P, seed=1)
    X = synth.make_swarm(sp["lower"], sp["upper"], S, seed=2, x_true=sp["x_true"])
    D = 4 + 3 * P
    with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        ev.set_variant(variant)
        ev.set_fit_im(fit_im)
        dX = ev.dev_alloc(S * D * 8); df = ev.dev_alloc(S * 8)
        dR = ev.dev_alloc(S * N * 8) if residual else None
        ev.upload(dX, X)
        run = (lambda: ev.residual_batch_dev(S, P, dX, dR, df)) if residual else (lambda: ev.objective_batch_dev(S, P, dX, df))
        for _ in range(3):
            run()
        ev.synchronize()
        ev.timer_begin()
        for _ in range(reps):
            run()
        ms = ev.timer_end() / reps
        g = ev.last_launch()
        ev.dev_free(dX); ev.dev_free(df)
        if dR: ev.dev_free(dR)
    return ms, g


if __name__ == "__main__":
    print("# S N P variant kernel_ms units/s  ns_per_point_particle  waves nseg")
    for (S, N, P) in [(4096, 65536, 24), (4096, 65536, 16), (4096, 65536, 8), (4096, 65536, 6), (4096, 65536, 1),
                      (4096, 65536, 0), (1024, 4096, 6), (204, 4096, 6), (50, 4096, 6), (204, 65536, 24),
                      (32768, 4096, 6)]:
        for variant in (0,):
            ms, g = time_case(S, N, P, variant)
            print("%6d %6d %3d v%d  %.4f ms  %.4g units/s  %.4f ns/(pt*particle)  waves %d nseg %d" % (
                S, N, P, variant, ms, S * N * max(P, 1) / (ms * 1e-3), ms * 1e6 / (S * N), g["waves"], g["segments"]))
    for fi, name in ((1, "fit_im=True (reference: last peak only)"), (2, "fit_im='sum' (all peaks)")):
        ms, g = time_case(4096, 65536, 24, 0, reps=3, fit_im=fi)
        print("C3 %s: %.4f ms  %.4g units/s" % (name, ms, 4096 * 65536 * 24 / (ms * 1e-3)))
        ms, g = time_case(1024, 4096, 6, 0, reps=10, fit_im=fi)
        print("C2 %s: %.4f ms" % (name, ms))
    ms, g = time_case(41, 16384, 12, 0, residual=True)
    print("C5 residual_batch B=41 N=16384 P=12: %.4f ms (%.4g units/s) waves %d nseg %d" % (ms, 41 * 16384 * 12 / (ms * 1e-3), g["waves"], g["segments"]))
    ms, g = time_case(4096, 65536, 24, 0, reps=3, residual=True)
    print("residual_batch B=4096 N=65536 P=24 (2.1 GB out): %.4f ms -> %.1f GB/s written" % (ms, 4096 * 65536 * 8 / (ms * 1e-3) / 1e9))

#!/usr/bin/env python3
"""Stress proof of the fence-free cross-XCD hand-over.

Swarms of up to 1024 particles finish the personal-best / argmin reduction inside ONE launch:
every workgroup of pso_select_kernel posts its minimum and its updated personal-best rows with
agent-scope write-through stores ordered by s_waitcnt vmcnt(0) -- no release fence -- and the
workgroup that draws the last ticket reads them all (csrc/pso.hip, NMRFIT_HANDOVER_FAST).  A stale
or torn read would not crash: it would silently pick a wrong global best or copy a half-written
row, and the swarm's trajectory would differ from then on for ever.

This tool runs the same swarm, from the same seed, twice on the same device:
    A  NMRFIT_HANDOVER_FAST        (the product default)
    B  NMRFIT_HANDOVER_TWO_LAUNCH  (the reduction as its own launch: nothing is handed over inside a
                                    launch, the kernel boundary orders everything)
and, every --fenced-every seeds, a third time with NMRFIT_HANDOVER_FENCED (release / acquire
fences), for `--gens` generations each with the stopping tests off, then compares x, v, p, fx, fp,
the best position and value bit for bit.  About 0.7 S particles improve their personal best in
every one of the first ~700 generations (measured with the numpy mirror), so every hand-over
carries fresh rows; a new seed starts a new swarm before that dries up.

One exchange = one generation's hand-over (one pso_select_kernel launch in mode A); the log also
counts the posts (exchanges x workgroups).  Exit code 1 on any mismatch.

    python tools/handover_stress.py --exchanges 2500000            # per shape; four shapes -> 1e7
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import pso, synth                      # noqa: E402
from nmrfit_amd.equations import Evaluator             # noqa: E402

SHAPES = {"204": (204, 4096, 6), "256": (256, 4096, 6), "512": (512, 4096, 6), "1024": (1024, 4096, 6),
          "50": (50, 4096, 6), "256x2048": (256, 2048, 3), "204x16384": (204, 16384, 12)}


def run(ev, sp, S, seed, gens, mode):
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
    if mode == "one_launch":
        # round 4: the product default for single-rank swarms of up to 1024 particles where a workgroup is a particle --
        # personal bests inside the objective launch, the fold deferred into the next launch's prologue (--one-launch:
        # this against TWO_LAUNCH with everything after the objective in kernels of its own)
        pass
    else:
        # the select kernel must run at every shape here: where a workgroup is a particle the product default does the
        # personal bests (and, small swarms, the whole generation) inside the objective launch
        sw.set_fused_pbest(False)
        sw.set_handover(mode)
    sw.init()
    ev.synchronize()
    t0 = time.perf_counter()
    sw.run(gens, check_every=gens)
    dt = time.perf_counter() - t0
    st = sw.state()
    st["best_x"], st["best_f"] = sw.best()
    st["status"] = sw.status()
    st["launches"] = sw.last_launches()
    sw.close()
    return st, dt


def same(a, b):
    bad = [k for k in ("x", "v", "p", "fx", "fp", "best_x") if not np.array_equal(a[k], b[k])]
    if a["best_f"] != b["best_f"]:
        bad.append("best_f")
    if a["status"] != b["status"]:
        bad.append("status")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exchanges", type=int, default=2500000, help="generations of mode A per shape")
    ap.add_argument("--gens", type=int, default=500, help="generations per seed")
    ap.add_argument("--shapes", default="204,256,512,1024")
    ap.add_argument("--fenced-every", type=int, default=25)
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--one-launch", action="store_true", help="mode A = the one-launch generation (product default, "
                    "<= 1024 particles) instead of the select kernel's fence-free hand-over")
    a = ap.parse_args()
    mode_a = "one_launch" if a.one_launch else "fast"
    total_ex = total_posts = total_bad = 0
    t_start = time.perf_counter()
    for name in a.shapes.split(","):
        S, N, P = SHAPES[name]
        wgs = (S + 3) // 4
        sp = synth.make_spectrum(N, P, seed=4)
        ex = bad_seeds = n_seeds = n_fenced = 0
        t_fast = t_two = t_fenced = 0.0
        g_fenced = 0
        last_print = time.perf_counter()
        with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
            seed = a.seed0
            while ex < a.exchanges:
                A, ta = run(ev, sp, S, seed, a.gens, mode_a)
                if a.one_launch:
                    assert A["launches"] == 1, A["launches"]
                B, tb = run(ev, sp, S, seed, a.gens, "two_launch")
                assert A["status"]["iteration"] == a.gens
                bad = same(A, B)
                if n_seeds % a.fenced_every == 0:
                    C, tc = run(ev, sp, S, seed, a.gens, "fenced")
                    bad += ["fenced:" + k for k in same(C, B)]
                    t_fenced += tc
                    g_fenced += a.gens
                    n_fenced += 1
                if bad:
                    bad_seeds += 1
                    print("MISMATCH shape %s seed %d: %s" % (name, seed, bad), flush=True)
                t_fast += ta
                t_two += tb
                ex += a.gens
                n_seeds += 1
                seed += 1
                if time.perf_counter() - last_print > 30:
                    last_print = time.perf_counter()
                    print("  ... shape %s: %d exchanges, %d mismatching seeds, %.0f s" % (
                        name, ex, bad_seeds, time.perf_counter() - t_start), flush=True)
        print("shape S=%d N=%d P=%d (%d workgroups): %d exchanges (%d posts) over %d seeds x %d generations, "
              "A (%s) vs TWO_LAUNCH mismatching seeds: %d; per generation A %.2f us, TWO_LAUNCH %.2f us, "
              "FENCED %.2f us (%d seeds)" % (S, N, P, wgs, ex, ex * wgs, n_seeds, a.gens, mode_a, bad_seeds,
                                            t_fast / ex * 1e6, t_two / ex * 1e6,
                                            (t_fenced / g_fenced * 1e6) if g_fenced else float("nan"), n_fenced),
              flush=True)
        total_ex += ex
        total_posts += ex * wgs
        total_bad += bad_seeds
    print("TOTAL: %d exchanges, %d posts, %d mismatching seeds, %.0f s" % (
        total_ex, total_posts, total_bad, time.perf_counter() - t_start), flush=True)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())

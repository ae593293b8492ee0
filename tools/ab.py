#!/usr/bin/env python3
"""A/B timing of several builds of libnmrfit_amd in ONE process on ONE device, interleaved
rounds, median and min reported (perf deltas from separate runs or boxes are not comparable:
devices differ by several per cent).  Usage:
    NMRFIT_LIBNAME=libab_w3.so nmrfit_amd/csrc/build.sh -DNMRFIT_BATCH_MIN_WAVES=3      (any -D knob the sources still carry;
                                                     the round 1-4 tuning knobs became constants in round 5: DESIGN.md 7a)
    python tools/ab.py nmrfit_amd/lib/libnmrfit_amd.so nmrfit_amd/lib/libab_w3.so [--variant 0] [--workload C3]
A spec may carry its own variant after a colon (same build, two kernel variants):
    python tools/ab.py nmrfit_amd/lib/libnmrfit_amd.so:0 nmrfit_amd/lib/libnmrfit_amd.so:7
Also prints the largest relative difference of f from the first spec's values.
"""
import argparse, ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import _cabi, synth
from nmrfit_amd.equations import Evaluator


def load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in _cabi.ALL_SIGNATURES.items():
        if not hasattr(L, name): continue      # (an older build of the library: entry points added since)
        fn = getattr(L, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
    L.nmrfit_last_error.argtypes = []; L.nmrfit_last_error.restype = ctypes.c_char_p
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--shape", default=None, help="S,N,P instead of a named workload")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.workload]
    if a.shape:
        import collections
        S_, N_, P_ = (int(t) for t in a.shape.split(","))
        cfg = collections.namedtuple("Shape", "S N P")(S_, N_, P_)
    sp = synth.make_spectrum(cfg.N, cfg.P, seed=1)
    X = synth.make_swarm(sp["lower"], sp["upper"], cfg.S, seed=2, x_true=sp["x_true"])
    import numpy as np
    paths = [p.rsplit(":", 1)[0] if p.rsplit(":", 1)[-1].isdigit() else p for p in a.libs]
    variants = [int(p.rsplit(":", 1)[1]) if p.rsplit(":", 1)[-1].isdigit() else a.variant for p in a.libs]
    handles = {}
    libs = [handles.setdefault(q, load(q)) if q not in handles else handles[q] for q in paths]
    times = {p: [] for p in a.libs}
    values = {}
    for r in range(a.rounds):
        for p, L, var in zip(a.libs, libs, variants):
            _cabi._LIB = L
            with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
                ev.set_variant(var)
                dX = ev.dev_alloc(X.nbytes); df = ev.dev_alloc(8 * cfg.S)
                ev.upload(dX, X)
                for _ in range(3):
                    ev.objective_batch_dev(cfg.S, cfg.P, dX, df)
                ev.synchronize()
                ev.timer_begin()
                for _ in range(a.reps):
                    ev.objective_batch_dev(cfg.S, cfg.P, dX, df)
                times[p].append(ev.timer_end() / a.reps)
                values[p] = ev.download(df, (cfg.S,))
                ev.dev_free(dX); ev.dev_free(df)
    base = statistics.median(times[a.libs[0]])
    for p in a.libs:
        t = times[p]
        diff = float(np.max(np.abs(values[p] - values[a.libs[0]]) / np.maximum(np.abs(values[a.libs[0]]), 1e-6)))
        print("%-44s median %.4f ms  min %.4f  max %.4f  (%.3fx of first)  max rel diff of f %.2e" % (
            os.path.basename(p), statistics.median(t), min(t), max(t), statistics.median(t) / base, diff))


if __name__ == "__main__":
    main()

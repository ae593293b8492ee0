#!/usr/bin/env python3
"""Per-generation A/B of several builds of libnmrfit_amd on the swarm loop (nmrfit_pso_run, stopping
tests off), interleaved in ONE process on ONE device, several rounds, all values printed:
    python tools/gen_ab.py nmrfit_amd/lib/libab_base.so nmrfit_amd/lib/libnmrfit_amd.so [--shapes "204,4096,6;1024,4096,6"]
Also checks that the builds end on the same global best, bit for bit."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrfit_amd import _cabi, synth, pso
from nmrfit_amd.equations import Evaluator


def load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in _cabi.SIGNATURES.items():
        fn = getattr(L, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
    L.nmrfit_last_error.argtypes = []; L.nmrfit_last_error.restype = ctypes.c_char_p
    return L


ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--shapes", default="50,4096,6;204,4096,6;512,4096,6;1024,4096,6;204,16384,12;4096,65536,24")
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
libs = [(os.path.basename(p), load(p)) for p in a.libs]
for shape in a.shapes.split(";"):
    S, N, P = (int(t) for t in shape.split(","))
    sp = synth.make_spectrum(N, P, seed=1)
    gens = 1500 if S * N * P < 1e9 else 150
    res = {n: [] for n, _ in libs}
    fgs = {}
    for rep in range(a.rounds):
        for name, L in libs:
            _cabi._LIB = L
            with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
                sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
                sw.run(max(20, gens // 10), check_every=1000)
                t0 = time.perf_counter()
                sw.run(gens, check_every=1000)
                res[name].append((time.perf_counter() - t0) / gens * 1e6)
                fgs[name] = sw.best()[1]
                sw.close()
    base = min(res[libs[0][0]])
    for name, _ in libs:
        print("S=%5d N=%6d P=%3d  %-28s %s us  (min %.2f, %.3fx of first)  best f %s" % (
            S, N, P, name, " / ".join("%.2f" % v for v in res[name]), min(res[name]), min(res[name]) / base,
            float(fgs[name]).hex()), flush=True)

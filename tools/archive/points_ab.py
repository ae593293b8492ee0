#!/usr/bin/env python3
"""Small-swarm experiment: does a build with fewer grid points per lane per
chunk (NMRFIT_POINTS = 4 or 2: more, shorter waves) shorten a generation of a small swarm?

    NMRFIT_LIBNAME=libab_p4.so nmrfit_amd/csrc/build.sh -DNMRFIT_POINTS=4
    NMRFIT_LIBNAME=libab_p2.so nmrfit_amd/csrc/build.sh -DNMRFIT_POINTS=2 -DNMRFIT_BATCHINV=2 -DNMRFIT_INTERLEAVE=2
    python tools/points_ab.py

Per-generation wall time of nmrfit_pso_run (stopping tests off), every (build, segments per
particle) pair of a shape interleaved in ONE process on ONE device, three rounds, all values
printed.  Segments are forced with NMRFIT_TARGET_WAVES (read at context creation); "auto" is the
host heuristic of that build."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrfit_amd import _cabi, synth, pso
from nmrfit_amd.equations import Evaluator

LIBDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmrfit_amd", "lib")
BUILDS = [("points=8", "libnmrfit_amd.so"), ("points=4", "libab_p4.so"), ("points=2", "libab_p2.so")]


def load(path):
    L = ctypes.CDLL(path)
    for name, argtypes in _cabi.SIGNATURES.items():
        fn = getattr(L, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
    L.nmrfit_last_error.argtypes = []; L.nmrfit_last_error.restype = ctypes.c_char_p
    return L


libs = [(n, load(os.path.join(LIBDIR, f))) for n, f in BUILDS if os.path.exists(os.path.join(LIBDIR, f))]
SHAPES = [(50, 4096, 6, (0, 8, 16)), (204, 4096, 6, (0, 4, 8, 16)), (512, 4096, 6, (0, 4, 8)),
          (1024, 4096, 6, (0, 2, 4, 8)), (204, 16384, 12, (0, 8, 16, 32))]
for (S, N, P, segs) in SHAPES:
    sp = synth.make_spectrum(N, P, seed=1)
    res = {}
    for rep in range(3):
        for name, L in libs:
            _cabi._LIB = L
            for nseg in segs:
                if nseg:
                    os.environ["NMRFIT_TARGET_WAVES"] = str(S * nseg)
                else:
                    os.environ.pop("NMRFIT_TARGET_WAVES", None)
                with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
                    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
                    sw.run(100, check_every=100)
                    t0 = time.perf_counter()
                    sw.run(1500, check_every=500)
                    dt = (time.perf_counter() - t0) / 1500 * 1e6
                    geom = ev.last_launch()["segments"]
                    fg = sw.status()["fg"]
                    sw.close()
                res.setdefault((name, "auto" if nseg == 0 else nseg, geom), []).append((dt, fg))
    print("S=%d N=%d P=%d" % (S, N, P))
    for (name, req, geom), vals in res.items():
        print("   %-9s segments %4s -> %2d : %s us   fg=%.12g" % (name, req, geom, " / ".join("%.2f" % v[0] for v in vals),
                                                               vals[0][1]))
    sys.stdout.flush()
os.environ.pop("NMRFIT_TARGET_WAVES", None)

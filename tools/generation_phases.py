#!/usr/bin/env python3
"""Where a one-launch swarm generation (personal bests in the objective launch, fold deferred into the next launch's
prologue) spends its time: shader-clock stamps of wave 0 of every workgroup at the phases of objective_kernel (a -DNMRFIT_DIAG_STAMPS build, nmrfit_diag_read_stamps).
    NMRFIT_LIBNAME=libab_stamps.so nmrfit_amd/csrc/build.sh -DNMRFIT_DIAG_STAMPS=1
    python tools/generation_phases.py nmrfit_amd/lib/libab_stamps.so [S N P]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import _cabi, synth, pso
from nmrfit_amd.equations import Evaluator

L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, argtypes in _cabi.ALL_SIGNATURES.items():
    fn = getattr(L, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
L.nmrfit_last_error.argtypes = []; L.nmrfit_last_error.restype = ctypes.c_char_p
L.nmrfit_diag_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
_cabi._LIB = L
S, N, P = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (204, 4096, 6)
# stamps (objective.hip, phase_stamp): 0 entry, 10 argmin partials in LDS, 11 folded, 1 position update done, 2 per-peak
# constants staged, 3 chunk loop done, 4 f known, 5 personal best on its way to memory
ORDER = [0, 10, 11, 1, 2, 3, 4, 5]
NAMES = ["entry", "argmin", "fold", "update", "staged", "chunks", "f known", "pbest out"]
sp = synth.make_spectrum(N, P, seed=1)
with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
    sw.run(200, check_every=1000)
    rows = []
    for rep in range(20):
        sw.step()
        ev.prof_enable(1)
        sw.step()                                  # (the second of two: its prologue folds the first)
        kms, _, mhz = ev.prof_read()
        nwg = min(S, 1024)
        buf = np.zeros((nwg, 16), dtype=np.uint64)
        assert L.nmrfit_diag_read_stamps(ev.handle, buf.ctypes.data_as(ctypes.c_void_p), nwg) == 0
        ev.prof_enable(0)
        t = buf.astype(np.int64)[:, ORDER]
        # (differences within a workgroup only: s_memtime counts per XCD)
        rows.append((kms[0] * 1e3, mhz, np.median(np.diff(t, axis=1), axis=0), np.median(t[:, -1] - t[:, 0])))
    launches = sw.last_launches()
    sw.close()
mhz = np.median([r[1] for r in rows])
print("S=%d N=%d P=%d: kernel %.2f us (HIP events, median of 20), %d launch per generation, shader clock %.0f MHz" % (
    S, N, P, np.median([r[0] for r in rows]), launches, mhz))
med = np.median([r[2] for r in rows], axis=0) / mhz
for i in range(len(ORDER) - 1):
    print("  %-10s -> %-10s %6.2f us  (median over workgroups)" % (NAMES[i], NAMES[i + 1], med[i]))
print("  entry -> personal best out %.2f us (median over workgroups; a build without the stamps is faster)" % (
    np.median([r[3] for r in rows]) / mhz))

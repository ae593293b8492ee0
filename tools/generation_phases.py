#!/usr/bin/env python3
"""Where a one-launch swarm generation spends its time: shader-clock stamps of wave 0 of every workgroup at the
phases of objective_kernel (a -DNMRFIT_DIAG_STAMPS build, nmrfit_diag_read_stamps).
    NMRFIT_LIBNAME=libab_stamps.so nmrfit_amd/csrc/build.sh -DNMRFIT_DIAG_STAMPS=1
    python tools/generation_phases.py nmrfit_amd/lib/libab_stamps.so [S N P]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrfit_amd import _cabi, synth, pso
from nmrfit_amd.equations import Evaluator

L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, argtypes in _cabi.SIGNATURES.items():
    fn = getattr(L, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
L.nmrfit_last_error.argtypes = []; L.nmrfit_last_error.restype = ctypes.c_char_p
L.nmrfit_diag_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
_cabi._LIB = L
S, N, P = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (204, 4096, 6)
NAMES = ["entry", "update", "staged", "chunks", "f known", "pbest stored", "ticket", "argmin", "row", "fold"]
sp = synth.make_spectrum(N, P, seed=1)
with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
    sw.run(200, check_every=100)
    rows = []
    for rep in range(20):
        ev.prof_enable(1)
        sw.step()
        kms, _, mhz = ev.prof_read()
        buf = np.zeros((S, 16), dtype=np.uint64)
        assert L.nmrfit_diag_read_stamps(ev.handle, buf.ctypes.data_as(ctypes.c_void_p), S) == 0
        ev.prof_enable(0)
        t = buf.astype(np.int64)
        last = int(np.argmax(t[:, 9]))            # the workgroup that finished (only one writes stamp 9 afresh)
        d_all = np.diff(t[:, :7], axis=1)          # phases every workgroup goes through
        fin = np.diff(t[last, 6:10])
        rows.append((kms[0] * 1e3, mhz, np.median(d_all, axis=0), fin, (t[:, 6].max() - t[:, 0].min()), t[last, 9] - t[:, 0].min()))
    sw.close()
mhz = np.median([r[1] for r in rows])
print("S=%d N=%d P=%d: kernel %.2f us (HIP events, median of 20), shader clock %.0f MHz" % (S, N, P, np.median([r[0] for r in rows]), mhz))
med = np.median([r[2] for r in rows], axis=0) / mhz
for i in range(6):
    print("  %-12s -> %-12s %6.2f us  (median over workgroups)" % (NAMES[i], NAMES[i + 1], med[i]))
fin = np.median([r[3] for r in rows], axis=0) / mhz
for i in range(3):
    print("  %-12s -> %-12s %6.2f us  (the finishing workgroup)" % (NAMES[6 + i], NAMES[7 + i], fin[i]))
print("  first entry -> last ticket %.2f us, first entry -> fold written %.2f us" % (
    np.median([r[4] for r in rows]) / mhz, np.median([r[5] for r in rows]) / mhz))

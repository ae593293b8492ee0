#!/usr/bin/env python3
"""A/B of residual_batch_dev (rows written to HBM) between library builds, C3 size by default:
   python tools/residual_ab.py libA.so libB.so [S N P]"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrfit_amd import _cabi, synth
from nmrfit_amd.equations import Evaluator
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab import load

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
S, N, P = nums if len(nums) == 3 else (2048, 65536, 24)
sp = synth.make_spectrum(N, P, seed=1)
X = synth.make_swarm(sp["lower"], sp["upper"], S, seed=2)
times = {p: [] for p in libs}
for r in range(5):
    for p in libs:
        _cabi._LIB = load(p)
        with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
            dX = ev.dev_alloc(X.nbytes); df = ev.dev_alloc(8 * S); dR = ev.dev_alloc(8 * S * N)
            ev.upload(dX, X)
            for _ in range(2):
                ev.residual_batch_dev(S, P, dX, dR, df)
            ev.synchronize(); ev.timer_begin()
            for _ in range(5):
                ev.residual_batch_dev(S, P, dX, dR, df)
            times[p].append(ev.timer_end() / 5)
            for d in (dX, df, dR):
                ev.dev_free(d)
for p in libs:
    t = times[p]
    print("%-30s median %.4f ms  min %.4f  (%.1f GB/s of rows)" % (os.path.basename(p), statistics.median(t), min(t), S * N * 8 / statistics.median(t) / 1e6))

#!/usr/bin/env python3
"""Timeline of nmrfit_amd.fit_many's three stages (prepare | run | read back) for 200 default jobs with pyswarm's rule:
when each batch is created, run and collected, on which thread.  python tools/pipeline_trace.py [generate 0/1]"""
import contextlib, io, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmrfit_amd
from nmrfit_amd import core, synth, batch

gen = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(8)]
def jobs(n):
    return [dict(data=synth.SynthData(specs[k % 8]["w"], specs[k % 8]["u"], specs[k % 8]["v"], specs[k % 8]["peaks"]),
                 lower=list(specs[k % 8]["lower"]), upper=list(specs[k % 8]["upper"]), options={"seed": 7 + k}) for k in range(n)]
log = []
t0 = [0.0]
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        ts = time.perf_counter()
        r = f(*a, **k)
        log.append((tag, threading.current_thread().name[-8:], (ts - t0[0]) * 1e3, (time.perf_counter() - t0[0]) * 1e3))
        return r
    setattr(obj, name, g)
wrap(core, "_batch_create", "create")
wrap(core, "_batch_collect", "collect")
wrap(batch.FitBatch, "run", "run")
wrap(batch.FitBatch, "generate", "  generate")
wrap(batch.FitBatch, "close", "  close")
with contextlib.redirect_stdout(io.StringIO()):
    nmrfit_amd.fit_many(jobs(8), generate=gen)
    log.clear()
    for rep in range(2):
        log.append(("---- rep %d" % rep, "", 0, 0))
        t0[0] = time.perf_counter()
        nmrfit_amd.fit_many(jobs(200), generate=gen)
        log.append(("total", "", 0, (time.perf_counter() - t0[0]) * 1e3))
for tag, th, a, b in log:
    print("%-12s %-9s %8.1f -> %8.1f ms  (%.1f)" % (tag, th, a, b, b - a))

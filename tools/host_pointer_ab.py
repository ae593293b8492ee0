#!/usr/bin/env python3
"""nmrfit_objective_batch with host pointers at C3 size (4096 x 65536 x 24): the pipelined upload (slices through pinned
memory, kernels overlapping the copies) against the plain call (NMRFIT_NO_HOST_PIPELINE=1), and the resident launch.
    python tools/host_pointer_ab.py            (runs itself twice, the second time with the knob)"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from nmrfit_amd import synth
    from nmrfit_amd.equations import Evaluator
    for name in ("C3", "C2"):
        sp, X = synth.make_workload(name)
        with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
            S, D = X.shape
            P = (D - 4) // 3
            dX, df = ev.dev_alloc(X.nbytes), ev.dev_alloc(S * 8)
            ev.upload(dX, X)
            for _ in range(20):
                ev.objective_batch_dev(S, P, dX, df)
            ev.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ev.objective_batch_dev(S, P, dX, df)
            ev.synchronize()
            res = (time.perf_counter() - t0) / 20 * 1e3
            f_dev = ev.download(df, (S,))
            f = ev.objective_batch(X)
            ev.objective_batch(X)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                f = ev.objective_batch(X)
                ts.append((time.perf_counter() - t0) * 1e3)
            print("%s %-9s resident %.4f ms, host-pointer call mean %.4f / min %.4f ms (+%.3f), f identical to the resident launch: %s"
                  % (name, "plain" if os.environ.get("NMRFIT_NO_HOST_PIPELINE") else "pipelined", res, np.mean(ts), np.min(ts),
                     np.mean(ts) - res, bool(np.array_equal(f, f_dev))), flush=True)
else:
    for env in ({}, {"NMRFIT_NO_HOST_PIPELINE": "1"}, {}, {"NMRFIT_NO_HOST_PIPELINE": "1"}):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)

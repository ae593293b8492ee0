"""
TEST / BENCH INFRASTRUCTURE ONLY (see oracle/nmrfit_oracle.py's header).

The reference's one parallel mode, restated for bench.py's cpu_baseline leg: pyswarm hands the
particles of a generation to `multiprocessing.Pool(processes).map(objective, ...)`
(reference utils.py:176-182 passes `processes=` through to pyswarm).  Workers are spawned, not
forked, so they never inherit the parent's GPU state.
"""
import multiprocessing as mp
import time

import numpy as np

from . import nmrfit_oracle as onp

_ARGS = None


def _init(w, u, v, weights):
    global _ARGS
    _ARGS = (w, u, v, weights)


def _one(x):
    return onp.objective(x, *_ARGS)


def timed_map(X, w, u, v, weights, processes, budget_s):
    """Evaluate rows of X through Pool.map in rounds of 4*processes until the budget is used.
    Returns (particles evaluated, seconds, values); pool start-up is not timed."""
    ctx = mp.get_context("spawn")
    with ctx.Pool(processes, initializer=_init, initargs=(w, u, v, weights)) as pool:
        pool.map(_one, list(X[:processes]))                      # warm the workers
        done, out = 0, []
        t0 = time.perf_counter()
        while done < X.shape[0]:
            rows = list(X[done:done + 4 * processes])
            out.extend(pool.map(_one, rows))
            done += len(rows)
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    return done, dt, np.asarray(out)

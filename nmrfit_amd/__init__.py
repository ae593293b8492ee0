"""
nmrfit_amd -- MI355X (gfx950) evaluator for nmrfit's objective function and the swarm loop
around it, behind the reference's own API for that path:

    nmrfit_amd.fit(data, lower, upper, ...) -> FitUtility        (nmrfit/core.py:64)
    nmrfit_amd.fit_many([(data, lower, upper), ...])             several spectra: one DEVICE BATCH, a launch per generation
    nmrfit_amd.batch.FitBatch(spectra, lowers, uppers, ...)      the batch itself (nmrfit_batch_* of the library)
    nmrfit_amd.equations.objective(x, w, u, v, weights)          (nmrfit/equations.py:152)
    nmrfit_amd.equations.Evaluator(...).objective_batch(X)       one launch per swarm generation
    nmrfit_amd.pso.DeviceSwarm / pso.pso                         (replaces pyswarm.pso)
    nmrfit_amd.Data(w, u, v)                                     (nmrfit/containers.py:8, scripted use)

Everything that evaluates the objective goes through libnmrfit_amd.so (include/nmrfit_amd.h);
there is no CPU fallback.  The once-per-dataset helpers either side of it (phase estimate, peak
picking, bounds, weights) are host code as in the reference.  Instrument I/O (nmrfit.load), the
matplotlib click selectors and plotting are out of scope.
"""
from .core import fit, fit_many  # noqa: F401
from . import batch, containers, equations, peaks, proc_autophase, pso, synth, utils  # noqa: F401
from .containers import Data  # noqa: F401

__version__ = "0.1.0"

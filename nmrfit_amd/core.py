"""
``fit`` -- the reference's entry point (nmrfit/core.py:64-95), same signature and return
value, executed on the MI355X.  ``load`` (instrument file parsing through nmrglue,
core.py:9-61) is outside the hot-path scope: build a ``Data``-like object (attributes
w, u, v, peaks) with the reference package or ``nmrfit_amd.synth`` and pass it in.
"""
from . import utils


def fit(data, lower, upper, expon=0.5, dynamic_weighting=True, fit_im=False, processes=1, summary=True,
        options={}):
    """Perform a fit of NMR spectroscopy data (reference: nmrfit/core.py:64).

    data : object with ``w, u, v`` (ndarrays) and ``peaks`` (each with ``bounds``, ``height``)
    lower, upper : parameter box, 4 + 3P floats (nmrfit/containers.py:193-217)
    expon, dynamic_weighting : error weighting (nmrfit/utils.py:191-224)
    fit_im : False (real part only, the default); True = the reference's imaginary term exactly as
             nmrfit/equations.py:197-209 computes it (last peak's line only); "sum" = all peaks.
             The Kramers-Kronig partner is evaluated in closed form on the GPU
    processes : accepted for compatibility; the batched GPU launch replaces the process pool
    summary : print the fit summary table
    options : swarmsize, maxiter, omega, phip, phig (+ minstep, minfunc, seed, device,
              check_every, polish, variant, exchange="rccl" for one process per GPU)

    Returns the FitUtility holding ``params``, ``error``, ``weights``.
    """
    f = utils.FitUtility(data, lower, upper, expon, dynamic_weighting, fit_im, processes, summary, options)
    f.fit()
    return f


def fit_many(jobs, threads=4, **kwargs):
    """Fit several spectra at once: ``jobs`` is a sequence of ``(data, lower, upper)`` triples (or dicts of ``fit``'s
    arguments), each fitted by ``fit`` with the same keyword arguments, on ``threads`` host threads.  Not in the
    reference (its users loop over ``nmrfit.fit``, nmrfit/core.py:64); here a 204-particle swarm fills a fraction
    of an MI355X, every fit has its own context and HIP stream and the library releases the GIL inside its calls,
    so independent fits overlap on the device: about 2.2x the fits per second of a plain loop with four threads
    (tools/concurrent_fits.py).  Results come back in the order of ``jobs`` and are the ones the plain loop gives
    (the swarm's random numbers depend on ``options['seed']`` only).  ``summary`` defaults to False here."""
    from concurrent.futures import ThreadPoolExecutor
    kwargs.setdefault("summary", False)

    def one(job):
        if isinstance(job, dict):
            return fit(**dict(kwargs, **job))
        data, lower, upper = job
        return fit(data, lower, upper, **kwargs)
    jobs = list(jobs)
    if threads <= 1 or len(jobs) <= 1:
        return [one(j) for j in jobs]
    with ThreadPoolExecutor(max_workers=int(threads)) as pool:
        return list(pool.map(one, jobs))

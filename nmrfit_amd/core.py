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

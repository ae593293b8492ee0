"""
Host mirror of the reference's phase module (nmrfit/proc_autophase.py): ``ps2`` (:9-36),
``ps`` (:39-68), ``autops`` (:71-104), ``approximate_phase`` (:107-139) and the two phase
scores (ACME :142-187, peak minima :190-219).  Host code, as in the reference: these run once
per dataset (Data.shift_phase, FitUtility.generate_result); inside the objective the rotation
is fused into the GPU kernel.  ``manual_ps`` (:222-300, a matplotlib slider GUI) is out of scope.

Units follow the reference: ``ps2`` takes radians, ``ps`` and the scores take DEGREES, and
``approximate_phase`` converts the optimiser's degrees to the radians ``Data`` stores.
"""
import numpy as np
import scipy.optimize


def _ramp(size, p0, p1):
    """exp(i (p0 + p1 j / size)), j the array index: the order of operations of the reference."""
    return np.exp(1.0j * (p0 + (p1 * np.arange(size) / size)))


def ps2(u, v, p0=0.0, p1=0.0, inv=False):
    """(u + i v) * exp(+-i (p0 + p1*j/N)), radians, j the array index; returns (real, imag)."""
    data = np.asarray(u) + 1j * np.asarray(v)
    apod = _ramp(data.shape[-1], p0, p1).astype(data.dtype)
    if inv:
        apod = 1 / apod
    data = apod * data
    return data.real, data.imag


def ps(data, p0=0.0, p1=0.0, inv=False):
    """Linear phase correction of a complex spectrum, p0 and p1 in degrees (proc_autophase.py:39-68)."""
    data = np.asarray(data)
    apod = _ramp(data.shape[-1], p0 * np.pi / 180.0, p1 * np.pi / 180.0).astype(data.dtype)
    if inv:
        apod = 1 / apod
    return apod * data


def _ps_acme_score(ph, data):
    """ACME score (Chen Li et al., J. Magn. Reson. 158 (2002) 164-168) as the reference evaluates
    it (proc_autophase.py:142-187): entropy of the normalised absolute first difference of the
    real part, plus 1000 x the sum of squares of its negative excursions."""
    real = np.real(ps(data, p0=ph[0], p1=ph[1]))
    slope = np.abs((real[1:] - real[:-1]) / 2.0)
    prob = slope / np.sum(slope)
    prob[prob == 0] = 1                       # 0 log 0 := 0
    entropy = np.sum(-prob * np.log(prob))
    neg = real - np.abs(real)                 # 2 x the negative part
    penalty = np.sum((neg / 2) ** 2) if np.sum(neg) < 0 else 0.0
    return entropy + 1000 * penalty


def _ps_peak_minima_score(ph, data):
    """|min left - min right| within 100 points of the tallest point (proc_autophase.py:190-219)."""
    real = np.real(ps(data, p0=ph[0], p1=ph[1]))
    i = np.argmax(real)
    return np.abs(np.min(real[i - 100:i]) - np.min(real[i:i + 100]))


_SCORES = {"acme": _ps_acme_score, "peak_minima": _ps_peak_minima_score}


def _optimise(data, fn, p0, p1):
    if not callable(fn):
        fn = _SCORES[fn]
    return scipy.optimize.fmin(fn, x0=[p0, p1], args=(data,), disp=False)


def autops(data, fn, p0=0.0, p1=0.0):
    """Nelder-Mead over (p0, p1) in degrees on the chosen score; returns the phased spectrum
    (proc_autophase.py:71-104)."""
    opt = _optimise(data, fn, p0, p1)
    return ps(data, p0=opt[0], p1=opt[1])


def approximate_phase(data, fn, p0=0.0, p1=0.0):
    """The same optimisation, returning (p0, p1) in RADIANS (proc_autophase.py:107-139) -- what
    Data.shift_phase('auto') stores."""
    opt = _optimise(data, fn, p0, p1)
    return opt[0] * np.pi / 180, opt[1] * np.pi / 180

"""
``ps2`` with the reference's signature (nmrfit/proc_autophase.py:9-36).  Host code, as in the
reference: it runs once per dataset (Data.shift_phase, FitUtility.generate_result); inside
the objective the rotation is fused into the GPU kernel.  The automatic phase estimators of
the reference module (autops, ACME/peak-minima scores, manual_ps GUI) are out of scope.
"""
import numpy as np


def ps2(u, v, p0=0.0, p1=0.0, inv=False):
    """(u + i v) * exp(+-i (p0 + p1*j/N)), radians, j the array index; returns (real, imag)."""
    data = np.asarray(u) + 1j * np.asarray(v)
    size = data.shape[-1]
    apod = np.exp(1.0j * (p0 + (p1 * np.arange(size) / size))).astype(data.dtype)
    if inv:
        apod = 1 / apod
    data = apod * data
    return data.real, data.imag

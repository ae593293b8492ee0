"""
CPU tier: the oracle's restatement of the Kramers-Kronig path (scipy quad per grid point,
nmrfit/equations.py:9-80) and of the fit_im=True objective against the golden vectors the
reference produced -- and the closed form the GPU uses (Lorentzian dispersion + Dawson's
integral, here via scipy.special.dawsn) against the same vectors, to show the two agree to
the quadrature's tolerance.
"""
import os

import numpy as np
import pytest

from oracle import nmrfit_oracle as onp
from nmrfit_amd import synth


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "kramers_kronig.npz"))


def test_oracle_kk_matches_reference(g):
    x = g["x"]
    r, yoff = x[2], x[3]
    im0 = onp.kk_relation_vectorized(g["w"], r, yoff, x[4], x[5], x[6])
    np.testing.assert_allclose(im0, g["imag_contribs"][0], rtol=0, atol=1e-12 * np.abs(g["imag_contribs"]).max())


def test_oracle_fit_im_objective_matches_reference(g):
    f = onp.objective_fit_im(g["X"][0], g["w"], g["u"], g["v"], g["weights"])
    assert f == pytest.approx(g["f_fit_im"][0], rel=1e-12)


def test_closed_form_agrees_with_the_reference_quadrature(g):
    """scipy quad's default tolerance is 1.49e-8; the closed form lands well inside it."""
    x = g["x"]
    r = x[2]
    scale = np.abs(g["imag_contribs"]).max()
    for k in range(3):
        cf = synth._dispersion(g["w"], r, x[4 + 3 * k], x[5 + 3 * k], x[6 + 3 * k])
        np.testing.assert_allclose(cf, g["imag_contribs"][k], rtol=0, atol=1e-9 * scale)
    cf = synth._dispersion(g["w_up"], r, x[4], x[5], x[6])
    np.testing.assert_allclose(cf, g["imag_up"], rtol=0, atol=1e-9 * scale)
    for key, args in (("imag_wide", g["wide_args"]), ("imag_needle", g["needle_args"])):
        cf = synth._dispersion(g["w2"], args[0], args[2], args[3], args[4])
        np.testing.assert_allclose(cf, g[key], rtol=0, atol=2e-8 * np.abs(g[key]).max())

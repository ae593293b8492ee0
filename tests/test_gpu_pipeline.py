"""
GPU tier: the scripted workflow of the reference's README from arrays to areas, every step on
this package -- Data -> shift_phase('auto') -> select_peaks('auto') -> generate_solution_bounds
-> fit (device-resident swarm on the HIP objective) -> generate_result.
"""
import numpy as np
import pytest

import nmrfit_amd
from nmrfit_amd import _cabi, synth

pytestmark = pytest.mark.gpu


def test_arrays_to_areas():
    assert _cabi.device_count() >= 1
    sp = synth.make_spectrum(8192, 4, seed=33, physical=True)
    xt = sp["x_true"]
    data = nmrfit_amd.Data(sp["w"], sp["u"], sp["v"])
    data.shift_phase(method="auto")
    # ACME lands near the generating phase (evaluated mid-spectrum, where p0 and p1 trade off)
    mid = lambda p0, p1: p0 + 0.5 * p1
    assert abs(mid(data.p0, data.p1) - mid(xt[0], xt[1])) < 0.15
    data.select_peaks(method="auto", thresh=0.1, window=0.02)
    assert len(data.peaks) == 4
    lower, upper = data.generate_solution_bounds()
    res = nmrfit_amd.fit(data, lower, upper, summary=False,
                         options={"swarmsize": 408, "maxiter": 1500, "seed": 3, "minfunc": -1.0, "minstep": -1.0})
    assert res.params.shape == (16,)
    # the fit explains the spectrum to the noise level ...
    with nmrfit_amd.equations.Evaluator(sp["w"], sp["u"], sp["v"], res.weights) as ev:
        f_truth = ev.objective_batch(xt)[0]
    assert res.error < 3.0 * f_truth
    # ... and finds the lines where they are, with areas close to the generating ones
    np.testing.assert_allclose(res.params[5::3], xt[5::3], atol=1e-3)
    np.testing.assert_allclose(res.get_areas(), xt[6::3], rtol=0.2)
    assert abs(res.calculate_area_fraction() - data.approximate_area_fraction()) < 0.2
    res.generate_result(scale=2)
    assert res.w.shape == (16384,) and len(res.real_contribs) == 4 and len(res.imag_contribs) == 4
    # res.V is the fitted real line shape on the upsampled grid (utils.py:289-295); against the
    # data rotated by the fitted phase it leaves noise only
    np.testing.assert_allclose(np.sum(res.real_contribs, axis=0), res.V, rtol=1e-12, atol=1e-12)
    V_data, _ = nmrfit_amd.proc_autophase.ps2(sp["u"], sp["v"], res.params[0], res.params[1])
    V_on_fit_grid = np.interp(res.w, sp["w"], V_data)
    assert np.sqrt(np.mean((V_on_fit_grid - res.V) ** 2)) < 0.02 * np.abs(V_data).max()


def test_fit_with_the_farfield_variant():
    """options={"variant": "farfield"}: same swarm, same seed -> the same trajectory to rounding,
    since the far-field kernel returns the direct kernel's values to ~4e-16."""
    sp = synth.make_spectrum(4096, 6, seed=12)
    data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
    opts = {"swarmsize": 128, "maxiter": 40, "seed": 4, "minfunc": -1.0, "minstep": -1.0}
    a = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False, options=opts)
    b = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False, options=dict(opts, variant="farfield"))
    assert b.error == pytest.approx(a.error, rel=1e-6)
    with pytest.raises(ValueError):
        nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False, options=dict(opts, variant="nope"))

"""
CPU tier: the callers either side of the hot path -- phase estimators, the Data container and
the peak utilities -- against fixtures the reference itself produced (tests/golden/
data_container.npz, oracle/make_golden.py section 8), plus property tests for the automatic peak
picker, whose third-party baseline routine (peakutils) is absent and therefore restated unpinned.
"""
import os

import numpy as np
import pytest

from nmrfit_amd import containers, peaks, proc_autophase, synth, utils


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "data_container.npz"))


def test_ps_and_scores_match_reference(g):
    z = g["u"] + 1j * g["v"]
    np.testing.assert_array_equal(proc_autophase.ps(z, p0=33.0, p1=-71.5), g["ps_deg"])
    np.testing.assert_array_equal(proc_autophase.ps(z, p0=33.0, p1=-71.5, inv=True), g["ps_deg_inv"])
    for ph, acme, minima in zip(g["score_phases"], g["acme"], g["peak_minima"]):
        assert proc_autophase._ps_acme_score(ph, z) == pytest.approx(acme, rel=1e-13)
        assert proc_autophase._ps_peak_minima_score(ph, z) == pytest.approx(minima, rel=1e-12, abs=1e-15)


def test_phase_estimators_match_reference(g):
    z = g["u"] + 1j * g["v"]
    # same scipy Nelder-Mead on the same score: same iterates up to rounding
    np.testing.assert_allclose(proc_autophase.approximate_phase(z, "acme"), g["approx_acme"], rtol=1e-8)
    np.testing.assert_allclose(proc_autophase.approximate_phase(z, "peak_minima", p0=5.0, p1=-3.0),
                               g["approx_minima"], rtol=1e-8)
    np.testing.assert_allclose(proc_autophase.autops(z, "acme"), g["autops_acme"], rtol=0, atol=1e-9 * np.abs(z).max())
    # a user-supplied score function is accepted, as in the reference
    p = proc_autophase.approximate_phase(z, proc_autophase._ps_acme_score)
    np.testing.assert_allclose(p, g["approx_acme"], rtol=1e-8)


def test_data_container_matches_reference(g):
    d = containers.Data(g["w"].copy(), g["u"].copy(), g["v"].copy())
    np.testing.assert_array_equal(d.V, g["u"])
    d.shift_phase(method="manual", p0=0.2, p1=-0.1)
    np.testing.assert_array_equal(d.V, g["manual_V"])
    np.testing.assert_array_equal(d.I, g["manual_I"])
    d.shift_phase(method="auto")
    np.testing.assert_allclose([d.p0, d.p1], g["auto_p"], rtol=1e-8)
    np.testing.assert_allclose(d.V, g["auto_V"], rtol=0, atol=1e-8 * np.abs(g["auto_V"]).max())
    d.shift_phase(method="brute", step=np.pi / 90)
    np.testing.assert_array_equal([d.p0, d.p1], g["brute_p"])
    np.testing.assert_array_equal(d.V, g["brute_V"])
    d.select_bounds(low=3.2, high=3.8)
    np.testing.assert_array_equal(d.w, g["crop_w"])
    np.testing.assert_array_equal(d.u, g["crop_u"])
    np.testing.assert_array_equal(d.v, g["crop_v"])
    sp = synth.make_spectrum(2048, 3, seed=int(g["seed"]), physical=True)
    d.peaks = sp["peaks"]
    np.testing.assert_array_equal(d.approximate_areas(), g["areas"])
    assert d.approximate_area_fraction() == g["area_fraction"]
    with pytest.raises(ValueError, match="Method must be 'auto', 'brute', or 'manual'."):
        d.shift_phase(method="nope")
    with pytest.raises(ValueError, match="Number of peaks must be specified"):
        d.select_peaks(method="manual")
    with pytest.raises(ValueError, match="Method must be 'auto' or 'manual'."):
        d.select_peaks(method="nope")
    for gui in (lambda: d.select_bounds(), lambda: d.select_peaks(method="manual", n=2),
                lambda: d.shift_phase(method="manual", plot=True)):
        with pytest.raises(NotImplementedError):
            gui()


def test_peak_helpers_match_reference(g):
    pk = utils.Peaks()
    for h in (3.0, -0.4, 2.5, 0.3, 0.35):
        q = utils.Peak()
        q.height = h
        pk.append(q)
    assert pk.average_height() == g["avg_height"]
    main, sats = pk.split()
    np.testing.assert_array_equal([q.height for q in main], g["split_main"])
    np.testing.assert_array_equal([q.height for q in sats], g["split_sats"])
    np.testing.assert_array_equal(np.array(utils.find_peak(g["w"], g["manual_V"], 3.3, 3.7), dtype=float), g["find_peak"])
    assert utils.sample_noise(g["w"], g["manual_V"], 3.0, 3.05) == pytest.approx(float(g["sample_noise"]), rel=1e-12)
    rng_state = np.random.get_state()
    np.random.seed(5)
    a = utils.rnd_data(0.1, np.zeros(1000))
    np.random.set_state(rng_state)
    assert abs(a.std() - 0.1) < 0.01 and a.shape == (1000,)


def test_baseline_restatement_properties():
    """peakutils.baseline restated (parity unpinned): a constant fit of noise-free data with peaks
    converges to the floor of the data, a cubic recovers a cubic background under sparse peaks."""
    x = np.linspace(0, 1, 2000)
    lines = 5.0 * np.exp(-((x - 0.3) / 0.01) ** 2) + 3.0 * np.exp(-((x - 0.7) / 0.02) ** 2)
    b0 = peaks.baseline(lines + 0.25, 0)
    assert b0.shape == x.shape and np.ptp(b0) == 0.0
    assert 0.25 <= b0[0] < 0.30
    cubic = 1.0 + 0.5 * x - 2.0 * x ** 2 + 1.5 * x ** 3
    b3 = peaks.baseline(lines + cubic, 3)
    assert np.max(np.abs(b3 - cubic)) < 0.1          # the 1e-3 coefficient tolerance stops it early


def test_auto_peak_selector_recovers_synthetic_lines():
    sp = synth.make_spectrum(4096, 6, seed=1)
    xt = sp["x_true"]
    d = containers.Data(sp["w"], sp["u"], sp["v"])
    d.shift_phase(method="manual", p0=xt[0], p1=xt[1])
    d.select_peaks(method="auto", thresh=0.1, window=0.02)
    assert len(d.peaks) == 6 and len(d.roibounds) == 6
    for k, p in enumerate(d.peaks):
        width, loc, area = xt[4 + 3 * k:7 + 3 * k]
        assert abs(p.loc - loc) < 0.05 * width
        assert abs(p.width - width) < 0.03 * width
        assert p.bounds == [p.loc - 2 * p.width, p.loc + 2 * p.width]
        assert 0.7 * area < p.area < 1.0 * area          # +-2 FWHM misses the Lorentzian tails
        assert "Location" in repr(p) and "Area" in repr(p)
    # the box built from the picked peaks contains the generating parameters
    lower, upper = d.generate_solution_bounds()
    assert len(lower) == len(upper) == 22
    assert np.all(np.array(lower) <= xt) and np.all(xt <= np.array(upper))
    # with thresh = 0 (the reference default) noise maxima come through as well
    d.select_peaks(method="auto")
    assert len(d.peaks) > 6


def test_sliding_window_argrelmax_equals_scipy():
    """peaks.argrelmax is scipy.signal.argrelmax (utils.py:731) computed in O(n): identical
    indices on random data, plateaus (strictness) and windows longer than the array."""
    import scipy.signal
    rng = np.random.default_rng(0)
    for trial in range(400):
        n = int(rng.integers(1, 300))
        order = int(rng.integers(1, 64))
        x = rng.integers(0, 6, n).astype(float) if trial % 2 else rng.standard_normal(n)
        np.testing.assert_array_equal(peaks.argrelmax(x, order), scipy.signal.argrelmax(x, order=order)[0])
    with pytest.raises(ValueError):
        peaks.argrelmax(np.arange(5.0), 0)

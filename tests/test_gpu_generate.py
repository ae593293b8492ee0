"""
GPU tier: the post-fit reconstruction (csrc/result.hip) -- FitUtility.generate_result (nmrfit/utils.py:226-295) for one
fit (nmrfit_generate_result) and for every fit of a device batch in one launch (nmrfit_batch_contributions), and the
README pipeline fit -> generate_result -> calculate_area_fraction (README.md:64-72; utils.py:297-322) through
nmrfit_amd.fit_many(generate=...).

Bars: the batch's arrays are BIT-IDENTICAL to the lone call's for the same parameter vector; the lone call is held to the
numpy restatement of what the reference computes around its contributions (sequential sums, ps2 both ways) at 1e-14 of
the spectrum's scale, and -- in tests/test_gpu_kk.py::test_generate_result_matches_reference -- to the
reference-generated golden vectors (real 1e-13, imaginary at the quadrature's 1e-8).
"""
import numpy as np
import pytest

from nmrfit_amd import proc_autophase, synth, utils
from nmrfit_amd.batch import FitBatch
from nmrfit_amd.equations import Evaluator

pytestmark = pytest.mark.gpu

_ATTRS = ("u", "v", "V", "I", "w")


def _problems(K, N=4096, peaks=(6, 4, 9, 1, 7, 6, 12, 3), seed0=140):
    return [synth.make_spectrum(N, peaks[k % len(peaks)], seed=seed0 + k, physical=True) for k in range(K)]


def _lone_result(sp, x, scale):
    data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
    fu = utils.FitUtility(data, sp["lower"], sp["upper"], summary=False)
    fu.params = x
    fu.generate_result(scale)
    return fu, data


def _same_result(a, b, data_a, data_b, tag):
    for name in _ATTRS:
        np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg="%s %s" % (tag, name))
    assert len(a.real_contribs) == len(b.real_contribs) == len(a.imag_contribs)
    for k in range(len(a.real_contribs)):
        np.testing.assert_array_equal(a.real_contribs[k], b.real_contribs[k], err_msg="%s real %d" % (tag, k))
        np.testing.assert_array_equal(a.imag_contribs[k], b.imag_contribs[k], err_msg="%s imag %d" % (tag, k))
    np.testing.assert_array_equal(data_a.V, data_b.V, err_msg=tag)
    np.testing.assert_array_equal(data_a.I, data_b.I, err_msg=tag)
    assert data_a.p0 == data_b.p0 and data_a.p1 == data_b.p1


def test_lone_generate_result_against_the_numpy_restatement():
    """u, v, V, I and the rotated spectrum against numpy doing what the reference does with the contributions
    (utils.py:251, 276-284): the sums are the same additions in the same order (bit-identical), the rotations differ by
    the rounding of sin / cos."""
    sp = synth.make_spectrum(3000, 5, seed=5, physical=True)       # (ragged: 5 full chunks + 440 points)
    x = np.array(sp["x_true"])
    x[0], x[1] = 0.7, -1.3
    for scale in (1, 1.5):
        fu, data = _lone_result(sp, x, scale)
        n = len(sp["w"]) if scale == 1 else int(scale * len(sp["w"]))
        assert fu.w.shape == (n,) and len(fu.real_contribs) == 5
        V = np.zeros(n)
        I = np.zeros(n)
        for k in range(5):
            V = V + fu.real_contribs[k]
            I = I + fu.imag_contribs[k]
        np.testing.assert_array_equal(fu.V, V)
        np.testing.assert_array_equal(fu.I, I)
        u, v = proc_autophase.ps2(V, I, inv=True, p0=x[0], p1=x[1])
        top = max(np.abs(V).max(), np.abs(I).max())
        np.testing.assert_allclose(fu.u, u, rtol=0, atol=1e-14 * top)
        np.testing.assert_allclose(fu.v, v, rtol=0, atol=1e-14 * top)
        Vd, Id = proc_autophase.ps2(sp["u"], sp["v"], x[0], x[1])
        top = max(np.abs(sp["u"]).max(), np.abs(sp["v"]).max())
        np.testing.assert_allclose(data.V, Vd, rtol=0, atol=1e-14 * top)
        np.testing.assert_allclose(data.I, Id, rtol=0, atol=1e-14 * top)
        assert data.p0 == x[0] and data.p1 == x[1]
        # the contributions themselves: the per-peak entry point the reference-named shims use
        with Evaluator(sp["w"], sp["u"], sp["v"], np.ones(len(sp["w"]))) as ev:
            real, imag = ev.contributions(x, None if scale == 1 else fu.w)
        np.testing.assert_array_equal(np.stack(fu.real_contribs), real)
        np.testing.assert_array_equal(np.stack(fu.imag_contribs), imag)


def test_a_data_object_with_shift_phase_gets_it_called():
    """The reference calls data.shift_phase(method='manual', p0, p1) (utils.py:251): a data object that has the method
    (nmrfit.containers.Data) keeps that behaviour on the lone path."""
    sp = synth.make_spectrum(1024, 2, seed=8)
    calls = []

    class Data(synth.SynthData):
        def shift_phase(self, method='auto', p0=0.0, p1=0.0):
            calls.append((method, p0, p1))
            self.p0, self.p1 = p0, p1
            self.V, self.I = proc_autophase.ps2(self.u, self.v, p0, p1)
    data = Data(sp["w"], sp["u"], sp["v"], sp["peaks"])
    fu = utils.FitUtility(data, sp["lower"], sp["upper"], summary=False)
    fu.params = np.array(sp["x_true"])
    fu.generate_result()
    assert calls == [("manual", sp["x_true"][0], sp["x_true"][1])]
    np.testing.assert_array_equal(data.V, proc_autophase.ps2(sp["u"], sp["v"], sp["x_true"][0], sp["x_true"][1])[0])


@pytest.mark.parametrize("K,N,S", [(1, 4096, 204), (5, 4096, 64), (13, 3000, 40), (3, 16384, 32)])
@pytest.mark.parametrize("scale", [1, 1.5])
def test_batch_generate_equals_the_lone_generate_result_bit_for_bit(K, N, S, scale):
    """Every array of FitBatch.generate -- one launch for all K fits (two parts from 6 fits on), mixed peak counts, a
    ragged grid -- equals FitUtility.generate_result for the same best position."""
    problems = _problems(K, N)
    with FitBatch([(sp["w"], sp["u"], sp["v"], sp["weights"]) for sp in problems], [sp["lower"] for sp in problems],
                  [sp["upper"] for sp in problems], swarmsize=S, seeds=[11 + k for k in range(K)]) as fb:
        fb.run(30, 8)
        best = fb.best()
        res = fb.generate(scale)
        again = fb.generate(scale)          # (a second call: the scratch of the first one is gone, same answer)
    for k, (sp, (x, _)) in enumerate(zip(problems, best)):
        fu, data = _lone_result(sp, x, scale)
        r = res[k]
        assert (r["w"] is None) == (scale == 1)
        if scale != 1:
            np.testing.assert_array_equal(r["w"], fu.w)
        np.testing.assert_array_equal(r["real"], np.stack(fu.real_contribs), err_msg=str(k))
        np.testing.assert_array_equal(r["imag"], np.stack(fu.imag_contribs), err_msg=str(k))
        for name in ("V", "I", "u", "v"):
            np.testing.assert_array_equal(r[name], getattr(fu, name), err_msg="%d %s" % (k, name))
            np.testing.assert_array_equal(r[name], again[k][name])
        np.testing.assert_array_equal(r["data_V"], data.V)
        np.testing.assert_array_equal(r["data_I"], data.I)


def test_batch_generate_edge_shapes():
    """Spectra of 1, 64, 513 and 700 points in one batch, peak counts 0, 1, 2, 3 (a fit without peaks: V = I = 0, u = v = 0),
    own grids and upsampled ones (int(2.5 * 1) = 2 points): against the lone call, bit for bit."""
    lengths, peaks = [1, 64, 513, 700], [0, 1, 2, 3]
    problems = []
    for k, (n, p) in enumerate(zip(lengths, peaks)):
        sp = synth.make_spectrum(max(n, 8), p, seed=800 + k, physical=True)
        for name in ("w", "u", "v", "weights"):
            sp[name] = sp[name][:n]
        problems.append(sp)
    with FitBatch([(sp["w"], sp["u"], sp["v"], sp["weights"]) for sp in problems], [sp["lower"] for sp in problems],
                  [sp["upper"] for sp in problems], swarmsize=16, seeds=[5, 6, 7, 8]) as fb:
        fb.run(10, 5)
        best = fb.best()
        for scale in (1, 2.5):
            res = fb.generate(scale)
            for k, (sp, (x, _)) in enumerate(zip(problems, best)):
                fu, data = _lone_result(sp, x, scale)
                r = res[k]
                n = lengths[k] if scale == 1 else int(scale * lengths[k])
                assert r["real"].shape == (peaks[k], n) and r["V"].shape == (n,) and r["data_V"].shape == (lengths[k],)
                for name in ("V", "I", "u", "v"):
                    np.testing.assert_array_equal(r[name], getattr(fu, name), err_msg="%d %s %r" % (k, name, scale))
                if peaks[k]:
                    np.testing.assert_array_equal(r["real"], np.stack(fu.real_contribs))
                    np.testing.assert_array_equal(r["imag"], np.stack(fu.imag_contribs))
                else:
                    assert not fu.real_contribs and not np.any(r["V"]) and not np.any(r["u"])
                np.testing.assert_array_equal(r["data_V"], data.V)
                np.testing.assert_array_equal(r["data_I"], data.I)


def test_batch_generate_before_the_first_generation_is_a_state_error():
    from nmrfit_amd import _cabi
    problems = _problems(2, 1024)
    with FitBatch([(sp["w"], sp["u"], sp["v"], sp["weights"]) for sp in problems], [sp["lower"] for sp in problems],
                  [sp["upper"] for sp in problems], swarmsize=8, seeds=[1, 2]) as fb:
        with pytest.raises(_cabi.NmrfitError) as ei:
            fb.generate()
        assert ei.value.code == _cabi.E_STATE
        fb.run(0, 1)                      # generation 0 only
        assert len(fb.generate()) == 2


def test_fit_many_generate_equals_the_readme_loop(monkeypatch, capsys):
    """fit -> generate_result -> calculate_area_fraction per spectrum (README.md:64-72) against
    fit_many(jobs, generate=True): several batches (BATCH_JOBS = 4), a job that runs alone, in job order, bit for bit."""
    import nmrfit_amd
    from nmrfit_amd import core
    monkeypatch.setattr(core, "BATCH_JOBS", 4)
    lengths = [2048] * 5 + [3072] + [2048] * 3
    specs = [synth.make_spectrum(n, 2 + k % 4, seed=500 + k, physical=True) for k, n in enumerate(lengths)]

    def jobs():      # (fresh data objects per run: generate_result writes p0, p1, V, I into them)
        return [dict(data=synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), lower=list(sp["lower"]),
                     upper=list(sp["upper"]), options={"seed": 900 + k, "swarmsize": 48, "maxiter": 70})
                for k, sp in enumerate(specs)]
    for scale in (True, 2):
        loop = []
        for job in jobs():
            f = nmrfit_amd.fit(job["data"], job["lower"], job["upper"], summary=False, options=job["options"])
            f.generate_result(scale=1 if scale is True else scale)
            loop.append(f)
        capsys.readouterr()
        many = nmrfit_amd.fit_many(jobs(), generate=scale)
        assert capsys.readouterr().out.count("Stopping search:") == len(specs)
        for k, (a, b) in enumerate(zip(many, loop)):
            np.testing.assert_array_equal(a.params, b.params)
            assert a.error == b.error
            _same_result(a, b, a.data, b.data, "job %d scale %r" % (k, scale))
            assert a.calculate_area_fraction() == b.calculate_area_fraction()
            if scale is True:
                assert a.w is a.data.w
    # without generate nothing of it exists
    plain = nmrfit_amd.fit_many(jobs())
    assert not hasattr(plain[0], "real_contribs") and np.array_equal(plain[0].params, loop[0].params)


def test_reconstruction_entry_points_validate_their_arguments():
    """Error codes, never crashes: null handles, one of real / imaginary without the other, output lengths without grids."""
    import ctypes
    from nmrfit_amd import _cabi
    L = _cabi.lib()
    sp = synth.make_spectrum(1024, 2, seed=3)
    x = _cabi.f64(sp["x_true"])
    out = np.empty((2, 1024))
    assert L.nmrfit_generate_result(None, 2, _cabi.ptr(x), 0, None, _cabi.ptr(out), _cabi.ptr(out), None, None) == _cabi.E_INVALID
    assert L.nmrfit_batch_contributions(None, None, None, None, None, None, None) == _cabi.E_INVALID
    with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        h = ev.handle
        assert L.nmrfit_generate_result(h, 2, None, 0, None, _cabi.ptr(out), _cabi.ptr(out), None, None) == _cabi.E_INVALID
        assert L.nmrfit_generate_result(h, 2, _cabi.ptr(x), 0, None, _cabi.ptr(out), None, None, None) == _cabi.E_INVALID   # real without imag
        assert L.nmrfit_generate_result(h, -1, _cabi.ptr(x), 0, None, None, None, None, None) == _cabi.E_INVALID
        assert L.nmrfit_generate_result(h, 2, _cabi.ptr(x), 0, None, None, None, None, None) == _cabi.OK                     # nothing asked for
        fit = np.empty((4, 1024))
        assert L.nmrfit_generate_result(h, 2, _cabi.ptr(x), 0, None, None, None, _cabi.ptr(fit), None) == _cabi.OK           # the sums alone
        real, imag, fit2, _ = ev.generate_result(x)
        np.testing.assert_array_equal(fit, fit2)
    with FitBatch([(sp["w"], sp["u"], sp["v"], sp["weights"])], [sp["lower"]], [sp["upper"]], swarmsize=8, seeds=[1]) as fb:
        fb.run(2, 2)
        n = np.array([1024], dtype=np.int64)
        assert L.nmrfit_batch_contributions(fb._h, _cabi.ptr(n), None, _cabi.ptr(out), _cabi.ptr(out), None, None) == _cabi.E_INVALID   # lengths without grids
        assert L.nmrfit_batch_contributions(fb._h, None, None, _cabi.ptr(out), None, None, None) == _cabi.E_INVALID               # real without imag
        assert L.nmrfit_batch_contributions(fb._h, None, None, None, None, None, None) == _cabi.OK
        neg = np.array([-5], dtype=np.int64)
        assert L.nmrfit_batch_contributions(fb._h, _cabi.ptr(neg), _cabi.ptr(out), None, None, _cabi.ptr(out), None) == _cabi.E_INVALID
        assert len(fb.generate()) == 1                    # (the batch is still usable after the refused calls)

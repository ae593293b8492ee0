"""
GPU tier, row f2 (BASELINE config 5): residual vectors for a forward-difference Jacobian in
one batched launch, and the scipy.least_squares adapter built on it.
"""
import os

import numpy as np
import pytest

from nmrfit_amd import _cabi, lsq, synth

pytestmark = pytest.mark.gpu


def test_c5_jacobian_rows_match_reference_golden(golden_dir):
    """The C5 fixture holds the reference's objective for the D+1 rows and two residual rows."""
    from nmrfit_amd.equations import Evaluator
    g = np.load(os.path.join(golden_dir, "objective_P12_N16384.npz"))
    with Evaluator(g["w"], g["u"], g["v"], g["weights"]) as ev:
        R, f = ev.residual_batch(g["X"], return_f=True)
        assert R.shape == (41, 16384)
        np.testing.assert_allclose(f, g["f"], rtol=1e-9)
        np.testing.assert_allclose(R[g["R_rows"]], g["R"], rtol=0, atol=1e-11 * np.abs(g["R"]).max())
        # the model's rows are the fixture's rows (same step rule as synth.jacobian_rows)
        m = lsq.ResidualModel(ev)
        rows, h = m.rows(g["X"][0])
        np.testing.assert_array_equal(rows, g["X"])
        np.testing.assert_allclose(h, g["h"], rtol=1e-8)


def test_jacobian_against_oracle_finite_differences():
    from nmrfit_amd.equations import Evaluator
    from oracle import c_oracle
    sp = synth.make_spectrum(4096, 4, seed=13)
    x = synth.make_swarm(sp["lower"], sp["upper"], 3, seed=14)[2]
    with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        m = lsq.ResidualModel(ev, sp["lower"], sp["upper"])
        J = m.jac(x)
        r0 = m.fun(x)
        rows, h = m.rows(x)
    assert J.shape == (4096, 16)
    Rref, fref = c_oracle.residual_batch(rows, sp["w"], sp["u"], sp["v"], sp["weights"], threads=8)
    Jref = ((Rref[1:] - Rref[0]) / h[:, None]).T / np.sqrt(4096)
    # differences of nearly equal residuals divided by h ~ 1e-8: 1e-13 absolute parity in R
    # becomes ~1e-5 relative in J; columns are compared on their own scale
    scale = np.abs(Jref).max(axis=0)
    np.testing.assert_allclose(J / scale, Jref / scale, atol=5e-5)
    assert np.linalg.norm(r0) == pytest.approx(fref[0], rel=1e-9)      # ||fun|| is the objective
    # steps flip at the upper bound
    xb = np.array(sp["upper"], dtype=float)
    assert (m.steps(xb) < 0).all()


def test_least_squares_beats_the_swarm_and_recovers_truth():
    from nmrfit_amd.equations import Evaluator
    from nmrfit_amd import pso
    from oracle import c_oracle
    # physical=True: the imaginary channel carries the dispersive line shape, so the phase is
    # identifiable and the generating parameters are the (noisy) optimum
    sp = synth.make_spectrum(4096, 3, seed=9, physical=True)
    with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        xs, fs = pso.pso(ev, sp["lower"], sp["upper"], swarmsize=204, maxiter=300, seed=5, verbose=False,
                         omega=pso.DEFAULTS["omega"], phip=pso.DEFAULTS["phip"], phig=pso.DEFAULTS["phig"])
        res = lsq.least_squares(ev, xs, sp["lower"], sp["upper"])
        xp, fp, _ = lsq.polish(ev, xs, sp["lower"], sp["upper"])
    f_truth = c_oracle.objective_batch(sp["x_true"], sp["w"], sp["u"], sp["v"], sp["weights"])[0]
    assert res.success and res.objective <= fs * (1 + 1e-12)
    assert res.objective <= f_truth * (1 + 1e-9)          # a local optimum at or below the noisy truth
    assert fp == res.objective and (xp == res.x).all()
    assert (res.x >= sp["lower"]).all() and (res.x <= sp["upper"]).all()
    ref = c_oracle.objective_batch(res.x, sp["w"], sp["u"], sp["v"], sp["weights"])[0]
    assert res.objective == pytest.approx(ref, rel=1e-9)
    np.testing.assert_allclose(res.x[6::3], sp["x_true"][6::3], rtol=0.05)    # areas within 5 %
    np.testing.assert_allclose(res.x[5::3], sp["x_true"][5::3], atol=2e-5)    # locations


def test_fit_with_polish_option():
    import nmrfit_amd
    sp = synth.make_spectrum(2048, 2, seed=4)
    data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
    a = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False,
                       options={"swarmsize": 64, "maxiter": 100, "seed": 3})
    b = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False,
                       options={"swarmsize": 64, "maxiter": 100, "seed": 3, "polish": True})
    assert b.error <= a.error


def test_fit_many_with_polish_equals_the_lone_fits():
    """options['polish'] in fit_many: the swarms run as one device batch, the refinement per fit on host threads with a
    context made like fit()'s -- params and error equal the lone fit(polish=True) bit for bit; with generate=True the
    reconstruction is made from the REFINED parameters."""
    import nmrfit_amd
    specs = [synth.make_spectrum(2048, 2 + k % 2, seed=30 + k, physical=True) for k in range(5)]

    def jobs():
        return [dict(data=synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), lower=list(sp["lower"]), upper=list(sp["upper"]),
                     options={"swarmsize": 64, "maxiter": 100, "seed": 3 + k, "polish": k != 1}) for k, sp in enumerate(specs)]
    many = nmrfit_amd.fit_many(jobs(), generate=True, threads=3)
    for k, job in enumerate(jobs()):
        one = nmrfit_amd.fit(job["data"], job["lower"], job["upper"], summary=False, options=job["options"])
        one.generate_result()
        np.testing.assert_array_equal(many[k].params, one.params, err_msg=str(k))
        assert many[k].error == one.error
        np.testing.assert_array_equal(many[k].V, one.V)
        np.testing.assert_array_equal(many[k].u, one.u)

"""
CPU tier (i): the oracle (numpy + plain-C restatements under oracle/) against the golden
vectors the REFERENCE itself produced (oracle/make_golden.py -> tests/golden/*.npz).
This is what pins the oracle; the HIP path is then checked against oracle + fixtures in
the -m gpu tests.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import nmrfit_oracle as onp
from oracle import c_oracle as oc
from nmrfit_amd import synth

RTOL = 1e-13     # oracle vs reference: same float64 ops, allow libm / summation-order noise


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["objective_P6_N4096.npz", "objective_P12_N16384.npz"])
def test_objective_and_residual_full_inputs(golden_dir, name):
    g = _load(golden_dir, name)
    w, u, v, wt, X = g["w"], g["u"], g["v"], g["weights"], g["X"]
    f_np = onp.objective_batch(X, w, u, v, wt)
    # op-for-op numpy restatement: bitwise on this machine, tolerance for portability
    np.testing.assert_allclose(f_np, g["f"], rtol=1e-15, atol=0)
    f_c = oc.objective_batch(X, w, u, v, wt)
    np.testing.assert_allclose(f_c, g["f"], rtol=RTOL, atol=0)
    R_c, f_c2 = oc.residual_batch(X[g["R_rows"]], w, u, v, wt)
    scale = np.abs(g["R"]).max()
    np.testing.assert_allclose(R_c, g["R"], rtol=0, atol=1e-13 * scale)
    np.testing.assert_allclose(onp.residual_batch(X[g["R_rows"]], w, u, v, wt), g["R"], rtol=0, atol=1e-15 * scale)
    np.testing.assert_allclose(f_c2, g["f"][g["R_rows"]], rtol=RTOL)


def test_generator_is_stable_and_c3_shape(golden_dir):
    """The C3-shape fixture stores only seeds + sha256 of the inputs: the generator must
    reproduce them bit for bit, then the oracle must reproduce the reference's f."""
    g = _load(golden_dir, "objective_P24_N65536.npz")
    sp = synth.make_spectrum(65536, 24, seed=int(g["seed"]))
    for k in ("w", "u", "v", "weights"):
        assert _sha(sp[k]) == str(g["sha_" + k]), "synthetic generator drifted for " + k
    X = synth.make_swarm(sp["lower"], sp["upper"], 8, seed=int(g["swarm_seed"]), x_true=sp["x_true"])
    np.testing.assert_array_equal(X, g["X"])
    f_c = oc.objective_batch(X, sp["w"], sp["u"], sp["v"], sp["weights"], threads=4)
    np.testing.assert_allclose(f_c, g["f"], rtol=RTOL)
    f_np = onp.objective_batch(X[:2], sp["w"], sp["u"], sp["v"], sp["weights"])
    np.testing.assert_allclose(f_np, g["f"][:2], rtol=1e-15)


def test_generator_matches_stored_small_inputs(golden_dir):
    g = _load(golden_dir, "objective_P6_N4096.npz")
    sp = synth.make_spectrum(4096, 6, seed=int(g["seed"]))
    for k in ("w", "u", "v", "weights"):
        np.testing.assert_array_equal(sp[k], g[k])
    X = synth.make_swarm(sp["lower"], sp["upper"], 50, seed=int(g["swarm_seed"]), x_true=sp["x_true"])
    np.testing.assert_array_equal(X, g["X"])


def test_edge_cases(golden_dir):
    g = _load(golden_dir, "objective_edge_cases.npz")
    n = int(g["n_cases"])
    assert n >= 29
    for i in range(n):
        k = "e%d" % i
        w, u, v, wt, x = g[k + "_w"], g[k + "_u"], g[k + "_v"], g[k + "_wt"], g[k + "_x"]
        f_ref, R_ref = float(g[k + "_f"]), g[k + "_R"]
        assert np.isfinite(f_ref)
        f_np = onp.objective(x, w, u, v, wt)
        assert f_np == pytest.approx(f_ref, rel=1e-15), k
        R_c, f_c = oc.residual_batch(x, w, u, v, wt)
        # large phases (|phi| ~ 1e3) amplify libm differences: 1e-12 relative is ample
        assert f_c[0] == pytest.approx(f_ref, rel=1e-12), k
        np.testing.assert_allclose(R_c[0], R_ref, rtol=0, atol=1e-12 * max(1.0, np.abs(R_ref).max()), err_msg=k)


def test_primitives(golden_dir):
    g = _load(golden_dir, "primitives.npz")
    p0, p1 = g["ps2_args"]
    V1, I1 = onp.ps2(g["u"], g["v"], p0=p0, p1=p1)
    np.testing.assert_array_equal(V1, g["V1"])
    np.testing.assert_array_equal(I1, g["I1"])
    V2, I2 = onp.ps2(g["u"], g["v"], p0=p0, p1=p1, inv=True)
    np.testing.assert_array_equal(V2, g["V2"])
    np.testing.assert_array_equal(I2, g["I2"])
    # SURVEY section 4 property 2: ps2 then inverse ps2 is the identity
    ub, vb = onp.ps2(V1, I1, p0=p0, p1=p1, inv=True)
    np.testing.assert_allclose(ub, g["u"], atol=1e-14)
    np.testing.assert_allclose(vb, g["v"], atol=1e-14)
    np.testing.assert_array_equal(onp.voigt(g["w"], *g["voigt_args"]), g["voigt"])


def test_voigt_area_normalised():
    """SURVEY section 4 property 1: integral of (voigt - yoff) over a wide grid is `a`."""
    w = np.linspace(-2000.0, 2000.0, 4_000_001)
    for r in (0.0, 0.4, 1.0):
        V = onp.voigt(w, r, 0.0, 0.7, 0.3, 2.5)
        area = np.sum(0.5 * (V[1:] + V[:-1]) * np.diff(w))
        # Lorentzian tails beyond +-2000: a*r*(2/pi)*atan-tail ~ a*r*width/(pi*2000)
        assert area == pytest.approx(2.5, abs=2.5 * r * 0.7 / (np.pi * 2000) * 1.1 + 1e-9)


def test_laplace_and_weights(golden_dir):
    g = _load(golden_dir, "weights.npz")
    np.testing.assert_array_equal(onp.laplace1d(g["lap_in"].copy()), g["lap_out"])
    np.testing.assert_array_equal(onp.laplace1d(g["lap_in"].copy(), n=3, omega=0.5), g["lap3_out"])
    assert g["lap_out"][0] == g["lap_in"][0] and g["lap_out"][-1] == g["lap_in"][-1]
    if "cw_weights" in g.files:
        sp = synth.make_spectrum(4096, 6, seed=int(g["cw_seed"]))
        np.testing.assert_array_equal(onp.compute_weights(sp["w"], sp["peaks"], 0.5), g["cw_weights"])
        np.testing.assert_array_equal(onp.compute_weights(sp["w"], sp["peaks"], 1.3), g["cw_weights_e13"])
        np.testing.assert_array_equal(onp.compute_weights(sp["w"][::-1], sp["peaks"], 0.5), g["cw_weights_rev"])


def test_objective_noiseless_truth_is_zero():
    """SURVEY section 4 property 3."""
    sp = synth.make_spectrum(4096, 6, seed=1, noise=0.0)
    f = onp.objective(sp["x_true"], sp["w"], sp["u"], sp["v"], sp["weights"])
    assert f < 1e-14


def test_objective_shift_invariance():
    """SURVEY section 4 property 4: shifting w and every loc together changes nothing
    (up to rounding of w - loc)."""
    sp = synth.make_spectrum(1024, 3, seed=5)
    x = synth.make_swarm(sp["lower"], sp["upper"], 2, seed=6)[1]
    f0 = oc.objective_batch(x, sp["w"], sp["u"], sp["v"], sp["weights"])[0]
    x2 = x.copy()
    x2[5::3] += 0.25
    f1 = oc.objective_batch(x2, sp["w"] + 0.25, sp["u"], sp["v"], sp["weights"])[0]
    assert f1 == pytest.approx(f0, rel=1e-11)


def test_float32_spectra_reference_golden(golden_dir):
    """tests/golden/objective_float32.npz (oracle/make_golden.py section 9): the reference run on
    float32 u, v -- its complex64 rotation (proc_autophase.py:29-32).  The numpy oracle follows that
    path exactly; inputs are regenerated by seed and pinned by their sha256."""
    d = np.load(os.path.join(golden_dir, "objective_float32.npz"))
    for tag in ("P6_N4096", "P12_N16384", "P24_N65536"):
        N, P, seed = (int(t) for t in d[tag + "_shape"])
        sp = synth.make_spectrum(N, P, seed=seed)
        u32, v32 = sp["u"].astype(np.float32), sp["v"].astype(np.float32)
        assert hashlib.sha256(u32.tobytes()).hexdigest() == str(d[tag + "_sha_u32"])
        assert hashlib.sha256(v32.tobytes()).hexdigest() == str(d[tag + "_sha_v32"])
        X, f = d[tag + "_X"], d[tag + "_f"]
        rows = range(X.shape[0]) if N <= 16384 else (0, 1, 5, 9)      # (the C3 shape takes 25 ms a row)
        got = np.array([onp.objective(X[i], sp["w"], u32, v32, sp["weights"]) for i in rows])
        np.testing.assert_array_equal(got, f[list(rows)])
        # near the optimum the complex64 rotation moves f by ~1e-7 relative: inside the 1e-6 bar
        rel = np.abs(f - d[tag + "_f_float64_spectrum"]) / d[tag + "_f_float64_spectrum"]
        assert rel.max() < 5e-7

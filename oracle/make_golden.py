#!/usr/bin/env python3
"""
ORACLE TOOLING — TEST INFRASTRUCTURE ONLY.  Runs in the build container only.

Generates tests/golden/*.npz by executing the REFERENCE's own hot-path code
(/root/reference/nmrfit/equations.py and proc_autophase.py, loaded in place, never
copied) on seeded synthetic inputs.  The fixtures are data (inputs + expected outputs);
no reference source travels.

How the reference is loaded:
  * `import nmrfit` fails with ModuleNotFoundError (nmrglue/peakutils/pyswarm are not
    installed), so only the two hot-path modules are loaded with
    importlib.util.spec_from_file_location under an empty stub package `nmrfit`
    (equations.py:6 does `from . import proc_autophase`);
  * equations.py:242 uses `np.float`, removed in numpy>=1.24 -> `np.float = float`
    before loading; utils.py:201-202 uses `np.int` likewise.
  * For FitUtility._compute_weights (utils.py:191-224) utils.py is loaded with EMPTY
    placeholder modules registered for its missing third-party imports (nmrglue,
    peakutils, pyswarm).  Nothing from those placeholders is ever called: only
    `_compute_weights` and `equations.laplace1d` execute.

Refuses to run when /root/reference is absent (e.g. on the GPU box).

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/nmrfit"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def load_reference():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py: /root/reference is not present; fixtures can only "
                         "be generated in the build container")
    if not hasattr(np, "float"):
        np.float = float          # equations.py:242
    if not hasattr(np, "int"):
        np.int = int              # utils.py:201-202
    pkg = types.ModuleType("nmrfit")
    pkg.__path__ = [REF]
    sys.modules["nmrfit"] = pkg
    mods = {}
    for name in ("proc_autophase", "equations"):
        spec = importlib.util.spec_from_file_location("nmrfit." + name, os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["nmrfit." + name] = m
        spec.loader.exec_module(m)
        setattr(pkg, name, m)
        mods[name] = m
    # utils.py: only for FitUtility._compute_weights
    try:
        for missing in ("nmrglue", "peakutils", "pyswarm"):
            if missing not in sys.modules:
                sys.modules[missing] = types.ModuleType(missing)
        import matplotlib
        matplotlib.use("Agg")
        spec = importlib.util.spec_from_file_location("nmrfit.utils", os.path.join(REF, "utils.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["nmrfit.utils"] = m
        spec.loader.exec_module(m)
        mods["utils"] = m
    except Exception as e:            # ordinary Python error -> weights fixture skipped
        print("utils.py not loadable (%s: %s); weights fixture from laplace1d only" % (type(e).__name__, e))
        mods["utils"] = None
    return mods


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_residual(eq, pa, x, w, u, v, weights):
    """weights*(V_data - V_fit) recomputed from the reference's ps2 + voigt exactly as
    equations.py:180-202 does (objective itself only returns the scalar)."""
    p0, p1, r, yoff = x[:4]
    V_data, _ = pa.ps2(u, v, p0=p0, p1=p1)
    V_fit = np.zeros_like(V_data)
    for i in range(4, len(x), 3):
        V_fit = V_fit + eq.voigt(w, r, yoff, x[i], x[i + 1], x[i + 2])
    return np.multiply(weights, (V_data - V_fit))


def main():
    mods = load_reference()
    eq, pa = mods["equations"], mods["proc_autophase"]
    from nmrfit_amd import synth

    os.makedirs(OUT, exist_ok=True)

    def fobj(X, sp):
        return np.array([eq.objective(X[i, :], sp["w"], sp["u"], sp["v"], sp["weights"], False)
                         for i in range(X.shape[0])])

    # --- 1. C1/C2 shape: full inputs stored -------------------------------------------
    sp = synth.make_spectrum(4096, 6, seed=1)
    X = synth.make_swarm(sp["lower"], sp["upper"], 50, seed=2, x_true=sp["x_true"])
    f = fobj(X, sp)
    R = np.stack([ref_residual(eq, pa, X[i], sp["w"], sp["u"], sp["v"], sp["weights"]) for i in (0, 1, 49)])
    np.savez_compressed(os.path.join(OUT, "objective_P6_N4096.npz"),
                        w=sp["w"], u=sp["u"], v=sp["v"], weights=sp["weights"], X=X, f=f,
                        R_rows=np.array([0, 1, 49]), R=R, seed=1, swarm_seed=2)
    print("P6_N4096   f[0]=%.17g f[1]=%.17g" % (f[0], f[1]))

    # --- 2. C5 shape: Jacobian rows, inputs stored -------------------------------------
    sp = synth.make_spectrum(16384, 12, seed=3)
    x0 = synth.make_swarm(sp["lower"], sp["upper"], 2, seed=4, x_true=None)[1]
    rows, h = synth.jacobian_rows(x0)
    f = fobj(rows, sp)
    R = np.stack([ref_residual(eq, pa, rows[i], sp["w"], sp["u"], sp["v"], sp["weights"]) for i in (0, 7)])
    np.savez_compressed(os.path.join(OUT, "objective_P12_N16384.npz"),
                        w=sp["w"], u=sp["u"], v=sp["v"], weights=sp["weights"], X=rows, h=h, f=f,
                        R_rows=np.array([0, 7]), R=R, seed=3, swarm_seed=4)
    print("P12_N16384 f[0]=%.17g" % f[0])

    # --- 3. C3 shape: inputs by seed (sha256 recorded), 8 particles --------------------
    sp = synth.make_spectrum(65536, 24, seed=1)
    X = synth.make_swarm(sp["lower"], sp["upper"], 8, seed=2, x_true=sp["x_true"])
    f = fobj(X, sp)
    np.savez_compressed(os.path.join(OUT, "objective_P24_N65536.npz"),
                        X=X, f=f, seed=1, swarm_seed=2,
                        sha_w=sha(sp["w"]), sha_u=sha(sp["u"]), sha_v=sha(sp["v"]),
                        sha_weights=sha(sp["weights"]))
    print("P24_N65536 f[0]=%.17g" % f[0])

    # --- 4. edge cases: ragged N, P=0/1, extreme parameters, float noise-free truth -----
    rng = np.random.default_rng(11)
    cases = {}
    idx = 0
    for N in (1, 2, 63, 64, 65, 257, 1000):
        for P in (0, 1, 3):
            w = np.sort(rng.uniform(-2.0, 5.0, N))            # non-uniform, crosses zero
            u = rng.standard_normal(N)
            v = rng.standard_normal(N)
            wt = 0.5 + rng.random(N)
            x = np.empty(4 + 3 * P)
            x[:4] = (rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi, np.pi), rng.random(), rng.uniform(-0.01, 0.01))
            x[4::3] = rng.uniform(0.01, 2.0, P)
            x[5::3] = rng.uniform(-2.0, 5.0, P)
            x[6::3] = rng.uniform(0.1, 3.0, P)
            cases["e%d" % idx] = (w, u, v, wt, x)
            idx += 1
    # extreme parameter rows on one 512-point grid
    N = 512
    w = np.linspace(3.0, 4.0, N)
    u = rng.standard_normal(N) * 0.1
    v = rng.standard_normal(N) * 0.1
    wt = np.ones(N)
    extreme = [
        [0.0, 0.0, 0.5, 0.0, 1e-5, 3.5, 1.0],              # needle: Gaussian underflows off-centre
        [0.0, 0.0, 0.5, 0.0, 50.0, 3.5, 1.0],              # much wider than the grid
        [3.0, 50.0, 0.5, 0.01, 0.01, 9.0, 1.0],            # loc outside the grid, large p1
        [-40.0, -75.5, 0.0, -0.01, 0.02, 3.3, -2.0],       # pure Gaussian, negative area, large |phase|
        [1.0, 1.0, 1.0, 0.0, 0.02, 3.7, 2.0],              # pure Lorentzian
        [0.1, 0.2, 1.7, 0.003, 0.02, 3.7, 2.0],            # r outside [0,1]
        [1e3, -1e3, 0.3, 0.0, 0.003, 3.50001, 0.5],        # phases far outside [-pi,pi]
        [0.3, -0.2, 0.6, 0.002, 0.004, 3.2, 0.005, 0.005, 3.2000001, 0.007],   # two nearly coincident peaks
    ]
    for x in extreme:
        cases["e%d" % idx] = (w, u, v, wt, np.array(x, dtype=float))
        idx += 1
    out = {}
    for k, (w_, u_, v_, wt_, x_) in cases.items():
        out[k + "_w"], out[k + "_u"], out[k + "_v"], out[k + "_wt"], out[k + "_x"] = w_, u_, v_, wt_, x_
        out[k + "_f"] = np.float64(eq.objective(x_, w_, u_, v_, wt_, False))
        out[k + "_R"] = ref_residual(eq, pa, x_, w_, u_, v_, wt_)
    out["n_cases"] = idx
    np.savez_compressed(os.path.join(OUT, "objective_edge_cases.npz"), **out)
    print("edge cases:", idx)

    # --- 5. ps2 / voigt primitives -------------------------------------------------------
    N = 777
    w = np.linspace(-1.0, 2.0, N)
    u = rng.standard_normal(N)
    v = rng.standard_normal(N)
    V1, I1 = pa.ps2(u, v, p0=0.7, p1=-2.3)
    V2, I2 = pa.ps2(u, v, p0=0.7, p1=-2.3, inv=True)
    vg = eq.voigt(w, 0.35, 0.004, 0.05, 0.61, 1.7)
    np.savez_compressed(os.path.join(OUT, "primitives.npz"), w=w, u=u, v=v, V1=V1, I1=I1, V2=V2, I2=I2,
                        voigt=vg, voigt_args=np.array([0.35, 0.004, 0.05, 0.61, 1.7]),
                        ps2_args=np.array([0.7, -2.3]))

    # --- 6. laplace1d + _compute_weights --------------------------------------------------
    x = rng.random(300) * 3
    lap = eq.laplace1d(x.copy())
    lap3 = eq.laplace1d(x.copy(), n=3, omega=0.5)
    d = dict(lap_in=x, lap_out=lap, lap3_out=lap3)
    if mods["utils"] is not None:
        sp = synth.make_spectrum(4096, 6, seed=1)

        class _D:
            pass
        data = _D()
        data.w, data.u, data.v, data.peaks = sp["w"], sp["u"], sp["v"], sp["peaks"]
        fu = mods["utils"].FitUtility(data, list(sp["lower"]), list(sp["upper"]), expon=0.5)
        wts = fu._compute_weights()
        fu2 = mods["utils"].FitUtility(data, list(sp["lower"]), list(sp["upper"]), expon=1.3)
        # reversed grid (core.py:60 hands out reversed views) exercises the lIdx>rIdx swap
        data_r = _D()
        data_r.w, data_r.u, data_r.v, data_r.peaks = sp["w"][::-1], sp["u"][::-1], sp["v"][::-1], sp["peaks"]
        fu3 = mods["utils"].FitUtility(data_r, list(sp["lower"]), list(sp["upper"]), expon=0.5)
        d.update(cw_seed=1, cw_weights=wts, cw_weights_e13=fu2._compute_weights(),
                 cw_weights_rev=fu3._compute_weights(),
                 cw_heights=np.array([p.height for p in sp["peaks"]]),
                 cw_bounds=np.array([p.bounds for p in sp["peaks"]]))
        print("compute_weights fixture written (min %.4f max %.4f)" % (wts.min(), wts.max()))
    np.savez_compressed(os.path.join(OUT, "weights.npz"), **d)

    # --- 6b. Data.generate_solution_bounds (containers.py:175-217), loaded like utils.py ---------
    try:
        spec = importlib.util.spec_from_file_location("nmrfit.containers", os.path.join(REF, "containers.py"))
        cm = importlib.util.module_from_spec(spec)
        sys.modules["nmrfit.containers"] = cm
        spec.loader.exec_module(cm)
        sp = synth.make_spectrum(4096, 6, seed=1)
        dat = cm.Data(sp["w"], sp["u"], sp["v"])
        dat.peaks = sp["peaks"]
        dat.p0, dat.p1 = 0.31, -0.17
        lo1, up1 = dat.generate_solution_bounds()
        lo2, up2 = dat.generate_solution_bounds(force_p0=True, force_p1=True)
        np.savez_compressed(os.path.join(OUT, "bounds.npz"), seed=1, p0=0.31, p1=-0.17, lower=np.array(lo1),
                            upper=np.array(up1), lower_forced=np.array(lo2), upper_forced=np.array(up2))
        print("bounds fixture written")
    except Exception as e:
        print("containers.py not loadable (%s: %s); bounds fixture skipped" % (type(e).__name__, e))

    # --- 7. Kramers-Kronig path: fit_im=True objective and generate_result pieces ---------------
    # (reference: scipy quad per grid point, ~4 ms each -> small grids only)
    sp = synth.make_spectrum(160, 3, seed=7, physical=True)
    X = synth.make_swarm(sp["lower"], sp["upper"], 4, seed=8, x_true=sp["x_true"])
    f_im = np.array([eq.objective(X[i, :], sp["w"], sp["u"], sp["v"], sp["weights"], True) for i in range(4)])
    f_re = np.array([eq.objective(X[i, :], sp["w"], sp["u"], sp["v"], sp["weights"], False) for i in range(4)])
    x = X[1]
    p0, p1, r, yoff = x[:4]
    real_c = np.stack([eq.voigt(sp["w"], r, yoff, x[i], x[i + 1], x[i + 2]) for i in range(4, len(x), 3)])
    imag_c = np.stack([eq.kk_relation_vectorized(sp["w"], r, yoff, x[i], x[i + 1], x[i + 2])
                       for i in range(4, len(x), 3)])
    V_ph, I_ph = pa.ps2(sp["u"], sp["v"], p0=p0, p1=p1)
    u_fit, v_fit = pa.ps2(real_c.sum(axis=0), imag_c.sum(axis=0), inv=True, p0=p0, p1=p1)
    # upsampled grid (generate_result(scale=1.5), utils.py:241) for one peak
    w_up = np.linspace(sp["w"].min(), sp["w"].max(), int(1.5 * sp["w"].shape[0]))
    imag_up = eq.kk_relation_vectorized(w_up, r, yoff, x[4], x[5], x[6])
    real_up = eq.voigt(w_up, r, yoff, x[4], x[5], x[6])
    # a wide and a needle line, grid crossing zero
    w2 = np.linspace(-1.0, 1.0, 96)
    imag_wide = eq.kk_relation_vectorized(w2, 0.25, 0.01, 0.6, 0.13, 1.7)
    imag_needle = eq.kk_relation_vectorized(w2, 0.8, 0.0, 0.004, -0.21, 0.02)
    np.savez_compressed(os.path.join(OUT, "kramers_kronig.npz"), w=sp["w"], u=sp["u"], v=sp["v"],
                        weights=sp["weights"], X=X, f_fit_im=f_im, f_real=f_re, x=x, real_contribs=real_c,
                        imag_contribs=imag_c, V=V_ph, I=I_ph, u_fit=u_fit, v_fit=v_fit, w_up=w_up,
                        imag_up=imag_up, real_up=real_up, w2=w2, imag_wide=imag_wide, imag_needle=imag_needle,
                        wide_args=np.array([0.25, 0.01, 0.6, 0.13, 1.7]),
                        needle_args=np.array([0.8, 0.0, 0.004, -0.21, 0.02]))
    print("kramers_kronig fixture: f_fit_im", f_im)
    print("done ->", OUT)



def data_fixtures(mods):
    """Section 8: the callers either side of the path -- proc_autophase's estimators
    (proc_autophase.py:39-219), the Data container's scripted methods (containers.py:51-130,
    219-252) and the deterministic helpers of utils.py (Peaks :14-55, find_peak :819-853,
    sample_noise :878-902, BoundsSelector.apply_bounds :416-442).  AutoPeakSelector needs
    peakutils (absent) and scipy.integrate.simps (removed from scipy): not generated."""
    from nmrfit_amd import synth
    pa, ut = mods["proc_autophase"], mods["utils"]
    spec = importlib.util.spec_from_file_location("nmrfit.containers", os.path.join(REF, "containers.py"))
    cm = importlib.util.module_from_spec(spec)
    sys.modules["nmrfit.containers"] = cm
    spec.loader.exec_module(cm)
    sp = synth.make_spectrum(2048, 3, seed=21, physical=True)
    w, u, v = sp["w"], sp["u"], sp["v"]
    z = u + 1j * v
    out = dict(w=w, u=u, v=v, seed=21)
    out["ps_deg"] = pa.ps(z, p0=33.0, p1=-71.5)
    out["ps_deg_inv"] = pa.ps(z, p0=33.0, p1=-71.5, inv=True)
    phases = np.array([[0.0, 0.0], [17.2, -11.5], [-40.0, 25.0], [5.0, 180.0]])
    out["score_phases"] = phases
    out["acme"] = np.array([pa._ps_acme_score(ph, z) for ph in phases])
    out["peak_minima"] = np.array([pa._ps_peak_minima_score(ph, z) for ph in phases])
    out["approx_acme"] = np.array(pa.approximate_phase(z, "acme"))
    out["approx_minima"] = np.array(pa.approximate_phase(z, "peak_minima", p0=5.0, p1=-3.0))
    out["autops_acme"] = pa.autops(z, "acme")
    # Data: manual / auto / brute phase, cropping, area helpers
    d = cm.Data(w.copy(), u.copy(), v.copy())
    d.shift_phase(method="manual", p0=0.2, p1=-0.1)
    out["manual_V"], out["manual_I"] = d.V, d.I
    d.shift_phase(method="auto")
    out["auto_p"] = np.array([d.p0, d.p1])
    out["auto_V"] = d.V
    d.shift_phase(method="brute", step=np.pi / 90)
    out["brute_p"] = np.array([d.p0, d.p1])
    out["brute_V"] = d.V
    d.select_bounds(low=3.2, high=3.8)
    out["crop_w"], out["crop_u"], out["crop_v"] = d.w, d.u, d.v
    d.peaks = sp["peaks"]
    out["areas"] = np.array(d.approximate_areas())
    out["area_fraction"] = d.approximate_area_fraction()
    # utils helpers
    pk = ut.Peaks()
    for h in (3.0, -0.4, 2.5, 0.3, 0.35):
        q = ut.Peak()
        q.height = h
        pk.append(q)
    main, sats = pk.split()
    out["avg_height"] = pk.average_height()
    out["split_main"] = np.array([q.height for q in main])
    out["split_sats"] = np.array([q.height for q in sats])
    V = out["manual_V"]
    out["find_peak"] = np.array(ut.find_peak(w, V, 3.3, 3.7), dtype=float)
    out["sample_noise"] = ut.sample_noise(w, V, 3.0, 3.05)
    np.savez_compressed(os.path.join(OUT, "data_container.npz"), **out)
    print("data_container fixture: auto phase", out["auto_p"], "brute", out["brute_p"], "truth", sp["x_true"][:2])


def float32_fixture(mods):
    """Section 9 (round 3): float32 spectra as nmrglue delivers them.  The
    reference then rotates u + i v in complex64 (proc_autophase.py:29-32: apod.astype(data.dtype))
    and only the w array promotes V_fit to float64 (equations.py:181,195); the C-ABI takes float64
    and the host side upcasts, so the two differ by the float32 rounding of the rotation.  Three
    shapes (C1/C2, C5, C3), inputs by seed with u, v cast to float32 (sha256 of the float32 arrays
    recorded), rows: the generating parameters, eight near-optimum particles (the generating
    parameters perturbed by 1e-3 of the box, where f is smallest and the relative gap largest) and
    three random particles of the box."""
    from nmrfit_amd import synth
    eq = mods["equations"]
    out = {}
    for tag, (N, P, seed) in {"P6_N4096": (4096, 6, 1), "P12_N16384": (16384, 12, 3), "P24_N65536": (65536, 24, 1)}.items():
        sp = synth.make_spectrum(N, P, seed=seed)
        u32, v32 = sp["u"].astype(np.float32), sp["v"].astype(np.float32)
        rng = np.random.default_rng(1000 + seed + P)
        box = sp["upper"] - sp["lower"]
        near = np.clip(sp["x_true"][None, :] + 1e-3 * box[None, :] * rng.uniform(-1.0, 1.0, (8, box.size)),
                       sp["lower"], sp["upper"])
        X = np.concatenate((sp["x_true"][None, :], near,
                            synth.make_swarm(sp["lower"], sp["upper"], 3, seed=77 + P, x_true=None)))
        f = np.array([eq.objective(X[i, :], sp["w"], u32, v32, sp["weights"], False) for i in range(X.shape[0])])
        out[tag + "_X"], out[tag + "_f"] = X, f
        out[tag + "_shape"] = np.array([N, P, seed])
        out[tag + "_sha_u32"], out[tag + "_sha_v32"] = sha(u32), sha(v32)
        f64 = np.array([eq.objective(X[i, :], sp["w"], sp["u"], sp["v"], sp["weights"], False) for i in range(X.shape[0])])
        out[tag + "_f_float64_spectrum"] = f64     # the same rows with the float64 spectrum, for scale
        print("float32 %-11s f[0]=%.17g  max |f32 - f64|/f64 = %.2e" % (tag, f[0], np.max(np.abs(f - f64) / f64)))
    np.savez_compressed(os.path.join(OUT, "objective_float32.npz"), **out)


if __name__ == "__main__":
    if "--only-data" in sys.argv:
        data_fixtures(load_reference())
    elif "--only-float32" in sys.argv:
        float32_fixture(load_reference())
    else:
        main()
        mods = load_reference()
        data_fixtures(mods)
        float32_fixture(mods)

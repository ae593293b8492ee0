"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU (numpy) restatement of the nmrfit objective hot path, op-for-op in the order the
reference performs its whole-array numpy passes.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; nothing under ``nmrfit_amd/`` does (the product path raises if the HIP
library is missing rather than falling back to this).

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference's own
``equations.py`` / ``proc_autophase.py`` in the build container and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function here
against those vectors (<= 1e-13 relative).  The PSO loop (``pso``) restates the
third-party ``pyswarm.pso`` (github.com/tisimst/pyswarm, unpinned master, absent from
/root/reference): that part is "parity unpinned" (no golden vectors exist for it;
pyswarm uses numpy's unseeded global RNG so trajectories cannot be pinned at all).

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
import numpy as np


# ----------------------------------------------------------------------------------
# nmrfit/proc_autophase.py:9-36  ps2
# ----------------------------------------------------------------------------------
def ps2(u, v, p0=0.0, p1=0.0, inv=False):
    """Zeroth + first order phase rotation of u + i v (radians).

    proc_autophase.py:29  data = u + 1j*v
    proc_autophase.py:30  size = data.shape[-1]
    proc_autophase.py:31  apod = exp(1j*(p0 + (p1*arange(size)/size))).astype(data.dtype)
    proc_autophase.py:33  if inv: apod = 1/apod
    proc_autophase.py:35  data = apod*data ; return real, imag
    """
    data = u + 1j * v
    size = data.shape[-1]
    apod = np.exp(1.0j * (p0 + (p1 * np.arange(size) / size))).astype(data.dtype)
    if inv:
        apod = 1 / apod
    data = apod * data
    return data.real, data.imag


# ----------------------------------------------------------------------------------
# nmrfit/equations.py:115-149  voigt
# ----------------------------------------------------------------------------------
def voigt(w, r, yoff, width, loc, a):
    """Area-parameterised pseudo-Voigt.

    equations.py:141  L = (2/(pi*width)) * 1/(1 + ((w-loc)/(0.5*width))**2)
    equations.py:144  G = (2/width)*sqrt(ln2/pi)*exp(-((w-loc)/(width/(2*sqrt(ln2))))**2)
    equations.py:147  V = yoff + a*(r*L + (1-r)*G)
    """
    L = (2 / (np.pi * width)) * 1 / (1 + ((w - loc) / (0.5 * width)) ** 2)
    G = (2 / width) * np.sqrt(np.log(2) / np.pi) * np.exp(
        -((w - loc) / (width / (2 * np.sqrt(np.log(2))))) ** 2)
    V = yoff + a * (r * L + (1 - r) * G)
    return V


# ----------------------------------------------------------------------------------
# nmrfit/equations.py:152-212  objective   (fit_im=False branch only)
# ----------------------------------------------------------------------------------
def residual(x, w, u, v, weights):
    """weights*(V_data - V_fit): the vector whose RMS is the objective.

    equations.py:177      p0, p1, r, yoff = x[:4]
    equations.py:180-181  V_data, I_data = ps2(u, v, p0, p1); V_fit = zeros_like(V_data)
    equations.py:188-195  for i in range(4, len(x), 3): V_fit = V_fit + voigt(...)
    equations.py:202      multiply(weights, (V_data - V_fit))
    """
    p0, p1, r, yoff = x[:4]
    V_data, _ = ps2(u, v, p0=p0, p1=p1)
    V_fit = np.zeros_like(V_data)
    for i in range(4, len(x), 3):
        width = x[i]
        loc = x[i + 1]
        a = x[i + 2]
        V_fit = V_fit + voigt(w, r, yoff, width, loc, a)
    return np.multiply(weights, (V_data - V_fit))


def objective(x, w, u, v, weights, fit_im=False):
    """equations.py:202  rmse = sqrt(square(weights*(V_data-V_fit)).mean())."""
    if fit_im:
        # equations.py:197-199,205-209 (Kramers-Kronig by scipy quad) is out of scope
        # (SURVEY.md section 8 row a7 / f3).
        raise NotImplementedError("fit_im=True is outside the hot-path scope")
    return np.sqrt(np.square(residual(x, w, u, v, weights)).mean(axis=None))


def objective_batch(X, w, u, v, weights):
    """The reference's per-particle loop (pyswarm: fx[i] = obj(x[i, :]); utils.py:176)."""
    X = np.asarray(X, dtype=np.float64)
    return np.array([objective(X[i, :], w, u, v, weights) for i in range(X.shape[0])])


def residual_batch(X, w, u, v, weights):
    X = np.asarray(X, dtype=np.float64)
    return np.stack([residual(X[i, :], w, u, v, weights) for i in range(X.shape[0])])


# ----------------------------------------------------------------------------------
# nmrfit/equations.py:215-238  laplace1d  (in place, Jacobi sweeps, ends fixed)
# ----------------------------------------------------------------------------------
def laplace1d(x, n=10, omega=0.33333333):
    for _ in range(0, n):
        x[1:-1] = (1. - omega) * x[1:-1] + omega * 0.5 * (x[2:] + x[:-2])
    return x


# ----------------------------------------------------------------------------------
# nmrfit/utils.py:191-224  FitUtility._compute_weights
# ----------------------------------------------------------------------------------
def compute_weights(w, peaks, expon=0.5):
    """peaks: sequence of objects with .bounds (2 floats) and .height.

    utils.py:201-202  np.int index arrays (np.int no longer exists; int is the same type)
    utils.py:205-211  lIdx/rIdx = argmin|w - bounds|, swapped if reversed
    utils.py:213,215  maxabs = |height| ; biggest = amax(maxabs)
    utils.py:220-221  weights[l:r+1] = (biggest/maxabs_i)**expon (later peaks overwrite)
    utils.py:223      laplace1d(weights)
    """
    n = len(peaks)
    lIdx = np.zeros(n, dtype=int)
    rIdx = np.zeros(n, dtype=int)
    maxabs = np.zeros(n)
    for i, p in enumerate(peaks):
        lIdx[i] = np.argmin(np.abs(w - p.bounds[0]))
        rIdx[i] = np.argmin(np.abs(w - p.bounds[1]))
        if lIdx[i] > rIdx[i]:
            lIdx[i], rIdx[i] = rIdx[i], lIdx[i]
        maxabs[i] = np.abs(p.height)
    biggest = np.amax(maxabs)
    weights = np.ones(len(w)) * 1.0
    for i in range(n):
        weights[lIdx[i]:rIdx[i] + 1] = np.power(biggest / maxabs[i], expon)
    return laplace1d(weights)


# ----------------------------------------------------------------------------------
# pyswarm.pso (third party, absent): call site nmrfit/utils.py:176-182.
# Restated from the published algorithm (tisimst/pyswarm master, pso.py).  PARITY
# UNPINNED against pyswarm itself (its source is not in /root/reference).  ``rng``
# replaces numpy's global RNG and is the injection seam of tests/test_pso_cpu.py: any
# object with ``random(shape)`` and ``uniform(size=shape)``; the draw ORDER follows
# pyswarm (x, then v, then per iteration rp, rg), so a feed that returns supplied arrays
# in that order pins the product's swarm rule to this restatement bit for bit.
# ``full_output=True`` returns the whole final state beside (xopt, fopt).
# ----------------------------------------------------------------------------------
def pso(func, lb, ub, args=(), swarmsize=100, omega=0.5, phip=0.5, phig=0.5,
        maxiter=100, minstep=1e-8, minfunc=1e-8, rng=None, verbose=False, full_output=False):
    rng = np.random.default_rng() if rng is None else rng
    lb = np.array(lb, dtype=float)
    ub = np.array(ub, dtype=float)
    assert len(lb) == len(ub), 'Lower- and upper-bounds must be the same length'
    assert np.all(ub > lb), 'All upper-bound values must be greater than lower-bound values'
    vhigh = np.abs(ub - lb)
    vlow = -vhigh
    S, D = swarmsize, len(lb)

    def done(xopt, fopt, reason, it):
        if full_output:
            return xopt, fopt, dict(x=x, v=v, p=p, fx=fx, fp=fp, g=g, fg=fg, it=it, reason=reason)
        return xopt, fopt

    x = rng.random((S, D))
    fp = np.ones(S) * np.inf
    p = np.zeros_like(x)
    fg = np.inf
    x = lb + x * (ub - lb)
    fx = np.array([func(x[i, :], *args) for i in range(S)])
    i_update = fx < fp
    p[i_update, :] = x[i_update, :].copy()
    fp[i_update] = fx[i_update]
    i_min = np.argmin(fp)
    if fp[i_min] < fg:
        fg = fp[i_min]
        g = p[i_min, :].copy()
    else:
        g = x[0, :].copy()          # no particle has a finite objective yet: pyswarm starts from particle 0
    v = vlow + rng.random((S, D)) * (vhigh - vlow)

    it = 1
    while it <= maxiter:
        rp = rng.uniform(size=(S, D))
        rg = rng.uniform(size=(S, D))
        v = omega * v + phip * rp * (p - x) + phig * rg * (g - x)
        x = x + v
        maskl = x < lb
        masku = x > ub
        x = x * (~np.logical_or(maskl, masku)) + lb * maskl + ub * masku
        fx = np.array([func(x[i, :], *args) for i in range(S)])
        i_update = fx < fp
        p[i_update, :] = x[i_update, :].copy()
        fp[i_update] = fx[i_update]
        i_min = np.argmin(fp)
        if fp[i_min] < fg:
            p_min = p[i_min, :].copy()
            stepsize = np.sqrt(np.sum((g - p_min) ** 2))
            if np.abs(fg - fp[i_min]) <= minfunc:
                if verbose:
                    print('Stopping search: Swarm best objective change less than {:}'.format(minfunc))
                return done(p_min, fp[i_min], 'minfunc', it)
            elif stepsize <= minstep:
                if verbose:
                    print('Stopping search: Swarm best position change less than {:}'.format(minstep))
                return done(p_min, fp[i_min], 'minstep', it)
            else:
                g = p_min.copy()
                fg = fp[i_min]
        it += 1
    if verbose:
        print('Stopping search: maximum iterations reached --> {:}'.format(maxiter))
    return done(g, fg, 'maxiter', maxiter)


# ----------------------------------------------------------------------------------
# Kramers-Kronig path (fit_im=True and generate_result).  Restated as the reference
# computes it: adaptive quadrature per grid point.  Slow by construction (~4 ms/point).
# nmrfit/equations.py:9-49 kk_equation, :52-80 kk_relation, :242 kk_relation_vectorized
# ----------------------------------------------------------------------------------
def kk_equation(x, r, yoff, width, loc, a, w):
    """equations.py:37-49: [V(w - x) - V(w + x)] / x, the integrand with the singularity folded."""
    L1 = (2 / (np.pi * width)) * 1 / (1 + ((x + w - loc) / (0.5 * width)) ** 2)
    G1 = (2 / width) * np.sqrt(np.log(2) / np.pi) * np.exp(-((x + w - loc) / (width / (2 * np.sqrt(np.log(2))))) ** 2)
    V1 = yoff + a * (r * L1 + (1 - r) * G1)
    L2 = (2 / (np.pi * width)) * 1 / (1 + ((-x + w - loc) / (0.5 * width)) ** 2)
    G2 = (2 / width) * np.sqrt(np.log(2) / np.pi) * np.exp(-((-x + w - loc) / (width / (2 * np.sqrt(np.log(2))))) ** 2)
    V2 = yoff + a * (r * L2 + (1 - r) * G2)
    return 1 / x * (V2 - V1)


def kk_relation(w, r, yoff, width, loc, a):
    """equations.py:79-80: quad(kk_equation, 0, inf)/pi for one w."""
    import scipy.integrate
    res, _ = scipy.integrate.quad(kk_equation, 0, np.inf, args=(r, yoff, width, loc, a, w))
    return res / np.pi


kk_relation_vectorized = np.vectorize(kk_relation, otypes=[float])      # equations.py:242 (np.float -> float)


def objective_fit_im(x, w, u, v, weights):
    """equations.py:152-212 with fit_im=True, INCLUDING its quirk: I_fit is assigned inside
    the peak loop (equations.py:199), so only the last peak's imaginary line is compared."""
    p0, p1, r, yoff = x[:4]
    V_data, I_data = ps2(u, v, p0=p0, p1=p1)
    V_fit = np.zeros_like(V_data)
    I_fit = np.zeros_like(I_data)
    for i in range(4, len(x), 3):
        width, loc, a = x[i], x[i + 1], x[i + 2]
        V_fit = V_fit + voigt(w, r, yoff, width, loc, a)
        I_fit = kk_relation_vectorized(w, r, yoff, width, loc, a)
    rmse = np.sqrt(np.square(np.multiply(weights, (V_data - V_fit))).mean(axis=None))
    rmse += np.sqrt(np.square(np.multiply(weights, (I_data - I_fit))).mean(axis=None))
    rmse /= 2.0
    return rmse


def generate_result(params, w_data, u, v, scale=1):
    """FitUtility.generate_result (utils.py:226-295) as a function: returns a dict with
    w, V, I (phase-corrected data), real_contribs, imag_contribs, V_fit, I_fit, u_fit, v_fit."""
    if scale == 1.0:
        w = w_data
    else:
        w = np.linspace(w_data.min(), w_data.max(), int(scale * w_data.shape[0]))
    V_fit = np.zeros_like(w)
    I_fit = np.zeros_like(w)
    p0, p1, r, yoff = params[:4]
    res = params[4:]
    V, I = ps2(u, v, p0, p1)                       # containers.py:68-78 shift_phase('manual')
    real_contribs, imag_contribs = [], []
    for i in range(0, len(res), 3):
        width, loc, a = res[i], res[i + 1], res[i + 2]
        real = voigt(w, r, yoff, width, loc, a)
        imag = kk_relation_vectorized(w, r, yoff, width, loc, a)
        real_contribs.append(real)
        imag_contribs.append(imag)
        V_fit = V_fit + real
        I_fit = I_fit + imag
    u_fit, v_fit = ps2(V_fit, I_fit, inv=True, p0=p0, p1=p1)
    return dict(w=w, V=V, I=I, real_contribs=real_contribs, imag_contribs=imag_contribs,
                V_fit=V_fit, I_fit=I_fit, u_fit=u_fit, v_fit=v_fit)

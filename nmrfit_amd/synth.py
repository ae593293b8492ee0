"""
Synthetic spectra and swarms for tests and bench.py.

The reference ships no example data (its .gitignore excludes examples/), so every
measurement here runs on spectra built by this generator: a sum of pseudo-Voigt lines
(the model of nmrfit/equations.py:115-149) with Gaussian noise (modelled on
nmrfit/utils.py:856-875 ``rnd_data``), de-phased by a known (p0, p1) so that the
fitter has something to find, plus the parameter box the reference would build for it
(nmrfit/containers.py:193-217).

This is a data generator, not an evaluator: it is never used to compute objectives.
"""
from collections import namedtuple
import numpy as np

Workload = namedtuple("Workload", "name S N P")

# BASELINE.json configs -> shapes ("Configs")
CONFIGS = {
    "C1": Workload("C1", 50, 4096, 6),
    "C2": Workload("C2", 1024, 4096, 6),
    "C3": Workload("C3", 4096, 65536, 24),
    "C4": Workload("C4", 32768, 65536, 24),   # 4096 per GPU on 8 GPUs
    "C5": Workload("C5", 41, 16384, 12),      # D+1 rows of a forward-difference Jacobian
}


class SynthPeak:
    """Attribute bag with the fields FitUtility reads (utils.py:58-93, 205-213)."""

    def __init__(self, loc, height, width, area):
        self.loc = loc
        self.height = height
        self.width = width
        self.area = area
        self.bounds = [loc - 2.0 * width, loc + 2.0 * width]   # utils.py:741-774: loc +- 2 FWHM

    def __repr__(self):
        return "SynthPeak(loc=%g, height=%g, width=%g, area=%g)" % (
            self.loc, self.height, self.width, self.area)


class SynthData:
    """Duck-type of nmrfit.containers.Data as far as fit() needs it (w, u, v, peaks)."""

    def __init__(self, w, u, v, peaks):
        self.w, self.u, self.v = w, u, v
        self.V, self.I = u[:], v[:]
        self.peaks = peaks


def _lineshape(w, r, width, loc, a):
    t = (w - loc) * (2.0 / width)
    t2 = t * t
    lor = (2.0 / (np.pi * width)) / (1.0 + t2)
    gau = (2.0 / width) * np.sqrt(np.log(2.0) / np.pi) * np.exp2(-t2)
    return a * (r * lor + (1.0 - r) * gau)


def _dispersion(w, r, width, loc, a):
    """Imaginary (dispersive) partner of ``_lineshape``: its Hilbert transform, which is what
    the reference's Kramers-Kronig integral (nmrfit/equations.py:9-80) evaluates numerically.
    Lorentzian 1/(1+t^2) -> t/(1+t^2); Gaussian exp(-x^2) -> (2/sqrt(pi))*Dawson(x)."""
    from scipy.special import dawsn
    t = (w - loc) * (2.0 / width)
    lor = (2.0 / (np.pi * width)) * t / (1.0 + t * t)
    gau = (2.0 / width) * np.sqrt(np.log(2.0) / np.pi) * (2.0 / np.sqrt(np.pi)) * dawsn(np.sqrt(np.log(2.0)) * t)
    return a * (r * lor + (1.0 - r) * gau)


def make_spectrum(N, P, seed=1, noise=1e-3, w_lo=3.0, w_hi=4.0, physical=False):
    """Returns dict(w,u,v,weights,x_true,lower,upper,peaks) for a P-peak, N-point spectrum.

    physical=False (the BASELINE/SURVEY 8(d) workload): the imaginary channel is pure noise --
    enough for the objective, which never looks at it when fit_im=False, but it leaves the
    phase degenerate with the areas (rotating by D scales the real part by cos D).
    physical=True: the imaginary channel carries the dispersive line shape, as a spectrometer
    delivers it, so (p0, p1) are identifiable; used by the convergence tests."""
    rng = np.random.default_rng(seed)
    w = np.linspace(w_lo, w_hi, N)
    p0, p1, r, yoff = 0.3, -0.2, 0.6, 0.002
    widths = 0.004 + 0.002 * rng.random(P)
    locs = np.linspace(w_lo + 0.1 * (w_hi - w_lo), w_hi - 0.1 * (w_hi - w_lo), P)
    areas = 0.005 * (1.0 + rng.random(P))
    x_true = np.empty(4 + 3 * P)
    x_true[:4] = (p0, p1, r, yoff)
    x_true[4::3], x_true[5::3], x_true[6::3] = widths, locs, areas

    V = np.zeros(N)
    for k in range(P):
        V += yoff + _lineshape(w, r, widths[k], locs[k], areas[k])
    sigma = noise * np.max(V)
    Vn = V + sigma * rng.standard_normal(N)
    In = sigma * rng.standard_normal(N)
    if physical:
        for k in range(P):
            In += _dispersion(w, r, widths[k], locs[k], areas[k])
    # de-phase: (u + i v) = (V + i I) * exp(-i phi), phi_j = p0 + p1*j/N
    phi = p0 + p1 * np.arange(N) / N
    z = (Vn + 1j * In) * np.exp(-1j * phi)
    u, v = np.ascontiguousarray(z.real), np.ascontiguousarray(z.imag)
    weights = 1.0 + rng.random(N)

    lower = [-np.pi, -np.pi, 0.0, -0.01]
    upper = [np.pi, np.pi, 1.0, 0.01]
    peaks = []
    for k in range(P):
        # containers.py:212-215 with bounds = loc -+ 2*width
        lower += [0.5 * widths[k], locs[k] - 0.2 * widths[k], 0.5 * areas[k]]
        upper += [1.5 * widths[k], locs[k] + 0.2 * widths[k], 1.5 * areas[k]]
        height = _lineshape(np.array([locs[k]]), r, widths[k], locs[k], areas[k])[0]
        peaks.append(SynthPeak(locs[k], height, widths[k], areas[k]))
    return dict(w=w, u=u, v=v, weights=weights, x_true=x_true,
                lower=np.array(lower), upper=np.array(upper), peaks=peaks, sigma=sigma)


def make_swarm(lower, upper, S, seed=2, x_true=None):
    """X = lb + U(0,1)*(ub-lb), row 0 optionally replaced by the generating parameters."""
    rng = np.random.default_rng(seed)
    lower = np.asarray(lower, dtype=np.float64)
    upper = np.asarray(upper, dtype=np.float64)
    X = lower + rng.random((S, lower.size)) * (upper - lower)
    if x_true is not None and S > 0:
        X[0, :] = x_true
    return np.ascontiguousarray(X)


def make_dense_swarm(S, P, seed=5, w_lo=3.0, w_hi=4.0):
    """X[S, 4+3P] of BROAD OVERLAPPING lines -- widths 0.3 .. 0.8 of the spectral span, centres anywhere in
    its middle 60 % -- the opposite of make_spectrum's sparse narrow lines: no Gaussian window misses any
    chunk and no peak is far from any chunk, so every (particle, point, peak) unit is evaluated in full.
    The same construction as tests/test_gpu_parity.py::adversarial_case("overlapping_broad"); bench.py
    reports the kernel's rate on it beside the headline (`dense_spectrum`)."""
    rng = np.random.default_rng(seed)
    span = w_hi - w_lo
    X = np.empty((S, 4 + 3 * P))
    X[:, 0] = rng.uniform(-np.pi, np.pi, S)
    X[:, 1] = rng.uniform(-np.pi, np.pi, S)
    X[:, 2] = rng.uniform(0, 1, S)
    X[:, 3] = rng.uniform(-0.01, 0.01, S)
    X[:, 4::3] = rng.uniform(0.3 * span, 0.8 * span, (S, P))
    X[:, 5::3] = rng.uniform(w_lo + 0.2 * span, w_hi - 0.2 * span, (S, P))
    X[:, 6::3] = rng.uniform(0.001, 0.01, (S, P))
    return np.ascontiguousarray(X)


def make_workload(name, S=None, seed=1):
    """(spectrum dict, X[S,D]) for one of BASELINE.json's configs."""
    cfg = CONFIGS[name]
    S = cfg.S if S is None else S
    spec = make_spectrum(cfg.N, cfg.P, seed=seed)
    X = make_swarm(spec["lower"], spec["upper"], S, seed=seed + 1, x_true=spec["x_true"])
    return spec, X


def jacobian_rows(x, rel_step=1.4901161193847656e-08):
    """The D+1 rows a forward-difference Jacobian needs (config C5): x, x + h_i e_i."""
    x = np.asarray(x, dtype=np.float64)
    D = x.size
    h = rel_step * np.maximum(1.0, np.abs(x))
    rows = np.tile(x, (D + 1, 1))
    rows[np.arange(1, D + 1), np.arange(D)] += h
    return rows, h

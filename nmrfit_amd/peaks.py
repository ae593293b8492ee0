"""
Host mirror of the reference's non-interactive peak utilities (nmrfit/utils.py): ``Peak`` (:58-93),
``Peaks`` (:14-55), ``AutoPeakSelector`` (:670-783), the programmatic path of ``BoundsSelector``
(:342-442, ``supress=True``), ``find_peak`` (:819-853), ``rnd_data`` (:856-875) and
``sample_noise`` (:878-902).  They produce the objects the hot path consumes: ``data.peaks``
(bounds, height -> the weights of FitUtility._compute_weights; width, loc, area -> the swarm box of
Data.generate_solution_bounds).  Host code, once per dataset, as in the reference.

Out of scope: the matplotlib click selectors (BoundsSelector without ``supress``, PeakSelector) and
the ``plot`` methods.

``peakutils.baseline`` (third party, not in the reference tree, not installed; the reference pins
no version) is restated here from its published algorithm as ``baseline`` -- PARITY UNPINNED for
that function; AutoPeakSelector is therefore tested through properties on synthetic spectra, not
against reference output.  ``scipy.integrate.simps`` (utils.py:769) no longer exists in scipy;
``simpson`` is the same rule.
"""
import numpy as np
import scipy.integrate
import scipy.interpolate
import scipy.ndimage
import scipy.signal


class Peak:
    """One line of a spectrum: loc, height, bounds = loc -+ 2 FWHM, width (FWHM), area."""

    def __repr__(self):
        fields = (("Location", self.loc), ("Height", self.height),
                  ("Bounds", "[%s, %s]" % (self.bounds[0], self.bounds[1])), ("Width", self.width),
                  ("Area", self.area))
        return "\n".join("               %s: %s" % kv for kv in fields)


class Peaks(list):
    """A list of Peak objects with the reference's two helpers."""

    def average_height(self):
        return sum(abs(p.height) for p in self) / len(self)

    def split(self):
        """(peaks, satellites): at or above / below the average absolute height."""
        h = self.average_height()
        main, sats = Peaks(), Peaks()
        for p in self:
            (main if abs(p.height) >= h else sats).append(p)
        return main, sats


def baseline(y, deg=3, max_it=100, tol=1e-3):
    """peakutils.baseline restated: iterated least-squares polynomial fit where, after every
    fit, the data above the fit are clipped to it, until the coefficients move by less than
    ``tol`` (relative).  Returns the baseline array.  The abscissa is scaled to
    [0, max|y|**(1/(deg+1))] as in the published code, to keep the Vandermonde matrix tame."""
    y = np.array(y, dtype=float)
    order = deg + 1
    coeffs = np.ones(order)
    x = np.linspace(0.0, abs(y).max() ** (1.0 / order), y.size)
    vander = np.vander(x, order)
    pinv = np.linalg.pinv(vander)
    base = y.copy()
    for _ in range(max_it):
        new = pinv @ y
        if np.linalg.norm(new - coeffs) / np.linalg.norm(coeffs) < tol:
            break
        coeffs = new
        base = vander @ coeffs
        y = np.minimum(y, base)
    return base


def argrelmax(x, order):
    """Indices i with x[i] strictly greater than every x[i-order .. i+order] (ends clipped):
    exactly scipy.signal.argrelmax(x, order=order)[0], which the reference calls (utils.py:731),
    but by two sliding-window maxima, O(n) instead of O(n * order) -- on the reference's
    100x-upsampled grid `order` is ~10^4 points and the scipy form takes a minute."""
    x = np.asarray(x)
    order = int(order)
    if order < 1:
        raise ValueError("Order must be an int >= 1")
    n = x.size
    padded = np.pad(x, order, mode="edge")
    win = scipy.ndimage.maximum_filter1d(padded, size=order, mode="nearest")   # win[j] = max padded[j-order//2 .. +order-1]
    at = np.arange(n) + order
    left = win[at - order + order // 2]          # max of the `order` points before i
    right = win[at + 1 + order // 2]             # max of the `order` points after i
    return np.nonzero((x > left) & (x > right))[0]


class AutoPeakSelector:
    """Peak picking by local non-maximum suppression + FWHM analysis (utils.py:670-783).

    The spectrum is upsampled 100x by linear interpolation, smoothed (Savitzky-Golay, 11 points,
    order 4) for the maxima search; a maximum counts when its height above the constant baseline
    exceeds ``thresh``.  Width = distance between the half-height crossings nearest to the
    maximum, bounds = loc -+ 2 widths, area = Simpson integral over the bounds above a local
    constant baseline."""

    def __init__(self, w, u, thresh, window):
        self.thresh = thresh
        self.window = window
        w = np.asarray(w, dtype=float)
        interp = scipy.interpolate.interp1d(w, np.asarray(u, dtype=float))
        self.w = np.linspace(w.min(), w.max(), int(len(w) * 100))
        self.u = interp(self.w)
        self.u_smoothed = scipy.signal.savgol_filter(self.u, 11, 4)
        self.baseline = baseline(self.u_smoothed, 0)[0]
        self.peaks = Peaks()

    def find_maxima(self):
        order = int(self.window / (self.w[1] - self.w[0]))
        for i in argrelmax(self.u_smoothed, order):
            p = Peak()
            p.loc = self.w[i]
            p.i = i
            p.height = self.u[i] - self.baseline
            if p.height > self.thresh:
                self.peaks.append(p)

    def find_width(self):
        kept = Peaks()
        above = self.u - self.baseline
        for p in self.peaks:
            side = np.sign(p.height / 2.0 - above)
            cross = side[:-1] - side[1:]              # < 0 where the curve falls through half height
            falling = np.where(cross < 0)[0]
            rising = np.where(cross > 0)[0]
            if falling.size == 0 or rising.size == 0:
                continue                               # (the reference raises here; nothing to measure)
            x_right = self.w[falling[np.argmin(np.abs(self.w[falling] - p.loc))]]
            x_left = self.w[rising[np.argmin(np.abs(self.w[rising] - p.loc))]]
            if not x_left < x_right:
                continue
            p.width = x_right - x_left
            p.bounds = [p.loc - 2 * p.width, p.loc + 2 * p.width]
            p.idx = np.where((self.w >= p.bounds[0]) & (self.w <= p.bounds[1]))
            p.baseline = baseline(self.u[p.idx], 0)[0]
            p.height = self.u[p.i] - p.baseline
            p.area = scipy.integrate.simpson(self.u[p.idx] - p.baseline, x=self.w[p.idx])
            kept.append(p)
        self.peaks = kept

    def find_peaks(self):
        self.find_maxima()
        self.find_width()


class BoundsSelector:
    """Crop (w, u, v) to low < w < high (utils.py:416-442).  Only the programmatic form: the
    reference's interactive form (two mouse clicks on a matplotlib figure) is out of scope."""

    def __init__(self, w, u, v, supress=True):
        if not supress:
            raise NotImplementedError("interactive bounds selection is a GUI feature of the reference; "
                                      "pass low and high")
        self.w, self.u, self.v = np.asarray(w), np.asarray(u), np.asarray(v)
        self.supress = True

    def apply_bounds(self, low=None, high=None):
        if low is None or high is None:
            raise ValueError("low and high are required")
        keep = np.where((self.w > low) & (self.w < high))
        self.w, self.u, self.v = self.w[keep], self.u[keep], self.v[keep]
        return self.w, self.u, self.v


def find_peak(x, y, low, high):
    """(height, location, index within the window) of the maximum of y for low <= x <= high."""
    sel = np.where((x <= high) & (x >= low))
    k = np.argmax(y[sel])
    return y[sel][k], x[sel][k], k


def rnd_data(width, origdata):
    """origdata + width * standard normal noise (numpy's global generator, as in the reference)."""
    return origdata + width * np.random.randn(origdata.size)


def sample_noise(X, Y, xstart, xstop):
    """Standard deviation of Y about a quadratic fit, over xstart <= X <= xstop."""
    sel = np.where((X <= xstop) & (X >= xstart))
    xs, ys = X[sel], Y[sel]
    return np.std(ys - np.poly1d(np.polyfit(xs, ys, 2))(xs))

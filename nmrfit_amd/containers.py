"""
Host mirror of the reference's ``Data`` container (nmrfit/containers.py:8-252) for scripted,
non-interactive use: it carries (w, u, v), the phase estimate and the picked peaks from which
the swarm box and the weights of the hot path are derived.

    data = Data(w, u, v)
    data.shift_phase(method='auto')              # ACME estimate -> p0, p1, V, I
    data.select_bounds(low=3.0, high=4.0)        # crop
    data.select_peaks(method='auto', thresh=0.1, window=0.02)
    lower, upper = data.generate_solution_bounds()
    fit = nmrfit_amd.fit(data, lower, upper)

Kept from the reference: attribute names (w, u, v, V, I, p0, p1, peaks, roibounds), method names,
arguments, defaults and error messages.  Not kept: everything that opens a matplotlib window --
``plot=True``, ``select_bounds()`` without low/high and ``select_peaks(method='manual')`` raise
NotImplementedError (GUI selectors are out of scope).  ``nmrfit.load``
(Varian/Bruker files through nmrglue) is out of scope too: build a Data from arrays.
"""
import numpy as np

from . import peaks as _peaks
from . import proc_autophase
from . import utils


def _no_gui(what):
    raise NotImplementedError(what + " opens a matplotlib window in the reference; interactive selectors and "
                              "plots are out of scope here")


class Data:
    def __init__(self, w, u, v):
        self.w = w
        self.u = u
        self.v = v
        self.V = self.u[:]
        self.I = self.v[:]

    def shift_phase(self, method='auto', p0=0.0, p1=0.0, step=np.pi / 360, plot=False):
        """Set p0, p1 (radians) and the phase-corrected V, I (containers.py:51-96):
        'manual' takes p0, p1 as given, 'auto' minimises the ACME score, 'brute' scans p0."""
        choice = method.lower()
        if choice == 'manual':
            self.p0, self.p1 = p0, p1
        elif choice == 'auto':
            self.p0, self.p1 = proc_autophase.approximate_phase(self.u + 1j * self.v, 'acme')
        elif choice == 'brute':
            self.p0, self.p1 = self._brute_phase(step=step)
        else:
            raise ValueError("Method must be 'auto', 'brute', or 'manual'.")
        self.V, self.I = proc_autophase.ps2(self.u, self.v, self.p0, self.p1)
        if plot is True:
            _no_gui("shift_phase(plot=True)")

    def _brute_phase(self, step=np.pi / 360):
        """Scan p0 over [-pi, pi) (p1 = 0): keep the angle that best levels the two ends of the
        real part while the spectrum stays upright (containers.py:98-110).  Like the reference it
        leaves V, I at the LAST angle scanned; shift_phase recomputes them."""
        best_p0, best_err = 0, np.inf
        n = max(1, int(len(self.V) / 5000))
        for angle in np.arange(-np.pi, np.pi, step):
            self.V, self.I = proc_autophase.ps2(self.u, self.v, angle, 0.0)
            err = np.sqrt((self.V[:n].mean() - self.V[-n:].mean()) ** 2)
            if err < best_err and np.max(self.V) > abs(np.min(self.V)):
                best_p0, best_err = angle, err
        return best_p0, 0.0

    def select_bounds(self, low=None, high=None):
        """Crop w, u, v to low < w < high (containers.py:112-130).  V and I are NOT cropped, as in
        the reference: call shift_phase again afterwards."""
        if low is None or high is None:
            _no_gui("select_bounds() without low and high")
        self.w, self.u, self.v = _peaks.BoundsSelector(self.w, self.u, self.v, supress=True).apply_bounds(low=low, high=high)

    def select_peaks(self, method='auto', n=None, one_click=False, thresh=0.0, window=0.02, plot=False):
        """Pick peaks on (w, V) (containers.py:132-173); sets ``peaks`` and ``roibounds``."""
        choice = method.lower()
        if choice == 'manual':
            if isinstance(n, int) and n > 0:
                _no_gui("select_peaks(method='manual')")
            raise ValueError("Number of peaks must be specified when using 'manual' flag")
        if choice != 'auto':
            raise ValueError("Method must be 'auto' or 'manual'.")
        selector = _peaks.AutoPeakSelector(self.w, self.V, thresh=thresh, window=window)
        selector.find_peaks()
        if plot is True:
            _no_gui("select_peaks(plot=True)")
        self.peaks = selector.peaks
        self.roibounds = [p.bounds for p in self.peaks]

    def generate_solution_bounds(self, force_p0=False, force_p1=False):
        """(lower, upper) lists of the 4 + 3P parameter bounds (containers.py:175-217)."""
        return utils.generate_solution_bounds(self.peaks, p0=getattr(self, "p0", 0.0), p1=getattr(self, "p1", 0.0),
                                              force_p0=force_p0, force_p1=force_p1)

    def approximate_areas(self):
        return [p.area for p in self.peaks]

    def approximate_area_fraction(self):
        """Satellite share of the total approximate area: peaks below the mean area are satellites."""
        areas = np.array(self.approximate_areas())
        mean = np.mean(areas)
        main = areas[areas >= mean].sum()
        sats = areas[areas < mean].sum()
        return sats / (main + sats)

"""
Wall-clock guards, NON-GATING (a timing ratio inside a parity test can flake under
clock or thermal noise).  Every test here is `xfail(strict=False)`: a miss is reported as XFAIL,
a hit as XPASS, neither fails the suite.  The figures themselves are in DESIGN.md.
"""
import numpy as np
import pytest

from nmrfit_amd import _cabi
from test_gpu_parity import adversarial_case

pytestmark = pytest.mark.gpu


@pytest.mark.xfail(strict=False, reason="wall-clock guard, non-gating")
@pytest.mark.parametrize("case", ["overlapping_broad", "dense_cluster_1e12", "needles_everywhere"])
def test_farfield_is_not_slower_where_it_cannot_help(case):
    """FARFIELD on spectra without far peaks: no cliff against DEFAULT (kernel alone, HIP events,
    median of 30 launches after 20 warm-up launches)."""
    from nmrfit_amd import equations
    N, P, S, w, u, v, wt, X = adversarial_case(case)
    with equations.Evaluator(w, u, v, wt) as ev:
        dX, df = ev.dev_alloc(X.nbytes), ev.dev_alloc(8 * S)
        ev.upload(dX, X)
        ms = {}
        for name, vid in (("default", _cabi.VARIANT_DEFAULT), ("farfield", _cabi.VARIANT_FARFIELD)):
            ev.set_variant(vid)
            for _ in range(20):
                ev.objective_batch_dev(S, P, dX, df)
            ev.prof_enable(30)
            for _ in range(30):
                ev.objective_batch_dev(S, P, dX, df)
            ms[name] = float(np.median(ev.prof_read()[0]))
            ev.prof_enable(0)
        ev.dev_free(dX)
        ev.dev_free(df)
    print("farfield adversarial %s: default %.1f us, farfield %.1f us" % (case, ms["default"] * 1e3, ms["farfield"] * 1e3))
    assert ms["farfield"] <= 1.25 * ms["default"], ms

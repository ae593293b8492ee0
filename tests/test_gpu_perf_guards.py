"""
Wall-clock guards, NON-GATING (a timing ratio inside a parity test can flake under
clock or thermal noise).  Every test here is `xfail(strict=False)`: a miss is reported as XFAIL,
a hit as XPASS, neither fails the suite.  The figures themselves are in DESIGN.md.
"""
import numpy as np
import pytest

from nmrfit_amd import _cabi
from tests.test_gpu_parity import adversarial_case

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


@pytest.mark.xfail(strict=False, reason="wall-clock guard, non-gating")
@pytest.mark.parametrize("S,N,P,limit_us", [(204, 4096, 6, 13.0), (1024, 4096, 6, 22.0)])
def test_one_launch_generations_stay_fast(S, N, P, limit_us):
    """Per-generation wall time of nmrfit_pso_run with the stopping tests off: the reference's default swarm
    (round 4: 11.7 us with the fold deferred into the next launch's prologue; 13.4 with the last-ticket form, 15.6 in
    round 3) and C2 (19.6 us; round 3: 27.3)."""
    import time
    from nmrfit_amd import equations, pso, synth
    sp = synth.make_spectrum(N, P, seed=1)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
        sw.run(100, check_every=1000)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            sw.run(1000, check_every=1000)
            best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
        assert sw.last_launches() == 1
        sw.close()
    print("generation %d x %d x %d: %.2f us" % (S, N, P, best))
    assert best <= limit_us, best


@pytest.mark.xfail(strict=False, reason="wall-clock guard, non-gating")
def test_device_batched_default_fits_stay_fast():
    """40 default fits (204 x 4096 x 6) as one device batch, stopping rule off: round 5 measured 1.9-2.3 us per fit and
    generation (217-262 fits/s; four host threads with a context each, round 4: 94 fits/s = 5.3 us)."""
    import time
    from nmrfit_amd import synth
    from nmrfit_amd.batch import FitBatch
    K = 40
    specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(K)]
    with FitBatch([(s["w"], s["u"], s["v"], s["weights"]) for s in specs], [s["lower"] for s in specs],
                  [s["upper"] for s in specs], swarmsize=204, seeds=list(range(K)), minstep=-1.0, minfunc=-1.0) as fb:
        fb.run(50, 50)
        t0 = time.perf_counter()
        fb.run(500, 500)
        us = (time.perf_counter() - t0) / 500 / K * 1e6
    print("device batch of %d default fits: %.2f us per fit and generation" % (K, us))
    assert us <= 2.7, us


@pytest.mark.xfail(strict=False, reason="wall-clock guard, non-gating")
def test_farfield_kernel_with_the_all_peak_imaginary_model_stays_at_three_waves():
    """Round 6: the far-field kernel with fit_im="sum" at C3 (4096 x 65536 x 24): 3.03 ms up to round 5 (190 VGPRs, two
    waves per SIMD), 1.63-1.68 ms with the pair expansions and the quarter-interval Dawson table."""
    from nmrfit_amd import equations, synth
    sp, X = synth.make_workload("C3")
    S, D = X.shape
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        ev.set_variant(_cabi.VARIANT_FARFIELD)
        ev.set_fit_im("sum")
        dX, df = ev.dev_alloc(X.nbytes), ev.dev_alloc(8 * S)
        ev.upload(dX, X)
        for _ in range(30):
            ev.objective_batch_dev(S, (D - 4) // 3, dX, df)
        ev.prof_enable(10)
        for _ in range(10):
            ev.objective_batch_dev(S, (D - 4) // 3, dX, df)
        ms = float(np.median(ev.prof_read()[0]))
        ev.prof_enable(0)
        ev.dev_free(dX)
        ev.dev_free(df)
    print("far-field kernel, fit_im='sum', C3: %.3f ms" % ms)
    assert ms <= 1.9, ms


@pytest.mark.xfail(strict=False, reason="wall-clock guard, non-gating")
def test_readme_pipeline_keeps_up_with_the_fits():
    """Round 6: fit -> generate_result -> area fractions for 200 default spectra through fit_many(generate=True) against
    fit_many alone, pyswarm's rule on (best of two each): the bar of VERDICT r5 is 0.8 (measured 0.82-0.95)."""
    import contextlib
    import io
    import time
    import nmrfit_amd
    from nmrfit_amd import synth
    specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(8)]

    def jobs():
        return [dict(data=synth.SynthData(specs[k % 8]["w"], specs[k % 8]["u"], specs[k % 8]["v"], specs[k % 8]["peaks"]),
                     lower=list(specs[k % 8]["lower"]), upper=list(specs[k % 8]["upper"]), options={"seed": 7 + k}) for k in range(200)]
    best = {False: 1e9, True: 1e9}
    with contextlib.redirect_stdout(io.StringIO()):
        nmrfit_amd.fit_many(jobs()[:8], generate=True)
        for _ in range(2):
            for gen in (False, True):
                t0 = time.perf_counter()
                res = nmrfit_amd.fit_many(jobs(), generate=gen)
                if gen:
                    [f.calculate_area_fraction() for f in res]
                best[gen] = min(best[gen], time.perf_counter() - t0)
    print("200 default spectra: fit only %.0f /s, with the reconstruction %.0f /s (%.2f)" % (200 / best[False], 200 / best[True], best[False] / best[True]))
    assert best[False] / best[True] >= 0.8

"""
GPU tier: device-batched fits (nmrfit_batch_*, csrc/batch.hip + objective_batch.hip).  K independent fits advance
together, one launch per generation for all of them; the bar is that every fit's trajectory -- positions, velocities,
personal bests, (g, fg), the stopping generation and the returned (params, error) -- is BIT-IDENTICAL to what the same
fit does alone through nmrfit_pso_* / nmrfit_amd.fit (whose own trajectory is pinned against the restated pyswarm and
the numpy mirror in tests/test_gpu_pso.py), in both launch geometries.  Reference: the per-spectrum loop over
nmrfit.fit (nmrfit/core.py:64, README.md:64-66), 204 particles each (nmrfit/utils.py:177).
"""
import numpy as np
import pytest

from nmrfit_amd import _cabi, pso, synth
from nmrfit_amd.batch import FitBatch
from nmrfit_amd.equations import Evaluator

pytestmark = pytest.mark.gpu


def _problems(K, N=4096, peaks=(6, 4, 9, 1, 7, 6, 12, 3), seed0=40):
    out = []
    for k in range(K):
        sp = synth.make_spectrum(N, peaks[k % len(peaks)], seed=seed0 + k)
        out.append(sp)
    return out


def _lone_swarms(problems, S, seeds, variant, **kw):
    evs, sws = [], []
    for sp, seed in zip(problems, seeds):
        ev = Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"])
        ev.set_variant(_cabi.variant_id(variant))
        sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, **kw)
        evs.append(ev)
        sws.append(sw)
    return evs, sws


def _close(evs, sws):
    for sw in sws:
        sw.close()
    for ev in evs:
        ev.close()


def _batch(problems, S, seeds, variant="default", **kw):
    return FitBatch([(sp["w"], sp["u"], sp["v"], sp["weights"]) for sp in problems], [sp["lower"] for sp in problems],
                    [sp["upper"] for sp in problems], swarmsize=S, seeds=seeds, variant=variant, **kw)


@pytest.mark.parametrize("geometry", ["workgroup", "wave"])
@pytest.mark.parametrize("K,S,N", [(1, 204, 4096), (3, 204, 4096), (16, 204, 4096), (5, 50, 4096), (4, 203, 5000),
                                   (3, 64, 16384)])
def test_batch_trajectories_equal_lone_swarms_bit_for_bit(K, S, N, geometry):
    """State after generation 0, after 1, 2, 3 and after 40 generations (read at different phases of the buffer
    ping-pong), mixed peak counts, swarm sizes that are not multiples of the workgroup, a ragged grid: x, v, p, fp, fx,
    the iteration counters, fg and the best row of every fit equal the lone swarm's."""
    problems = _problems(K, N)
    seeds = [1000 + 7 * k for k in range(K)]
    kw = dict(minstep=-1.0, minfunc=-1.0)        # stopping rule off: every generation runs
    evs, sws = _lone_swarms(problems, S, seeds, "default", **kw)
    try:
        with _batch(problems, S, seeds, **kw) as fb:
            fb.set_geometry(geometry)
            assert fb.geometry()["mode"] == geometry
            fb.step()                                # generation 0
            for sw in sws:
                sw.init()
                sw.step()                            # (folds generation 0)
            done = 0
            for upto in (0, 1, 2, 3, 40):
                while done < upto:
                    fb.step()
                    for sw in sws:
                        sw.step()
                    done += 1
                for k, sw in enumerate(sws):
                    a, b = fb.state(k), sw.state()
                    for name in ("x", "v", "p", "fp", "fx"):
                        np.testing.assert_array_equal(a[name], b[name], err_msg="fit %d %s after %d" % (k, name, upto))
                st = fb.status()
                best = fb.best()
                for k, sw in enumerate(sws):
                    ls = sw.status()
                    assert (st[k]["iteration"], st[k]["stop"], st[k]["fg"]) == (ls["iteration"], ls["stop"], ls["fg"]), (k, upto)
                    xb, fbest = sw.best()
                    np.testing.assert_array_equal(best[k][0], xb)
                    assert best[k][1] == fbest
    finally:
        _close(evs, sws)


@pytest.mark.parametrize("geometry", ["workgroup", "wave"])
def test_batch_run_stops_every_fit_by_its_own_rule(geometry):
    """pyswarm's rule on (defaults 1e-8): the fits of a batch stop at different generations, a stopped fit is left alone
    while the others go on, and (params, error, stopping generation, reason) equal the lone nmrfit_pso_run's."""
    K, S = 6, 204
    problems = _problems(K)
    seeds = [5 + k for k in range(K)]
    evs, sws = _lone_swarms(problems, S, seeds, "default")
    try:
        lone = []
        for sw in sws:
            sw.run(600, 7)
            lone.append((sw.status(), sw.best()))
        with _batch(problems, S, seeds) as fb:
            fb.set_geometry(geometry)
            fb.run(600, 7)
            st, best = fb.status(), fb.best()
        stops = set()
        for k in range(K):
            ls, (xb, fbest) = lone[k]
            assert (st[k]["iteration"], st[k]["stop"], st[k]["fg"]) == (ls["iteration"], ls["stop"], ls["fg"]), k
            np.testing.assert_array_equal(best[k][0], xb)
            assert best[k][1] == fbest
            stops.add(ls["iteration"])
        assert len(stops) > 1, "the fits should not all stop in the same generation: %r" % stops
    finally:
        _close(evs, sws)


def test_batch_farfield_variant_and_per_fit_constants():
    """The far-field kernel (what fit() selects from grid x peaks = 1e5 on) and per-fit swarm constants / stopping
    thresholds."""
    K, S, N = 3, 96, 16384
    problems = _problems(K, N, peaks=(12, 9, 14))
    seeds = [77, 78, 79]
    omega, phip, phig = [-0.2134, 0.5, -0.1], [-0.3344, 0.5, 1.0], [2.3259, 0.5, 1.5]
    evs, sws = [], []
    try:
        for k, sp in enumerate(problems):
            ev = Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"])
            ev.set_variant(_cabi.VARIANT_FARFIELD)
            sws.append(pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seeds[k], omega=omega[k], phip=phip[k],
                                       phig=phig[k], minstep=1e-8, minfunc=[1e-8, 1e-5, -1.0][k]))
            evs.append(ev)
        for sw in sws:
            sw.run(120, 16)
        for geometry in ("workgroup", "wave"):
            with _batch(problems, S, seeds, variant="farfield", omega=omega, phip=phip, phig=phig,
                        minfunc=[1e-8, 1e-5, -1.0]) as fb:
                fb.set_geometry(geometry)
                fb.run(120, 16)
                st, best = fb.status(), fb.best()
            for k, sw in enumerate(sws):
                ls = sw.status()
                assert (st[k]["iteration"], st[k]["stop"]) == (ls["iteration"], ls["stop"]), (geometry, k)
                xb, fbest = sw.best()
                np.testing.assert_array_equal(best[k][0], xb)
                assert best[k][1] == fbest
    finally:
        _close(evs, sws)


@pytest.mark.parametrize("fit_im,variant,N,P", [(True, "default", 4096, (6, 3, 5)), ("sum", "default", 4096, (4, 6, 2)),
                                                  (True, "farfield", 16384, (12, 9, 10)), ("sum", "farfield", 16384, (12, 7, 33))])
def test_batch_with_the_imaginary_channel(fit_im, variant, N, P):
    """fit_im=True (the reference's last-peak-only imaginary term, nmrfit/equations.py:197-209) and "sum" in a device
    batch: still the lone swarms' trajectories, bit for bit, with pyswarm's rule on."""
    K, S = 3, 72
    problems = [synth.make_spectrum(N, P[k], seed=70 + k) for k in range(K)]
    seeds = [31, 32, 33]
    evs, sws = [], []
    try:
        for sp, seed in zip(problems, seeds):
            ev = Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"])
            ev.set_variant(_cabi.variant_id(variant))
            ev.set_fit_im(fit_im)
            evs.append(ev)
            sws.append(pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed))
        for sw in sws:
            sw.run(90, 8)
        with FitBatch([(sp["w"], sp["u"], sp["v"], sp["weights"]) for sp in problems], [sp["lower"] for sp in problems],
                      [sp["upper"] for sp in problems], swarmsize=S, seeds=seeds, variant=variant, fit_im=fit_im) as fb:
            assert fb.geometry()["mode"] == "wave"            # (the imaginary channel is batched in the wave = particle form)
            fb.run(90, 8)
            st, best = fb.status(), fb.best()
            for k, sw in enumerate(sws):
                a, b = fb.state(k), sw.state()
                for name in ("x", "v", "p", "fp"):
                    np.testing.assert_array_equal(a[name], b[name], err_msg="fit %d %s" % (k, name))
                ls = sw.status()
                assert (st[k]["iteration"], st[k]["stop"], st[k]["fg"]) == (ls["iteration"], ls["stop"], ls["fg"]), k
                xb, fbest = sw.best()
                np.testing.assert_array_equal(best[k][0], xb)
                assert best[k][1] == fbest
        with pytest.raises(_cabi.NmrfitError) as ei:      # (NOREC has no batched form)
            FitBatch([(sp["w"], sp["u"], sp["v"], sp["weights"]) for sp in problems], [sp["lower"] for sp in problems],
                     [sp["upper"] for sp in problems], swarmsize=S, seeds=seeds, variant="norec", fit_im="sum")
        assert ei.value.code == _cabi.E_UNSUPPORTED
    finally:
        _close(evs, sws)


def test_fit_many_batches_what_it_can_and_equals_the_plain_loop(capsys):
    """nmrfit_amd.fit_many: jobs of equal shape go through one device batch, the others (another grid length, fit_im)
    through fit(); every result equals the plain loop's bit for bit, in job order."""
    import nmrfit_amd
    jobs, opts = [], []
    for k, (N, P) in enumerate([(4096, 6), (4096, 4), (8192, 5), (4096, 7), (4096, 6), (8192, 3)]):
        sp = synth.make_spectrum(N, P, seed=60 + k)
        jobs.append((synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), list(sp["lower"]), list(sp["upper"])))
        opts.append({"seed": 300 + k, "maxiter": 80, "swarmsize": 120})
    loop = [nmrfit_amd.fit(*job, summary=False, options=o) for job, o in zip(jobs, opts)]
    dict_jobs = [dict(data=j[0], lower=j[1], upper=j[2], options=o) for j, o in zip(jobs, opts)]
    dict_jobs[3]["fit_im"] = True                     # a group of one: runs through fit()
    loop[3] = nmrfit_amd.fit(*jobs[3], fit_im=True, summary=False, options=opts[3])
    capsys.readouterr()
    many = nmrfit_amd.fit_many(dict_jobs, threads=2)
    out = capsys.readouterr().out
    assert out.count("Stopping search:") == len(jobs)
    for a, b in zip(many, loop):
        np.testing.assert_array_equal(a.params, b.params)
        assert a.error == b.error
        np.testing.assert_array_equal(a.weights, b.weights)
    # batch=False: the threaded path gives the same answers
    again = nmrfit_amd.fit_many(dict_jobs, threads=3, batch=False)
    for a, b in zip(again, loop):
        np.testing.assert_array_equal(a.params, b.params)


def test_batch_edge_shapes_two_parts_tiny_grid_no_peaks():
    """17 fits (two parts on two streams: 8 + 9), a 600-point grid (two chunks: no workgroup = particle form exists),
    three particles per swarm, peak counts 0, 1, 2, ...: still the lone swarms' trajectories, bit for bit."""
    K, S, N = 17, 3, 600
    problems = [synth.make_spectrum(N, k % 4, seed=90 + k) for k in range(K)]
    seeds = [2 ** 63 + 11 * k for k in range(K)]
    kw = dict(minstep=-1.0, minfunc=-1.0)
    evs, sws = _lone_swarms(problems, S, seeds, "default", **kw)
    try:
        with _batch(problems, S, seeds, **kw) as fb:
            assert fb.geometry()["mode"] == "wave"
            with pytest.raises(_cabi.NmrfitError) as ei:
                fb.set_geometry("workgroup")
            assert ei.value.code == _cabi.E_UNSUPPORTED
            fb.run(12, 5)
            for sw in sws:
                sw.run(12, 5)
            st, best = fb.status(), fb.best()
            for k, sw in enumerate(sws):
                a, b = fb.state(k), sw.state()
                for name in ("x", "v", "p", "fp", "fx"):
                    np.testing.assert_array_equal(a[name], b[name], err_msg="fit %d %s" % (k, name))
                ls = sw.status()
                assert (st[k]["iteration"], st[k]["stop"], st[k]["fg"]) == (ls["iteration"], ls["stop"], ls["fg"]), k
                xb, fbest = sw.best()
                np.testing.assert_array_equal(best[k][0], xb)
                assert best[k][1] == fbest
    finally:
        _close(evs, sws)


@pytest.mark.parametrize("geometry", ["workgroup", "wave"])
def test_batch_with_a_fit_that_never_has_a_finite_objective(geometry):
    """pyswarm seeds g with x[0] while no particle has a finite objective (tests/test_pso_cpu.py pins that rule against the
    restated loop): one fit of a batch whose spectrum is NaN (its fg stays +inf, its record carries x[0]) beside two
    ordinary ones -- each still equals its lone swarm, and the ordinary ones are not disturbed by their neighbour."""
    K, S, N = 3, 40, 4096
    problems = _problems(K, N)
    problems[0] = dict(problems[0], u=np.full(N, np.nan))
    seeds = [61, 62, 63]
    evs, sws = _lone_swarms(problems, S, seeds, "default")
    try:
        for sw in sws:
            sw.run(25, 4)
        with _batch(problems, S, seeds) as fb:
            fb.set_geometry(geometry)
            fb.run(25, 4)
            st, best = fb.status(), fb.best()
            for k, sw in enumerate(sws):
                a, b = fb.state(k), sw.state()
                for name in ("x", "v", "p", "fp", "fx"):
                    np.testing.assert_array_equal(a[name], b[name], err_msg="fit %d %s" % (k, name))
                ls = sw.status()
                assert st[k]["iteration"] == ls["iteration"] and st[k]["stop"] == ls["stop"], k
                assert (st[k]["fg"] == ls["fg"]) or (np.isinf(st[k]["fg"]) and np.isinf(ls["fg"])), k
                xb, fbest = sw.best()
                np.testing.assert_array_equal(best[k][0], xb)
                assert best[k][1] == fbest or (np.isinf(fbest) and np.isinf(best[k][1]))
        assert np.isinf(st[0]["fg"]) and np.isfinite(st[2]["fg"])
    finally:
        _close(evs, sws)


def test_full_length_default_fits_batched_equal_the_lone_fits():
    """The reference's default fit at full length -- 204 particles, 2000 generations (nmrfit/utils.py:177-178), stopping
    rule off so that every generation runs -- 18 spectra as one device batch (two parts, two streams, the wave geometry)
    against 18 lone nmrfit_amd.fit calls: 4e5 objective evaluations per fit, params and error equal bit for bit."""
    import contextlib
    import io
    import nmrfit_amd
    jobs = []
    for k in range(18):
        sp = synth.make_spectrum(4096, 6, seed=200 + k)
        jobs.append((synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), list(sp["lower"]), list(sp["upper"])))
    with contextlib.redirect_stdout(io.StringIO()):
        many = nmrfit_amd.fit_many([dict(data=j[0], lower=j[1], upper=j[2], options={"seed": 900 + k, "minstep": -1.0, "minfunc": -1.0})
                                    for k, j in enumerate(jobs)])
        lone = [nmrfit_amd.fit(*j, summary=False, options={"seed": 900 + k, "minstep": -1.0, "minfunc": -1.0}) for k, j in enumerate(jobs)]
    for a, b in zip(many, lone):
        np.testing.assert_array_equal(a.params, b.params)
        assert a.error == b.error


def test_fit_many_in_several_batches_equals_the_plain_loop(monkeypatch):
    """A job list longer than BATCH_JOBS: several device batches, the next one prepared on a second thread while the
    current one runs, leftovers of the spans batched together -- every fit bit-identical to the plain loop."""
    import nmrfit_amd
    from nmrfit_amd import core
    monkeypatch.setattr(core, "BATCH_JOBS", 5)
    lengths = [1024] * 4 + [1536] + [1024] * 3 + [1536, 2048] + [1024] * 3       # 13 jobs: spans of 5, 5, 3
    jobs = []
    for k, n in enumerate(lengths):
        sp = synth.make_spectrum(n, 2 + k % 3, seed=300 + k)
        jobs.append(dict(data=synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), lower=list(sp["lower"]),
                         upper=list(sp["upper"]), options={"seed": 50 + k, "swarmsize": 40, "maxiter": 60}))
    many = nmrfit_amd.fit_many(jobs)
    for k, job in enumerate(jobs):
        one = nmrfit_amd.fit(job["data"], job["lower"], job["upper"], summary=False, options=job["options"])
        assert np.array_equal(many[k].params, one.params) and many[k].error == one.error, k


def test_fit_many_over_the_devices_of_one_process():
    """fit_many(jobs, devices=[...]): a host thread per device, job k on devices[k % n], each share a device batch of
    its own -- rehearsed with the one card listed twice: same results as the plain call, in job order."""
    import nmrfit_amd
    jobs = []
    for k in range(9):
        sp = synth.make_spectrum(4096, 3 + k % 4, seed=120 + k)
        jobs.append(dict(data=synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), lower=list(sp["lower"]),
                         upper=list(sp["upper"]), options={"seed": 700 + k, "maxiter": 50, "swarmsize": 64}))
    want = nmrfit_amd.fit_many(jobs, generate=True)
    got = nmrfit_amd.fit_many(jobs, devices=[0, 0], generate=True)      # (two host threads, each with its own pipeline; the
    also = nmrfit_amd.fit_many(jobs, devices="all")                     # staged copies back share one pinned buffer pair per device)
    for a, b, c in zip(want, got, also):
        np.testing.assert_array_equal(a.params, b.params)
        np.testing.assert_array_equal(a.params, c.params)
        assert a.error == b.error == c.error
        np.testing.assert_array_equal(a.u, b.u)
        np.testing.assert_array_equal(np.stack(a.imag_contribs), np.stack(b.imag_contribs))
    with pytest.raises(ValueError):
        nmrfit_amd.fit_many(jobs, devices=[0], shard=True)
    with pytest.raises(ValueError):
        nmrfit_amd.fit_many(jobs, devices=[])


@pytest.mark.parametrize("fit_im,variant", [(False, "default"), (False, "farfield"), (True, "default"), ("sum", "default")])
def test_ragged_batch_trajectories_equal_lone_swarms_bit_for_bit(fit_im, variant):
    """Round 6: the fits of a batch may differ in grid length (spectra cropped per dataset,
    nmrfit/containers.py:112-130).  Lengths from one chunk to 13 (full and ragged last chunks, 1 to 4 chunks per block),
    mixed peak counts, two parts: every fit's state after 0, 1, 2 and 25 generations equals the lone swarm's."""
    lengths = [3000, 6000, 4096, 512, 700, 5555, 3333, 6144, 4500]
    K, S = len(lengths), 33
    problems = [synth.make_spectrum(n, 1 + (3 * k) % 7, seed=70 + k, physical=bool(fit_im)) for k, n in enumerate(lengths)]
    seeds = [2000 + 3 * k for k in range(K)]
    kw = dict(minstep=-1.0, minfunc=-1.0)
    evs, sws = _lone_swarms(problems, S, seeds, variant, **kw)
    try:
        for ev in evs:
            ev.set_fit_im(fit_im)
        with _batch(problems, S, seeds, variant=variant, fit_im=fit_im, **kw) as fb:
            assert fb.N is None and fb.geometry()["mode"] == "wave"
            with pytest.raises(_cabi.NmrfitError):
                fb.set_geometry("workgroup")          # (one launch geometry for all fits: the workgroup form needs equal lengths)
            fb.step()
            for sw in sws:
                sw.init()
                sw.step()
            done = 0
            for upto in (0, 1, 2, 25):
                while done < upto:
                    fb.step()
                    for sw in sws:
                        sw.step()
                    done += 1
                for k, sw in enumerate(sws):
                    a, b = fb.state(k), sw.state()
                    for name in ("x", "v", "p", "fp", "fx"):
                        np.testing.assert_array_equal(a[name], b[name], err_msg="fit %d (N = %d) %s after %d" % (k, lengths[k], name, upto))
            for k, ((x, f), sw) in enumerate(zip(fb.best(), sws)):
                xb, fb_ = sw.best()
                np.testing.assert_array_equal(x, xb)
                assert f == fb_
    finally:
        _close(evs, sws)


def test_batch_with_swarms_of_different_sizes():
    """Round 6: the swarms of a batch may differ in size (options['swarmsize'] is per fit, nmrfit/utils.py:177) -- with
    grids of different lengths on top: the launch has room for the largest swarm, the smaller ones' spare workgroups
    idle; every fit still follows its lone swarm bit for bit and stops by its own rule."""
    lengths = [4096, 3000, 700, 5000, 4096, 2048, 6000]
    sizes = [204, 17, 64, 100, 3, 260, 51]
    K = len(lengths)
    problems = [synth.make_spectrum(n, 1 + (2 * k) % 6, seed=170 + k) for k, n in enumerate(lengths)]
    seeds = [900 + k for k in range(K)]
    evs, sws = [], []
    try:
        for sp, seed, S in zip(problems, seeds, sizes):
            ev = Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"])
            evs.append(ev)
            sws.append(pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed))
        with FitBatch([(sp["w"], sp["u"], sp["v"], sp["weights"]) for sp in problems], [sp["lower"] for sp in problems],
                      [sp["upper"] for sp in problems], swarmsize=sizes, seeds=seeds) as fb:
            assert fb.S is None and fb.N is None and fb.geometry()["mode"] == "wave"
            fb.step()
            for sw in sws:
                sw.init()
                sw.step()
            for upto in range(1, 4):
                fb.step()
                for sw in sws:
                    sw.step()
                for k, sw in enumerate(sws):
                    a, b = fb.state(k), sw.state()
                    assert a["x"].shape == (sizes[k], len(problems[k]["lower"]))
                    for name in ("x", "v", "p", "fp", "fx"):
                        np.testing.assert_array_equal(a[name], b[name], err_msg="fit %d %s after %d" % (k, name, upto))
            fb.run(150, 16)
            for sw in sws:
                sw.run(150, 16)
            st = fb.status()
            for k, ((x, f), sw) in enumerate(zip(fb.best(), sws)):
                ls = sw.status()
                assert (st[k]["iteration"], st[k]["stop"]) == (ls["iteration"], ls["stop"]), k
                xb, fbest = sw.best()
                np.testing.assert_array_equal(x, xb)
                assert f == fbest
    finally:
        _close(evs, sws)


def test_fit_many_batches_spectra_of_different_lengths():
    """fit_many: jobs whose spectra differ in length share a device batch (the key no longer holds N); results equal the
    plain loop's bit for bit, with and without the reconstruction."""
    import nmrfit_amd
    rng = np.random.default_rng(3)
    lengths = [int(n) for n in rng.integers(3000, 6001, 14)]
    specs = [synth.make_spectrum(n, 2 + k % 5, seed=700 + k, physical=True) for k, n in enumerate(lengths)]

    def jobs():
        return [dict(data=synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), lower=list(sp["lower"]), upper=list(sp["upper"]),
                     options={"seed": 40 + k, "swarmsize": 60 + 7 * (k % 3), "maxiter": 90}) for k, sp in enumerate(specs)]
    from nmrfit_amd import utils
    keys = set()
    for j in jobs():
        f = utils.FitUtility(j["data"], j["lower"], j["upper"], options=j["options"])
        keys.add(f._batch_key(f._plan()))
    assert len(keys) == 1
    many = nmrfit_amd.fit_many(jobs(), generate=1.25)
    for k, job in enumerate(jobs()):
        one = nmrfit_amd.fit(job["data"], job["lower"], job["upper"], summary=False, options=job["options"])
        one.generate_result(1.25)
        assert np.array_equal(many[k].params, one.params) and many[k].error == one.error, k
        for name in ("u", "v", "V", "I", "w"):
            np.testing.assert_array_equal(getattr(many[k], name), getattr(one, name), err_msg="%d %s" % (k, name))
        np.testing.assert_array_equal(np.stack(many[k].imag_contribs), np.stack(one.imag_contribs))
        np.testing.assert_array_equal(many[k].data.V, one.data.V)
        assert many[k].w.shape == (int(1.25 * lengths[k]),)


def test_batch_argument_validation():
    sp = synth.make_spectrum(4096, 3, seed=1)
    spec = (sp["w"], sp["u"], sp["v"], sp["weights"])
    with pytest.raises(AssertionError):
        FitBatch([spec], [sp["upper"]], [sp["lower"]])
    with pytest.raises(ValueError):      # the four arrays of ONE spectrum differ in length (spectra may differ from each other)
        FitBatch([spec, (sp["w"][:100], sp["u"][:100], sp["v"][:99], sp["weights"][:100])], [sp["lower"]] * 2, [sp["upper"]] * 2)
    with pytest.raises(_cabi.NmrfitError) as ei:
        FitBatch([spec], [sp["lower"]], [sp["upper"]], variant="norec")
    assert ei.value.code == _cabi.E_UNSUPPORTED
    with pytest.raises(_cabi.NmrfitError):
        FitBatch([spec], [sp["lower"]], [sp["upper"]], device=1 << 20)
    with FitBatch([spec], [sp["lower"]], [sp["upper"]], swarmsize=16, seeds=[3]) as fb:
        with pytest.raises(_cabi.NmrfitError):
            fb.status()                                # before the first generation
        fb.run(0, 1)
        assert fb.status()[0]["iteration"] == 0


_SHARD_RANK = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import nmrfit_amd
from nmrfit_amd import synth

jobs = []
for k, (N, P) in enumerate([(4096, 6), (4096, 4), (4096, 7), (2048, 3), (4096, 5), (4096, 6), (4096, 2)]):
    sp = synth.make_spectrum(N, P, seed=80 + k)
    jobs.append(dict(data=synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), lower=list(sp["lower"]),
                     upper=list(sp["upper"]), options={"seed": 500 + k, "maxiter": 60, "swarmsize": 96}))
res = nmrfit_amd.fit_many(jobs, shard=%(shard)s)
print(json.dumps(dict(rank=int(os.environ.get("RANK", "0")), params=[list(map(float, r.params)) for r in res],
                      errors=[float(r.error) for r in res])))
"""


@pytest.mark.parametrize("world", [2, 3])
def test_fit_many_shards_jobs_over_ranks_on_one_gpu(world, tmp_path):
    """The spectra-parallel multi-GPU mode rehearsed on one card: `world` rank processes (each shown device 0 only, the
    way a launcher isolates ranks), fit_many(jobs, shard=True): rank r fits jobs r, r + world, ... as device batches
    of its own, the records are gathered over the rendezvous channel, and every rank returns all seven results -- equal
    bit for bit to the unsharded call's.  No collective touches the fits (replicas)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(_SHARD_RANK % dict(root=root, shard="True"))
    alone = tmp_path / "alone.py"
    alone.write_text(_SHARD_RANK % dict(root=root, shard="False"))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    ref = subprocess.run([sys.executable, str(alone)], env=env, capture_output=True, text=True, timeout=300)
    assert ref.returncode == 0, ref.stderr[-2000:]
    want = json.loads([l for l in ref.stdout.splitlines() if l.startswith("{")][-1])
    renv = dict(env, WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                NMRFIT_RDZV_TOKEN="shard%d" % os.getpid())
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(renv, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        got = json.loads([l for l in o.splitlines() if l.startswith("{")][-1])
        assert got["params"] == want["params"] and got["errors"] == want["errors"], got["rank"]

"""
GPU tier (iii/iv): the device-resident swarm (csrc/pso.hip through the C-ABI) against its
numpy mirror, the sharded form against the single swarm, and nmrfit_amd.fit() end to end.
"""
import numpy as np
import pytest

from nmrfit_amd import _cabi, pso, synth
from tests import swarm_support

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def problem():
    from nmrfit_amd import equations
    assert _cabi.device_count() >= 1
    sp = synth.make_spectrum(2048, 3, seed=5)
    ev = equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"])
    yield sp, ev
    ev.close()


@pytest.mark.parametrize("S", [100, 204, 512, 516, 700, 4099])   # 100 x 13: fused single-workgroup tail; larger: the many-workgroup select kernel
def test_device_swarm_matches_numpy_mirror_bitwise(problem, S):
    """Same Philox stream, same IEEE update arithmetic (no FMA contraction in the swarm
    kernels), same objective values (the mirror evaluates through the same GPU kernel with
    the same launch geometry) => x, v, p, fp, g identical bit for bit after 10 generations."""
    sp, ev = problem
    seed = 77
    dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
    host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
    dev.init()
    host.init()
    st = dev.state()
    np.testing.assert_array_equal(st["x"], host.x)
    np.testing.assert_array_equal(st["v"], host.v)
    np.testing.assert_array_equal(st["fx"], host.fx)
    np.testing.assert_array_equal(dev.candidate(), host.candidate())
    dev.apply_global(dev.candidate()[None, :])
    host.apply_global(host.candidate()[None, :])
    for _ in range(10):
        dev.step_local()
        host.step_local()
        np.testing.assert_array_equal(dev.candidate(), host.candidate())
        dev.apply_global(dev.candidate()[None, :])
        host.apply_global(host.candidate()[None, :])
    st = dev.state()
    for k in ("x", "v", "p", "fx", "fp"):
        np.testing.assert_array_equal(st[k], getattr(host, k), err_msg=k)
    xb, fb = dev.best()
    np.testing.assert_array_equal(xb, host.best_x)
    assert fb == host.best_f
    assert dev.status() == dict(iteration=10, stop=0, fg=host.fg)
    dev.close()


def test_sharded_device_swarms_equal_single(problem):
    """Three shards (uneven: 70/67/67) on one GPU, candidates gathered on the host."""
    sp, ev = problem
    S, seed = 204, 5
    one = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
    shards = []
    for r in range(3):
        off, n = pso.shard(S, r, 3)
        shards.append(pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, offset=off, S_local=n, seed=seed,
                                      minfunc=-1.0, minstep=-1.0))
    one.init()
    one.apply_global(one.candidate()[None, :])
    for s in shards:
        s.init()
    cands = np.stack([s.candidate() for s in shards])
    for s in shards:
        s.apply_global(cands)
    for _ in range(8):
        one.step_local()
        one.apply_global(one.candidate()[None, :])
        for s in shards:
            s.step_local()
        cands = np.stack([s.candidate() for s in shards])
        for s in shards:
            s.apply_global(cands)
    x1 = one.state()["x"]
    np.testing.assert_array_equal(np.concatenate([s.state()["x"] for s in shards]), x1)
    b1 = one.best()
    for s in shards:
        b = s.best()
        np.testing.assert_array_equal(b[0], b1[0])
        assert b[1] == b1[1]
        s.close()
    one.close()


def test_stop_flag_on_device_and_run_polling(problem):
    """nmrfit_pso_run polls the stop flag every check_every generations; generations after
    the stop are no-ops on the device, so the answer does not depend on check_every."""
    sp, ev = problem
    res = []
    for ce in (1, 7, 64):
        sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 64, seed=3)      # fused generation loop
        sw.run(400, check_every=ce)
        st = sw.status()
        res.append((st["stop"], st["iteration"], sw.best()))
        sw.close()
    assert res[0][0] in (1, 2)
    for r in res[1:]:
        assert r[0] == res[0][0] and r[1] == res[0][1]
        np.testing.assert_array_equal(r[2][0], res[0][2][0])
        assert r[2][1] == res[0][2][1]
    # and the mirror stops at the same generation with the same answer
    host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], 64, seed=3)
    xh, fh = pso.run_sharded(host, pso.LocalExchange(), 400)
    assert host.stop == res[0][0] and host.iteration == res[0][1]
    np.testing.assert_array_equal(xh, res[0][2][0])
    # the same for a swarm large enough to take the unfused loop
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 600, seed=4)
    sw.run(300, check_every=5)
    st = sw.status()
    host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], 600, seed=4)
    xh, fh = pso.run_sharded(host, pso.LocalExchange(), 300)
    assert (st["stop"], st["iteration"]) == (host.stop, host.iteration)
    xb, fb = sw.best()
    np.testing.assert_array_equal(xb, xh)
    assert fb == fh
    sw.close()


def test_bounds_validation(problem):
    sp, ev = problem
    with pytest.raises(AssertionError):
        pso.DeviceSwarm(ev, sp["upper"], sp["lower"], 8)


def test_fit_api_end_to_end(problem, capsys):
    """nmrfit_amd.fit with the reference signature on a synthetic spectrum: recovers the
    generating parameters to within the box scale and reaches the noise floor."""
    import nmrfit_amd
    from oracle import c_oracle
    sp = synth.make_spectrum(4096, 3, seed=9)
    data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
    res = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), expon=0.5, dynamic_weighting=True,
                         summary=True, options={"swarmsize": 204, "maxiter": 600, "seed": 1})
    out = capsys.readouterr().out
    assert "Fit Summary:" in out and "Stopping search:" in out
    assert res.params.shape == (13,) and np.isfinite(res.error)
    assert res.weights.shape == (4096,) and res.weights.min() >= 1.0
    # the returned error is the reference objective at the returned parameters
    ref = c_oracle.objective_batch(res.params, sp["w"], sp["u"], sp["v"], res.weights)[0]
    assert res.error == pytest.approx(ref, rel=1e-9)
    f_truth = c_oracle.objective_batch(sp["x_true"], sp["w"], sp["u"], sp["v"], res.weights)[0]
    assert res.error <= 1.5 * f_truth
    # pyswarm's minfunc rule stops long before the areas are pinned down (they trade off
    # against r and the widths); what is guaranteed is the box and the objective level
    assert (res.params >= sp["lower"]).all() and (res.params <= sp["upper"]).all()
    np.testing.assert_allclose(res.params[5::3], sp["x_true"][5::3], atol=2e-3)      # peak locations
    assert 0.0 <= res.calculate_area_fraction() <= 1.0
    # dynamic_weighting=False -> unit weights (utils.py:172-173)
    res2 = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), dynamic_weighting=False, summary=False,
                          options={"swarmsize": 64, "maxiter": 50, "seed": 2})
    assert (res2.weights == 1.0).all()
    with pytest.raises(ValueError):
        nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), fit_im="nonsense", summary=False)


def test_native_rccl_single_rank(problem):
    """The product exchange on real hardware as far as one GPU allows: a one-rank RCCL
    communicator created through the C-ABI (nmrfit_comm_*: dlopen'ed librccl, no torch).  The
    bookkeeping collectives work, a swarm with the communicator attached runs whole generations
    in one C call (ncclAllGather + fold on the kernels' stream) and reproduces the plain device
    loop bit for bit, and nmrfit_amd.fit(options={"exchange": "rccl"}) takes that same route."""
    import sys
    import nmrfit_amd
    sp, ev = problem
    ex = pso.RcclExchange(ev)
    info = ex.info()
    assert info["rank"] == 0 and info["world"] == 1 and info["rccl_version"] > 0
    ex.barrier()
    np.testing.assert_array_equal(ex.all_reduce([1.5, -2.0, 7.0], "max"), [1.5, -2.0, 7.0])
    np.testing.assert_array_equal(ex.all_reduce([1.5, -2.0], "sum"), [1.5, -2.0])
    assert ex.broadcast_seed(0xFEDCBA9876543210) == 0xFEDCBA9876543210
    rec = np.arange(14.0)
    np.testing.assert_array_equal(ex.gather_host(rec), rec[None, :])
    a = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 300, seed=9)
    xa, fa = pso.run_sharded(a, ex, 120, check_every=7)
    st = a.status()
    a.close()
    b = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 300, seed=9)
    b.run(120, check_every=7)
    xb, fb = b.best()
    assert b.status() == st, (b.status(), st)
    b.close()
    np.testing.assert_array_equal(xa, xb)
    assert fa == fb
    # step by step (what bench.py does), against the host-staged fold
    c = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 700, seed=3, minfunc=-1.0, minstep=-1.0)
    d = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 700, seed=3, minfunc=-1.0, minstep=-1.0)
    c.set_comm(ex)
    c.init()
    d.init()
    for _ in range(6):
        c.step()
    d.apply_global(d.candidate()[None, :])
    for _ in range(5):
        d.step_local()
        d.apply_global(d.candidate()[None, :])
    assert c.status() == d.status() and c.status()["iteration"] == 5
    np.testing.assert_array_equal(c.state()["x"], d.state()["x"])
    c.set_comm(None)
    c.close()
    d.close()
    ex.close()
    data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
    r1 = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False,
                        options={"swarmsize": 100, "maxiter": 40, "seed": 5, "exchange": "rccl"})
    r2 = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False,
                        options={"swarmsize": 100, "maxiter": 40, "seed": 5})
    np.testing.assert_array_equal(r1.params, r2.params)
    assert r1.error == r2.error
    # a READY RcclExchange as options['exchange'] (it used to fail in nmrfit_pso_set_comm, because fit()
    # works on its own context and a communicator belonged to the context it was made on): a communicator now serves
    # any context of its device -- its all-gather runs on the swarm's own stream -- so the caller's exchange is used
    # as it is, fit after fit, without another ncclCommInitRank; it stays usable afterwards
    ex2 = pso.RcclExchange(ev)
    for _ in range(3):
        r3 = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False,
                            options={"swarmsize": 100, "maxiter": 40, "seed": 5, "exchange": ex2})
        np.testing.assert_array_equal(r3.params, r2.params)
        assert r3.error == r2.error and r3._device() == ev.device
        assert ex2.handle.value and ex2.info()["world"] == 1       # (not closed, not replaced)
    ex2.barrier()
    ex2.close()
    with pytest.raises(ValueError, match="has been closed"):
        nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False,
                       options={"swarmsize": 100, "maxiter": 4, "seed": 5, "exchange": ex2})


def test_no_torch_in_the_product_path():
    """A fresh interpreter that fits with the RCCL exchange never imports torch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import nmrfit_amd\n"
        "from nmrfit_amd import synth\n"
        "sp = synth.make_spectrum(2048, 3, seed=5)\n"
        "data = synth.SynthData(sp['w'], sp['u'], sp['v'], sp['peaks'])\n"
        "r = nmrfit_amd.fit(data, list(sp['lower']), list(sp['upper']), summary=False,\n"
        "                   options={'swarmsize': 64, 'maxiter': 10, 'seed': 5, 'exchange': 'rccl'})\n"
        "assert 'torch' not in sys.modules, 'torch was imported'\n"
        "print('NO_TORCH_OK', r.error)\n" % root)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0 and "NO_TORCH_OK" in out.stdout, out.stderr[-3000:]


def test_c4_shape_eight_shards_equal_one_swarm():
    """BASELINE config C4 as far as one GPU allows: eight DeviceSwarm shards (4096 particles
    each, offsets q*4096 of a 32768-particle swarm, N = 65536, P = 24) on ONE device against
    the single 32768-particle swarm.  After 3 generations every shard holds the same (g, fg)
    as the single swarm, the shards' positions tile it bit for bit, and each shard's objective
    values equal the matching slice of one objective_batch over the whole swarm."""
    from nmrfit_amd import equations
    c4 = synth.CONFIGS["C4"]
    sp = synth.make_spectrum(c4.N, c4.P, seed=1)
    G, S = 8, c4.S
    per = S // G
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        kw = dict(seed=1234, minfunc=-1.0, minstep=-1.0)
        one = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, **kw)
        shards = [pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, offset=q * per, S_local=per, **kw) for q in range(G)]
        one.init()
        one.apply_global(one.candidate()[None, :])
        for s in shards:
            s.init()
        cands = np.stack([s.candidate() for s in shards])
        for s in shards:
            s.apply_global(cands)
        for _ in range(3):
            one.step_local()
            one.apply_global(one.candidate()[None, :])
            for s in shards:
                s.step_local()
            cands = np.stack([s.candidate() for s in shards])
            for s in shards:
                s.apply_global(cands)
        st1 = one.state()
        f_all = ev.objective_batch(st1["x"])
        np.testing.assert_array_equal(f_all, st1["fx"])          # swarm launch == plain batched launch
        b1 = one.best()
        for q, s in enumerate(shards):
            st = s.state()
            sl = slice(q * per, (q + 1) * per)
            for k in ("x", "v", "p", "fx", "fp"):
                np.testing.assert_array_equal(st[k], st1[k][sl], err_msg="shard %d %s" % (q, k))
            np.testing.assert_array_equal(st["fx"], f_all[sl])
            b = s.best()
            np.testing.assert_array_equal(b[0], b1[0])
            assert b[1] == b1[1]
            assert s.status() == one.status()
            s.close()
        assert one.status()["iteration"] == 3
        one.close()


def test_closing_the_evaluator_closes_its_swarms():
    from nmrfit_amd import equations
    sp = synth.make_spectrum(512, 1, seed=3)
    ev = equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"])
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 8, seed=1)
    sw.run(3)
    ev.close()                  # must not leave the swarm with a dangling context
    assert not sw._h.value
    sw.close()                  # idempotent
    del sw, ev


@pytest.mark.parametrize("S,P", [(12, 100), (12, 300), (300, 120), (3, 960), (1, 2), (2, 0)])
def test_device_swarm_wide_parameter_vectors(S, P):
    """Many-peak models (D = 4 + 3P up to 2884) and degenerate swarms, on both sides of the
    fused-tail threshold: still bit-identical to the numpy mirror."""
    from nmrfit_amd import equations
    sp = synth.make_spectrum(1024, P, seed=9)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=31, minfunc=-1.0, minstep=-1.0)
        host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=31, minfunc=-1.0, minstep=-1.0)
        dev.init(); host.init()
        for _ in range(4):
            np.testing.assert_array_equal(dev.candidate(), host.candidate())
            dev.apply_global(dev.candidate()[None, :]); host.apply_global(host.candidate()[None, :])
            dev.step_local(); host.step_local()
        st = dev.state()
        for k in ("x", "v", "p", "fx", "fp"):
            np.testing.assert_array_equal(st[k], getattr(host, k), err_msg=k)
        # the fused loop (nmrfit_pso_run) from the same start reaches the same state
        dev2 = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=31, minfunc=-1.0, minstep=-1.0)
        dev2.run(4, check_every=3)
        dev.apply_global(dev.candidate()[None, :])
        np.testing.assert_array_equal(dev2.state()["x"], dev.state()["x"])
        assert dev2.best()[1] == dev.best()[1]
        dev.close(); dev2.close()


def test_ticket_select_long_run(problem):
    """600 generations through nmrfit_pso_run at S = 512 (the last-ticket select kernel with its
    cross-workgroup hand-over, 128 workgroups) against the numpy mirror stepped on the host: any
    stale read of another workgroup's personal bests would show up as a diverging trajectory."""
    sp, ev = problem
    S, seed, gens = 512, 19, 600
    dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
    host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
    dev.run(gens, check_every=200)
    host.init()
    host.apply_global(host.candidate()[None, :])
    for _ in range(gens):
        host.step_local()
        host.apply_global(host.candidate()[None, :])
    st = dev.state()
    for k in ("p", "fp", "fx"):
        np.testing.assert_array_equal(st[k], getattr(host, k), err_msg=k)
    xb, fb = dev.best()
    np.testing.assert_array_equal(xb, host.best_x)
    assert fb == host.best_f and dev.status()["iteration"] == gens
    dev.close()


@pytest.mark.parametrize("mode", ["fused_pbest", "fast", "fenced", "two_launch"])
@pytest.mark.parametrize("S,N,P,variant,fit_im", [
    (204, 4096, 6, "default", False),      # the reference's default swarm: 51 select workgroups
    (1024, 4096, 6, "default", False),     # C2: 256 workgroups, the largest swarm that hands over inside one launch
    (1000, 2048, 3, "default", False),     # ragged last workgroup
    (130, 700, 5, "default", False),       # ragged grid, just above the single-workgroup tail
    (204, 4096, 6, "farfield", False),
    (50, 4096, 6, "default", False),       # eight segments per particle = ONE eight-wave workgroup (round 4): update,
    (300, 4096, 6, "norec", False),        # evaluation, f and personal best in the objective launch ("fused_pbest")
    (204, 16384, 12, "farfield", False),
    (120, 4096, 3, "default", True),       # the reference's fit_im=True
    (120, 4096, 3, "default", "sum"),
    (1024, 4096, 3, "default", True),      # four segments per particle: f written by the workgroup, with the
    (1024, 4096, 3, "farfield", "sum"),    # imaginary channel's second sum
])
def test_handover_modes_match_numpy_mirror(S, N, P, variant, fit_im, mode):
    """The generation's second half in every form -- personal bests inside the objective launch where a
    workgroup holds a whole particle ("fused_pbest": the default; the 512- and 1024-particle shapes here), or
    the select kernel with its cross-workgroup hand-over in its three forms (nmrfit_pso_set_handover:
    fence-free agent-scope stores, release / acquire fences, reduction as its own launch; fused personal
    bests switched off so that the select kernel runs at every shape) -- against the numpy mirror: same
    Philox stream, same update arithmetic, same summation order => every state array bit-identical,
    whatever the polling interval."""
    from nmrfit_amd import equations
    sp = synth.make_spectrum(N, P, seed=11)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        ev.set_variant(_cabi.variant_id(variant))
        ev.set_fit_im(fit_im)
        gens = 40
        host = swarm_support.HostSwarm(lambda X: ev.objective_batch(X, fit_im=fit_im), sp["lower"], sp["upper"], S, seed=23,
                             minfunc=-1.0, minstep=-1.0)
        host.init()
        host.apply_global(host.candidate()[None, :])
        for _ in range(gens):
            host.step_local()
            host.apply_global(host.candidate()[None, :])
        for ce in (gens, 7):
            dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=23, minfunc=-1.0, minstep=-1.0)
            if mode != "fused_pbest":
                dev.set_fused_pbest(False)
                dev.set_handover(mode)
            dev.run(gens, check_every=ce)
            if (S, N, P) in ((204, 4096, 6), (50, 4096, 6), (204, 16384, 12)) and fit_im is False:
                assert ev.last_launch()["waves_per_workgroup"] == 8 and ev.last_launch()["segments"] == 8
            st = dev.state()
            for k in ("x", "v", "p", "fx", "fp"):
                np.testing.assert_array_equal(st[k], getattr(host, k), err_msg="%s (check_every=%d)" % (k, ce))
            xb, fb = dev.best()
            np.testing.assert_array_equal(xb, host.best_x)
            assert fb == host.best_f
            assert dev.status() == dict(iteration=gens, stop=0, fg=host.fg)
            np.testing.assert_array_equal(dev.candidate(), host.candidate())
            dev.close()
    with pytest.raises(_cabi.NmrfitError):
        with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
            pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 8).set_handover(7)


@pytest.mark.parametrize("mode", ["fused_pbest", "fast", "fenced", "two_launch"])
def test_handover_modes_stop_rule(mode):
    """With the stopping tests armed every hand-over form stops at the same generation with the
    same answer as the numpy mirror, and later launches are no-ops."""
    from nmrfit_amd import equations
    sp = synth.make_spectrum(4096, 6, seed=1)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], 204, seed=8)
        xh, fh = pso.run_sharded(host, pso.LocalExchange(), 2000)
        assert host.stop in (1, 2) and host.iteration < 2000
        for ce in (64, 5):
            dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 204, seed=8)
            if mode != "fused_pbest":
                dev.set_fused_pbest(False)
                dev.set_handover(mode)
            dev.run(2000, check_every=ce)
            st = dev.status()
            assert (st["stop"], st["iteration"]) == (host.stop, host.iteration)
            xb, fb = dev.best()
            np.testing.assert_array_equal(xb, xh)
            assert fb == fh
            before = dev.state()
            dev.run(50, check_every=50)          # stopped: nothing moves
            after = dev.state()
            for k in before:
                np.testing.assert_array_equal(before[k], after[k])
            dev.close()


@pytest.mark.parametrize("S,N,P", [(204, 4096, 6), (256, 2048, 3), (512, 4096, 6), (1024, 4096, 6)])
def test_handover_stress_short(S, N, P):
    """The looped, in-suite form of tools/handover_stress.py (whose 1e7-exchange log is under
    profiles/r03/): 6 seeds x 500 generations with the fence-free hand-over against the same
    swarm run with the reduction as its own launch (no hand-over inside a launch at all), from the
    same seed, every state array compared bit for bit.  Value, index and personal-best rows must
    each be COMPLETE in memory before the ticket is drawn -- on gfx950 neither a barrier nor a
    workgroup-scope fence waits for global stores -- and any stale or torn read of another
    workgroup's post bends the trajectory, which never heals.  (About 70 % of the particles improve
    their personal best in every one of these generations -- numpy mirror, 256 particles: 170-195 per
    generation over the first 700 -- so every hand-over carries fresh rows.)"""
    from nmrfit_amd import equations
    sp = synth.make_spectrum(N, P, seed=4)
    gens = 500
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        for seed in range(99, 105):
            a = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
            b = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
            c = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
            d = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
            a.set_handover("fast")            # (opt-in since round 5: the default hands nothing over inside a launch)
            a.set_fused_pbest(False)          # a, b, d: the select kernel runs at every shape (fence-free / two launches /
            b.set_fused_pbest(False)          # the textbook release-acquire form the fence-free one is an optimisation of:
            b.set_handover("two_launch")      # ADVICE r3 -- a driver or toolchain change that broke FAST's ordering
            d.set_fused_pbest(False)          # would show here as a difference between a and d)
            d.set_handover("fenced")
            for sw in (a, b, c, d):           # c: the defaults (a workgroup is a particle: the whole generation in the objective launch, fold deferred)
                sw.run(gens, check_every=250)
            assert c.last_launches() == 1 and a.last_launches() == 2   # (defaults: the deferred fold, up to 1024 particles)
            sa, sb, sc, sd = a.state(), b.state(), c.state(), d.state()
            for k in ("x", "v", "p", "fx", "fp"):
                np.testing.assert_array_equal(sa[k], sb[k], err_msg="%s (seed %d)" % (k, seed))
                np.testing.assert_array_equal(sc[k], sb[k], err_msg="%s (seed %d, defaults)" % (k, seed))
                np.testing.assert_array_equal(sa[k], sd[k], err_msg="%s (seed %d, fast vs fenced)" % (k, seed))
            assert a.status() == b.status() == c.status() == d.status() and a.status()["iteration"] == gens
            np.testing.assert_array_equal(a.best()[0], b.best()[0])
            np.testing.assert_array_equal(c.best()[0], b.best()[0])
            np.testing.assert_array_equal(d.best()[0], b.best()[0])
            for sw in (a, b, c, d):
                sw.close()


def test_communicator_guards(problem):
    """A communicator serves swarms of ANY context of its device (the all-gather runs on the swarm's own stream:
    one ncclCommInitRank for fit after fit), one swarm at a time; nmrfit_comm_destroy refuses while a swarm still
    points at the communicator."""
    from nmrfit_amd import equations
    sp, ev = problem
    ex = pso.RcclExchange(ev)
    assert "HIP device 0" in ex.describe() and "rank 0 of 1" in ex.describe()
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 16, seed=1)
    sw.set_comm(ex)
    sw.run(3)
    want = sw.state()
    sw.set_comm(None)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev2:
        other = pso.DeviceSwarm(ev2, sp["lower"], sp["upper"], 16, seed=1)
        other.set_comm(ex)                     # another context, the same device
        # ONE swarm at a time: a second swarm is refused while `other` holds the communicator -- two
        # swarms' all-gathers, issued from two host threads, would pair up differently on different ranks
        with pytest.raises(_cabi.NmrfitError) as e:
            sw.set_comm(ex)
        assert e.value.code == _cabi.E_STATE and "one swarm at a time" in str(e.value)
        other.set_comm(ex)                     # (re-attaching the holder is a no-op)
        other.run(3)
        got = other.state()
        for k in want:
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        with pytest.raises(_cabi.NmrfitError) as e:
            ex.close()
        assert e.value.code == _cabi.E_STATE
        other.close()       # detaches
    sw.close()
    ex.close()


_FIT_RANK = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import nmrfit_amd
from nmrfit_amd import pso, synth
from tests import swarm_support
sp = synth.make_spectrum(4096, 6, seed=21)
data = synth.SynthData(sp['w'], sp['u'], sp['v'], sp['peaks'])
opts = {"swarmsize": 203, "maxiter": 80, "exchange": swarm_support.SocketExchange()}       # UNSEEDED: rank 0's seed must win
opts.update(%(extra)r)
res = nmrfit_amd.fit(data, list(sp['lower']), list(sp['upper']), summary=False, options=opts)
res.generate_result()
json.dump({"params": [float(v).hex() for v in res.params], "error": float(res.error).hex(), "seed": int(res.seed),
           "device": res._device(), "V0": float(res.V[0]).hex()},
          open(os.path.join(%(out)r, "fit_rank%%d.json" %% int(os.environ["RANK"])), "w"))
"""


@pytest.mark.parametrize("world,extra", [(2, {"device": 0}), (3, {"device": 0, "variant": "default"})])
def test_fit_api_with_several_ranks_on_one_gpu(tmp_path, world, extra):
    """nmrfit_amd.fit() itself with world > 1 -- seed broadcast, shard,
    DeviceSwarm(offset, S_local), run_sharded -- on hardware, as far as one GPU allows: `world`
    processes share the card, each calls fit(options={"exchange": SocketExchange(), "device": 0})
    UNSEEDED (the record staged through the host: RCCL refuses two ranks on one device).  Every rank
    returns the same params / error bit for bit, and they equal a single-rank fit given the seed
    rank 0 broadcast (the swarm is the same whatever the sharding: 203 particles split 102 + 101 or
    68 + 68 + 67)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import nmrfit_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    code = _FIT_RANK % dict(root=root, extra=extra, out=str(tmp_path))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NMRFIT_RDZV_TOKEN="fit%d_%d" % (os.getpid(), port), NMRFIT_RDZV_TIMEOUT="120")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    got = [json.load(open(os.path.join(str(tmp_path), "fit_rank%d.json" % r))) for r in range(world)]
    for g in got[1:]:
        assert g["params"] == got[0]["params"] and g["error"] == got[0]["error"] and g["seed"] == got[0]["seed"]
        assert g["V0"] == got[0]["V0"]
    assert all(g["device"] == 0 for g in got)
    sp = synth.make_spectrum(4096, 6, seed=21)
    data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
    opts = {"swarmsize": 203, "maxiter": 80, "seed": got[0]["seed"]}
    if "variant" in extra:
        opts["variant"] = extra["variant"]
    one = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False, options=opts)
    assert [float(v).hex() for v in one.params] == got[0]["params"]
    assert float(one.error).hex() == got[0]["error"]



class _PhiloxFeed:
    """oracle.pso's ``rng`` seam fed with the draws the device swarm consumes (tests/test_pso_cpu.py::PhiloxFeed)."""

    def __init__(self, seed, S, D):
        self.seed, self.S, self.D, self.gen = seed, S, D, 0
        self._init = list(pso.uniform2(seed, 0, S, D, 0))
        self._pair = []

    def random(self, shape):
        return self._init.pop(0)

    def uniform(self, size):
        if not self._pair:
            self.gen += 1
            self._pair = list(pso.uniform2(self.seed, self.gen, self.S, self.D, 0))
        return self._pair.pop(0)


@pytest.mark.parametrize("S,maxiter,thresholds", [(64, 400, {}), (204, 300, {}), (512, 40, dict(minfunc=-1.0, minstep=-1.0)),
                                                  (1500, 25, dict(minfunc=-1.0, minstep=-1.0))])
def test_device_swarm_equals_the_restated_pyswarm_bit_for_bit(problem, S, maxiter, thresholds):
    """VERDICT r3 item 3, on the device: nmrfit_pso_run (the loop fit() runs) against the restated pyswarm loop
    (oracle.pso; call site nmrfit/utils.py:176-182) fed the same Philox draws, with pyswarm's own calling
    convention for the objective -- ONE particle per call, through the scalar shim -- which gives the batch's
    values bit for bit because f does not depend on the launch geometry.  Same stop generation, same reason,
    same returned (x, f), same final positions and personal bests.  Every select path: one-workgroup tail
    (64), the whole generation in the objective launch with its fold deferred (204, 512), two launches (1500)."""
    from oracle import nmrfit_oracle as onp
    sp, ev = problem
    seed = 4242 + S
    D = len(sp["lower"])
    kw = dict(omega=pso.DEFAULTS["omega"], phip=pso.DEFAULTS["phip"], phig=pso.DEFAULTS["phig"], minstep=1e-8, minfunc=1e-8)
    kw.update(thresholds)
    def one(x):                 # pyswarm's calling convention: one particle per call (the scalar shim's path)
        return float(ev.objective_batch(x[None, :])[0])

    def rows_of_one_matrix(xrow):
        """The restatement evaluates `[func(x[i, :]) for i in range(S)]`: rows of ONE matrix, in order.  For the
        larger swarms the S calls of a generation are served from one batched launch over that matrix (found
        through the row view's .base) -- the same values bit for bit, checked below -- to keep the test short."""
        base = xrow.base if xrow.base is not None else xrow
        if getattr(rows_of_one_matrix, "tag", None) is not base:
            rows_of_one_matrix.tag = base
            rows_of_one_matrix.f = ev.objective_batch(np.ascontiguousarray(base))
            rows_of_one_matrix.i = 0
        v = rows_of_one_matrix.f[rows_of_one_matrix.i]
        rows_of_one_matrix.i += 1
        return float(v)
    X0 = synth.make_swarm(sp["lower"], sp["upper"], 9, seed=3)
    np.testing.assert_array_equal(ev.objective_batch(X0), np.array([one(x) for x in X0]))
    np.testing.assert_array_equal(np.array([rows_of_one_matrix(X0[i, :]) for i in range(9)]), ev.objective_batch(X0))
    xo, fo, st = onp.pso(one if S <= 64 else rows_of_one_matrix, sp["lower"], sp["upper"], swarmsize=S, maxiter=maxiter,
                         rng=_PhiloxFeed(seed, S, D), full_output=True, **kw)
    dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=seed, **kw)
    dev.run(maxiter, check_every=7)
    status = dev.status()
    xb, fb = dev.best()
    state = dev.state()
    dev.close()
    assert {"minfunc": 1, "minstep": 2, "maxiter": 0}[st["reason"]] == status["stop"]
    assert status["iteration"] == st["it"] and status["fg"] == st["fg"]
    np.testing.assert_array_equal(xb, xo)
    assert fb == fo
    for k in ("x", "v", "p", "fx", "fp"):
        np.testing.assert_array_equal(state[k], st[k], err_msg=k)
    if not thresholds:
        assert st["reason"] in ("minfunc", "minstep")      # pyswarm's defaults do stop these searches


@pytest.mark.parametrize("S", [50, 204, 512, 1500])
def test_no_finite_objective_yet_seeds_g_with_particle_zero(S):
    """pyswarm's `else: g = x[0, :]` after the first evaluation (no particle has a finite objective): the device
    swarm, its numpy mirror and the restated loop agree, on every select path, while the objective is NaN (NaN
    weights) for generations 0 and 1 and real afterwards."""
    from nmrfit_amd import equations
    sp = synth.make_spectrum(4096, 6, seed=9)
    D = len(sp["lower"])
    nan_w = np.full_like(sp["weights"], np.nan)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], nan_w) as ev:
        kw = dict(minfunc=-1.0, minstep=-1.0)
        dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=31, **kw)
        host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=31, **kw)
        dev.init()
        host.init()
        x0 = host.x[0].copy()
        for sw in (dev, host):
            sw.apply_global(sw.candidate()[None, :])
        c = dev.candidate()
        assert np.isinf(c[0]) and c[0] > 0
        np.testing.assert_array_equal(c[1:], x0)             # the record carries x[0], not the zero row p[0]
        np.testing.assert_array_equal(dev.best()[0], x0)
        for gen in range(1, 7):
            if gen == 2:
                ev.set_weights(sp["weights"])
            for sw in (dev, host):
                sw.step_local()
            np.testing.assert_array_equal(dev.candidate(), host.candidate())
            for sw in (dev, host):
                sw.apply_global(sw.candidate()[None, :])
            if gen == 1:
                assert np.isinf(dev.status()["fg"])
                np.testing.assert_array_equal(dev.candidate()[1:], host.x[0])
        st = dev.state()
        for k in ("x", "v", "p", "fx", "fp"):
            np.testing.assert_array_equal(st[k], getattr(host, k), err_msg=k)
        assert np.isfinite(dev.status()["fg"]) and dev.status()["fg"] == host.fg
        dev.close()
        # the fused loop (nmrfit_pso_step: fold inside the select launch) takes the same branch
        ev.set_weights(nan_w)
        dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=31, **kw)
        dev.run(3, check_every=1)
        xb, fb = dev.best()
        assert np.isinf(fb) and dev.status()["iteration"] == 3
        np.testing.assert_array_equal(xb, x0)
        dev.close()


@pytest.mark.parametrize("P", [130, 132, 133, 440, 452, 460, 598, 606, 700])
@pytest.mark.parametrize("fit_im", [False, True, "sum"])
def test_lds_budget_on_both_sides_of_every_threshold(P, fit_im):
    """the LDS budget that picks the kernel variant (scaled records above ~450 peaks, far-field
    scratch above ~600) counts the kernel's static LDS, the Dawson table of the imaginary channel and -- in a
    swarm generation with D <= 400 (P <= 132) -- the per-wave copies of the updated rows.  On either side of each
    threshold a swarm generation runs (no HIP launch error) and equals the numpy mirror, with and without the
    imaginary channel, for the direct and the far-field kernel."""
    from nmrfit_amd import equations
    sp = synth.make_spectrum(1024, P, seed=9)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        for variant in ("default", "farfield"):
            ev.set_variant(_cabi.variant_id(variant))
            ev.set_fit_im(fit_im)
            dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 6, seed=5, minfunc=-1.0, minstep=-1.0)
            host = swarm_support.HostSwarm(lambda X: ev.objective_batch(X, fit_im=fit_im), sp["lower"], sp["upper"], 6, seed=5,
                                 minfunc=-1.0, minstep=-1.0)
            dev.run(2, check_every=2)
            host.init()
            host.apply_global(host.candidate()[None, :])
            for _ in range(2):
                host.step_local()
                host.apply_global(host.candidate()[None, :])
            st = dev.state()
            for k in ("x", "fx", "fp"):
                np.testing.assert_array_equal(st[k], getattr(host, k), err_msg="%s %s" % (variant, k))
            dev.close()


@pytest.mark.parametrize("S,N,P,variant", [(50, 4096, 6, "default"), (204, 4096, 6, "default"), (204, 16384, 12, "farfield"),
                                           (256, 2048, 3, "norec"), (257, 4096, 6, "default"), (1024, 4096, 6, "default"),
                                           (1025, 4096, 6, "default")])
def test_one_launch_generation(S, N, P, variant):
    """Round 4: a single-rank generation of up to 1024 particles (2048 with eight-wave workgroups) is ONE launch where
    a workgroup is a particle -- the objective kernel updates the position, evaluates and updates the personal best,
    and the rest of the generation (argmin over fp, candidate record, pyswarm's acceptance / stopping rule: the body
    of pyswarm.pso's loop, nmrfit/utils.py:176-182) is deferred into the NEXT launch's prologue, every workgroup for
    itself (csrc/pso_update.h, PsoFused::tail).  Bit-identical to the two-launch form and to the numpy mirror, with
    the stopping rule armed (same stop generation, same returned best) and disarmed, polled at odd intervals (every
    poll folds the waiting generation in a launch of its own); larger swarms keep the separate launch."""
    from nmrfit_amd import equations
    sp = synth.make_spectrum(N, P, seed=11)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        ev.set_variant(_cabi.variant_id(variant))
        for kw, gens in ((dict(minfunc=-1.0, minstep=-1.0), 60), ({}, 400)):
            host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=41, **kw)
            xh, fh = pso.run_sharded(host, pso.LocalExchange(), gens)
            res = {}
            for fused in (True, False):
                dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=41, **kw)
                dev.set_fused_tail(fused)
                dev.run(gens, check_every=9)
                assert dev.last_launches() == (1 if (fused and S <= 1024) else 2), (S, fused, dev.last_launches())
                st = dev.state()
                for k in ("x", "v", "p", "fx", "fp"):
                    np.testing.assert_array_equal(st[k], getattr(host, k), err_msg="%s fused=%s" % (k, fused))
                xb, fb = dev.best()
                np.testing.assert_array_equal(xb, xh)
                assert fb == fh
                assert dev.status() == dict(iteration=host.iteration, stop=host.stop, fg=host.fg)
                np.testing.assert_array_equal(dev.candidate(), host.candidate())
                res[fused] = dev.status()
                dev.close()
            if not kw:
                assert res[True]["stop"] in (1, 2)          # pyswarm's defaults do stop these searches


@pytest.mark.parametrize("S,N,P", [(204, 4096, 6), (1024, 4096, 6)])
def test_deferred_fold_is_invisible_from_outside(S, N, P):
    """The deferred fold (test_one_launch_generation) leaves the last generation unfolded on the device until the
    next launch -- or until somebody looks: every entry point that shows or continues the swarm's state folds it
    first (csrc/pso.hip, flush_fold).  One generation at a time through nmrfit_pso_step with a different reader
    after each (status, best, state, candidate, none at all), against the numpy mirror at every generation."""
    from nmrfit_amd import equations
    sp = synth.make_spectrum(N, P, seed=12)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        kw = dict(minfunc=1e-3, minstep=1e-8)                # stops after some tens of generations: the stop is deferred too
        host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=77, **kw)
        dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=77, **kw)
        host.init()
        host.apply_global(host.candidate()[None, :])
        dev.init()
        dev.step()                                           # folds generation 0
        stopped_at = None
        for gen in range(1, 121):
            host.step_local()
            host.apply_global(host.candidate()[None, :])
            dev.step()
            assert dev.last_launches() == 1
            reader = gen % 5
            if reader == 0:
                assert dev.status() == dict(iteration=host.iteration, stop=host.stop, fg=host.fg), gen
            elif reader == 1:
                xb, fb = dev.best()
                np.testing.assert_array_equal(xb, host.best_x)
                assert fb == host.best_f
            elif reader == 2:
                st = dev.state()
                for k in ("x", "v", "p", "fx", "fp"):
                    np.testing.assert_array_equal(st[k], getattr(host, k), err_msg="%s at generation %d" % (k, gen))
            elif reader == 3:
                np.testing.assert_array_equal(dev.candidate(), host.candidate())
            if host.stop and stopped_at is None:
                stopped_at = gen
        assert stopped_at is not None and stopped_at < 110          # ... and ten more no-op generations after it
        assert dev.status() == dict(iteration=host.iteration, stop=host.stop, fg=host.fg)
        st = dev.state()
        for k in ("x", "v", "p", "fx", "fp"):
            np.testing.assert_array_equal(st[k], getattr(host, k), err_msg=k)
        dev.close()


def test_deferred_fold_survives_a_change_of_kernel_between_generations():
    """A generation is waiting to be folded and the next objective launch cannot do it: the caller switched to a
    kernel without the eight-wave workgroup form -- an A/B variant (A/B library) or the imaginary channel (every
    library) -- so that 204 x 4096 x 6 is no longer one workgroup per particle.  launch_objective reports it before
    launching anything, the swarm folds in a launch of its own and goes on in the two-launch form; switching back
    resumes the one-launch form.  Against the numpy mirror."""
    from nmrfit_amd import equations
    S, N, P = 204, 4096, 6
    sp = synth.make_spectrum(N, P, seed=13)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        kw = dict(minfunc=-1.0, minstep=-1.0)
        mode = [False]
        host = swarm_support.HostSwarm(lambda X: ev.objective_batch(X, fit_im=mode[0]), sp["lower"], sp["upper"], S, seed=5, **kw)
        dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=5, **kw)
        host.init()
        host.apply_global(host.candidate()[None, :])
        dev.init()
        dev.step()
        if _cabi.has_ab_variants():
            walk = (("default", False, 1), ("quad", False, 3), ("default", False, 1), ("single", False, 3), ("farfield", False, 1))
        else:   # (3: objective, posts, final reduction -- the default hand-over is TWO_LAUNCH since round 5)
            walk = (("default", False, 1), ("default", True, 3), ("default", False, 1), ("norec", "sum", 3), ("farfield", False, 1))
        for variant, fit_im, launches in walk:
            ev.set_variant(_cabi.variant_id(variant))
            ev.set_fit_im(fit_im)
            mode[0] = fit_im
            for _ in range(7):
                host.step_local()
                host.apply_global(host.candidate()[None, :])
                dev.step()
                assert dev.last_launches() == launches, (variant, dev.last_launches())
        st = dev.state()
        for k in ("x", "v", "p", "fx", "fp"):
            np.testing.assert_array_equal(st[k], getattr(host, k), err_msg=k)
        assert dev.status() == dict(iteration=host.iteration, stop=host.stop, fg=host.fg)
        dev.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_walk_over_the_swarm_interface_matches_the_mirror(seed):
    """Differential fuzz of the C-ABI's swarm entry points around the deferred fold: a random sequence of generations
    (one call, several calls, nmrfit_pso_run, host-staged step_local + apply_global), readers (status, best, state,
    candidate), knob changes (fused tail / fused personal bests / hand-over mode) and kernel-variant changes on ONE
    swarm -- after every operation that reads, the device must equal the numpy mirror bit for bit, and at the end
    everything must.  Whatever the order, no entry point may see, or continue from, an unfolded generation."""
    from nmrfit_amd import equations
    rng = np.random.default_rng(seed)
    S, N, P = (204, 4096, 6) if seed != 3 else (640, 4096, 6)
    sp = synth.make_spectrum(N, P, seed=20 + seed)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        kw = dict(minfunc=-1.0, minstep=-1.0) if seed != 2 else dict(minfunc=3e-4, minstep=1e-8)
        host = swarm_support.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=900 + seed, **kw)
        dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=900 + seed, **kw)
        host.init()
        host.apply_global(host.candidate()[None, :])
        dev.init()
        dev.step()

        def host_generations(n):
            for _ in range(n):
                host.step_local()
                host.apply_global(host.candidate()[None, :])

        def check_all(tag):
            st = dev.state()
            for k in ("x", "v", "p", "fx", "fp"):
                np.testing.assert_array_equal(st[k], getattr(host, k), err_msg="%s after %s" % (k, tag))
            assert dev.status() == dict(iteration=host.iteration, stop=host.stop, fg=host.fg), tag
            xb, fb = dev.best()
            np.testing.assert_array_equal(xb, host.best_x, err_msg=tag)
            assert fb == host.best_f, tag

        log = []
        for op_i in range(160):
            op = int(rng.integers(0, 14))
            log.append(op)
            tag = "op %d (#%d of %s)" % (op, op_i, log[-12:])
            if op <= 3:                       # one generation, one call
                dev.step()
                host_generations(1)
            elif op == 4:                     # a few generations through the run loop (polls every 2 or 5)
                n = int(rng.integers(1, 9))
                if not host.stop:             # (nmrfit_pso_run counts its own iterations: after a stop it returns at the first poll)
                    dev.run(n, check_every=int(rng.choice([2, 5])))
                    host_generations(n)
            elif op == 5:                     # the host-staged form of a generation
                dev.step_local()
                dev.apply_global(dev.candidate()[None, :])
                host_generations(1)
            elif op == 6:
                assert dev.status() == dict(iteration=host.iteration, stop=host.stop, fg=host.fg), tag
            elif op == 7:
                xb, fb = dev.best()
                np.testing.assert_array_equal(xb, host.best_x, err_msg=tag)
                assert fb == host.best_f, tag
            elif op == 8:
                np.testing.assert_array_equal(dev.candidate(), host.candidate(), err_msg=tag)
            elif op == 9:
                check_all(tag)
            elif op == 10:
                dev.set_fused_tail(bool(rng.integers(0, 2)))
            elif op == 11:
                dev.set_fused_pbest(bool(rng.integers(0, 2)))
            elif op == 12:
                dev.set_handover(str(rng.choice(["fast", "fenced", "two_launch"])))
            elif op == 13:                    # another kernel for the generations that follow (the mirror evaluates through the same context)
                ev.set_variant(_cabi.variant_id(str(rng.choice(["default", "farfield", "norec", "quad", "staged"] if _cabi.has_ab_variants()
                                                                  else ["default", "farfield", "norec", "norec", "default"]))))
        check_all("the end (%s)" % log[-12:])
        if seed == 2:
            assert host.stop in (1, 2)        # this one stops on the way: the no-op generations after it are part of the walk
        dev.close()


def test_fit_many_matches_the_plain_loop():
    """nmrfit_amd.fit_many: several spectra fitted at once on host threads (one context and HIP stream per fit, the
    GIL released inside the library's calls) -- the same results, in order, as looping over nmrfit_amd.fit like the
    reference's users loop over nmrfit.fit (nmrfit/core.py:64)."""
    import nmrfit_amd
    jobs = []
    for k in range(5):
        sp = synth.make_spectrum(2048 + 512 * k, 3 + k, seed=300 + k)
        jobs.append((synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), list(sp["lower"]), list(sp["upper"])))
    opts = {"swarmsize": 120, "maxiter": 150, "seed": 11}
    plain = [nmrfit_amd.fit(d, lo, hi, summary=False, options=opts) for d, lo, hi in jobs]
    many = nmrfit_amd.fit_many(jobs, threads=3, options=opts)
    as_dicts = nmrfit_amd.fit_many([dict(data=d, lower=lo, upper=hi) for d, lo, hi in jobs], threads=5, options=opts)
    assert len(many) == len(as_dicts) == len(jobs)
    for a, b, c in zip(plain, many, as_dicts):
        np.testing.assert_array_equal(a.params, b.params)
        np.testing.assert_array_equal(a.params, c.params)
        assert a.error == b.error == c.error
        np.testing.assert_array_equal(a.weights, b.weights)


@pytest.mark.parametrize("fit_im", [True, "sum"])
def test_one_launch_generation_with_the_imaginary_channel(fit_im):
    """The deferred fold rides in every objective instantiation whose workgroup is the particle -- also the ones with
    the imaginary channel (four-wave workgroups only): 1024 x 4096 x 4 with fit_im=True / "sum" against the numpy mirror
    evaluating through the same context."""
    from nmrfit_amd import equations
    S, N, P = 1024, 4096, 4
    sp = synth.make_spectrum(N, P, seed=31)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        ev.set_fit_im(fit_im)
        kw = dict(minfunc=-1.0, minstep=-1.0)
        host = swarm_support.HostSwarm(lambda X: ev.objective_batch(X, fit_im=fit_im), sp["lower"], sp["upper"], S, seed=17, **kw)
        xh, fh = pso.run_sharded(host, pso.LocalExchange(), 40)
        dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=17, **kw)
        dev.run(40, check_every=7)
        assert dev.last_launches() == 1
        st = dev.state()
        for k in ("x", "v", "p", "fx", "fp"):
            np.testing.assert_array_equal(st[k], getattr(host, k), err_msg=k)
        xb, fb = dev.best()
        np.testing.assert_array_equal(xb, xh)
        assert fb == fh
        dev.close()

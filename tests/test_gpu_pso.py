"""
GPU tier (iii/iv): the device-resident swarm (csrc/pso.hip through the C-ABI) against its
numpy mirror, the sharded form against the single swarm, and nmrfit_amd.fit() end to end.
"""
import numpy as np
import pytest

from nmrfit_amd import _cabi, pso, synth

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
    host = pso.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
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
    host = pso.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], 64, seed=3)
    xh, fh = pso.run_sharded(host, pso.LocalExchange(), 400)
    assert host.stop == res[0][0] and host.iteration == res[0][1]
    np.testing.assert_array_equal(xh, res[0][2][0])
    # the same for a swarm large enough to take the unfused loop
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 600, seed=4)
    sw.run(300, check_every=5)
    st = sw.status()
    host = pso.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], 600, seed=4)
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


_RCCL_SCRIPT = r"""
import socket, sys
import torch                      # before any nmrfit_amd GPU call: one shared HIP runtime
import torch.distributed as dist
import numpy as np
sys.path.insert(0, %r)
import nmrfit_amd
from nmrfit_amd import pso, synth
from nmrfit_amd.equations import Evaluator
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% port, rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
sp = synth.make_spectrum(2048, 3, seed=5)
ev = Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"])
ex = pso.TorchExchange()
assert ex.backend == "nccl" and ex.world == 1
a = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 300, seed=9)
xa, fa = pso.run_sharded(a, ex, 120, check_every=7)
st = a.status(); a.close()
b = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 300, seed=9)
b.run(120, check_every=7)
xb, fb = b.best()
assert b.status() == st, (b.status(), st)
b.close()
assert (xa == xb).all() and fa == fb
f = ev.objective_batch(xa)        # the context is back on its own stream and still usable
assert abs(f[0] - fa) <= 1e-12 * fa
data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
r1 = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False,
                    options={"swarmsize": 100, "maxiter": 40, "seed": 5, "exchange": pso.TorchExchange()})
r2 = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False,
                    options={"swarmsize": 100, "maxiter": 40, "seed": 5})
assert (r1.params == r2.params).all() and r1.error == r2.error
ev.close()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_run_sharded_over_rccl_single_rank():
    """pso.run_sharded with a torch "nccl" group (one rank): the candidate all-gather runs on the
    GPU on the same stream as the swarm kernels (RcclGeneration) and the result equals the plain
    device loop; nmrfit_amd.fit(options={"exchange": ...}) takes the same route.  Runs in its own
    process because torch must be imported before libnmrfit_amd is loaded (shared HIP runtime)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT % root], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stderr[-3000:]


def test_rccl_needs_torch_first(problem):
    """In THIS process libnmrfit_amd was loaded first: the RCCL generation must refuse loudly
    instead of letting torch fail to find the GPU."""
    sp, ev = problem
    if not _cabi.loaded_before_torch():
        pytest.skip("torch was imported before the library in this process")
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 16, seed=1)
    with pytest.raises(RuntimeError, match="import torch before"):
        pso.RcclGeneration(sw, pso.LocalExchange())
    sw.close()


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


@pytest.mark.parametrize("S,P", [(12, 100), (12, 300), (300, 120), (3, 1000), (1, 2), (2, 0)])
def test_device_swarm_wide_parameter_vectors(S, P):
    """Many-peak models (D = 4 + 3P up to 3004) and degenerate swarms, on both sides of the
    fused-tail threshold: still bit-identical to the numpy mirror."""
    from nmrfit_amd import equations
    sp = synth.make_spectrum(1024, P, seed=9)
    with equations.Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        dev = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=31, minfunc=-1.0, minstep=-1.0)
        host = pso.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=31, minfunc=-1.0, minstep=-1.0)
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
    host = pso.HostSwarm(ev.objective_batch, sp["lower"], sp["upper"], S, seed=seed, minfunc=-1.0, minstep=-1.0)
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

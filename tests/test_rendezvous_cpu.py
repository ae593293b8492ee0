"""
CPU tier (iv) without torch: the standard-library rendezvous (nmrfit_amd/rendezvous.py) that
hands RCCL's unique id to the ranks, and the sharded generation loop over it
(swarm_support.SocketExchange) with real processes -- world sizes 2, 3 and 8 (the C4 rank count) on
127.0.0.1.  Every rank must end with the single-rank answer bit for bit.  Also: the
self-launching `bench.py --gpus N` must fail fast and loudly when its ranks fail (here: no GPU),
never hang; its --launch-timeout ends ranks that never arrive and names them; every rank's own
watchdog does the same when the launcher is someone else's (torch.distributed.run).
"""
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(rank, world, port, token, extra):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NMRFIT_RDZV_TOKEN=token, NMRFIT_RDZV_TIMEOUT="60")
    os.environ.pop("NMRFIT_RDZV_PORT", None)
    os.environ.update(extra)


def _channel_worker(rank, world, port, token, extra, q):
    sys.path.insert(0, ROOT)
    _env(rank, world, port, token, extra)
    from nmrfit_amd import rendezvous
    with rendezvous.Channel() as ch:
        uid = ch.broadcast(bytes(range(128)) if rank == 0 else b"")
        parts = ch.all_gather(b"r%d" % rank * (rank + 1))
        ch.barrier()
        q.put((rank, uid, parts))


@pytest.mark.parametrize("world,extra", [(2, {}), (5, {}), (3, {"NMRFIT_RDZV_PORT": "auto"})])
def test_channel_broadcast_and_all_gather(world, extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    if extra.get("NMRFIT_RDZV_PORT") == "auto":      # explicit TCP port instead of the published file
        extra = {"NMRFIT_RDZV_PORT": str(_free_port())}
    token = "t%d_%d" % (os.getpid(), time.time_ns())
    ps = [ctx.Process(target=_channel_worker, args=(r, world, port, token, extra, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    expect = [b"r%d" % r * (r + 1) for r in range(world)]
    for rank, uid, parts in got:
        assert uid == bytes(range(128))
        assert parts == expect


def _late_worker(rank, world, port, token, delay, q):
    sys.path.insert(0, ROOT)
    time.sleep(delay)
    _env(rank, world, port, token, {})
    from nmrfit_amd import rendezvous
    with rendezvous.Channel() as ch:
        q.put((rank, ch.broadcast(b"id-from-rank-0" if rank == 0 else b"")))


def test_channel_rejects_strangers_and_tolerates_late_ranks():
    """A connection that does not know the launch token never becomes a rank, and a rank that
    arrives seconds late (a slow `import`, a cold GPU) still joins."""
    import hashlib
    import tempfile
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    token = "x%d_%d" % (os.getpid(), time.time_ns())
    ps = [ctx.Process(target=_late_worker, args=(r, 2, port, token, 0.0 if r == 0 else 3.0, q)) for r in range(2)]
    for p in ps:
        p.start()
    # while rank 0 waits for rank 1: find its published port and knock with a wrong token
    key = "127.0.0.1|%d|%s|%d" % (port, token, os.getuid())
    path = os.path.join(tempfile.gettempdir(), "nmrfit_rdzv_%s" % hashlib.sha256(key.encode()).hexdigest()[:24])
    deadline = time.time() + 30
    while not os.path.exists(path) and time.time() < deadline:
        time.sleep(0.05)
    host, prt = open(path).read().strip().rsplit(":", 1)
    s = socket.create_connection((host, int(prt)), timeout=5)
    s.sendall(b"\x10\x00\x00\x00\x00\x00\x00\x00" + b"GET / HTTP/1.1\r\n\r\n")
    s.close()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert got == {0: b"id-from-rank-0", 1: b"id-from-rank-0"}
    assert not os.path.exists(path)          # rank 0 removed the file once everybody had joined


def test_single_rank_channel_needs_no_network():
    sys.path.insert(0, ROOT)
    from nmrfit_amd import rendezvous
    ch = rendezvous.Channel(rank=0, world=1)
    assert ch.all_gather(b"x") == [b"x"] and ch.broadcast(b"y") == b"y"
    ch.barrier()
    ch.close()


def _swarm_worker(rank, world, port, token, S, maxiter, seed, out_dir):
    sys.path.insert(0, ROOT)
    _env(rank, world, port, token, {})
    from nmrfit_amd import pso, synth
    from tests import swarm_support
    from oracle import c_oracle
    sp = synth.make_spectrum(512, 2, seed=5)

    def evaluate(X):
        return c_oracle.objective_batch(X, sp["w"], sp["u"], sp["v"], sp["weights"], threads=1)
    ex = swarm_support.SocketExchange()
    assert (ex.rank, ex.world) == (rank, world)
    seed = ex.broadcast_seed(seed if rank == 0 else 999)        # rank 0's seed wins
    assert float(ex.all_reduce([float(rank)], "max")[0]) == world - 1
    off, n = pso.shard(S, ex.rank, ex.world)
    sw = swarm_support.HostSwarm(evaluate, sp["lower"], sp["upper"], swarmsize=S, offset=off, S_local=n, seed=seed,
                       minfunc=-1.0, minstep=-1.0)
    x, f = pso.run_sharded(sw, ex, maxiter=maxiter)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, f=f, g=sw.g, fg=sw.fg, it=sw.iteration, xs=sw.x)
    ex.barrier()
    ex.close()


@pytest.mark.parametrize("world,S", [(2, 20), (3, 17), (8, 40)])
def test_sharded_swarm_over_sockets_equals_single_rank(tmp_path, world, S):
    sys.path.insert(0, ROOT)
    from nmrfit_amd import pso, synth
    from tests import swarm_support
    from oracle import c_oracle
    maxiter, seed = 12, 4242
    ctx = mp.get_context("spawn")
    port = _free_port()
    token = "s%d_%d" % (os.getpid(), time.time_ns())
    ps = [ctx.Process(target=_swarm_worker, args=(r, world, port, token, S, maxiter, seed, str(tmp_path)))
          for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(180)
        assert p.exitcode == 0
    sp = synth.make_spectrum(512, 2, seed=5)
    sw1 = swarm_support.HostSwarm(lambda X: c_oracle.objective_batch(X, sp["w"], sp["u"], sp["v"], sp["weights"], threads=1),
                        sp["lower"], sp["upper"], swarmsize=S, seed=seed, minfunc=-1.0, minstep=-1.0)
    x1, f1 = pso.run_sharded(sw1, pso.LocalExchange(), maxiter=maxiter)
    xs = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        np.testing.assert_array_equal(d["x"], x1)
        assert float(d["f"]) == f1 and int(d["it"]) == maxiter
        np.testing.assert_array_equal(d["g"], sw1.g)
        xs.append(d["xs"])
    np.testing.assert_array_equal(np.concatenate(xs), sw1.x)


def test_bench_self_launch_fails_fast_when_ranks_fail():
    """`python bench.py --gpus 2` with no launcher environment starts its own two ranks.  Without a
    GPU they fail at device selection; the launcher must notice, end the other rank, and exit
    non-zero within seconds -- never sit in a rendezvous waiting for a rank that is gone."""
    from nmrfit_amd import _cabi
    if _cabi.device_count() > 0:
        pytest.skip("needs a box without a GPU")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--cpu-seconds", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=120)
    assert out.returncode != 0
    assert time.time() - t0 < 60
    assert "rank" in out.stderr and ("no HIP device is visible" in out.stderr or "NO_DEVICE" in out.stderr
                                     or "error -2" in out.stderr or "hipGetDeviceCount" in out.stderr)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def _stalled_bench(stall, extra_env=None, launch_timeout="3"):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["NMRFIT_BENCH_TEST_STALL"] = stall
    env.update(extra_env or {})
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--cpu-seconds", "0", "--launch-timeout", launch_timeout], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120)
    return out, time.time() - t0


def test_bench_launch_timeout_ends_ranks_that_never_arrive():
    """Ranks that hang before they could notice anything themselves (here: a test stall ahead of the
    rank's own watchdog; on hardware: anything) are ended by the launcher's deadline: non-zero exit,
    the ranks still alive named on stderr, no JSON line, and no process left behind."""
    out, dt = _stalled_bench("pre:all:90")      # (the launcher's deadline falls 15 s after the ranks' own: 3 + 15 s)
    assert out.returncode == 124, (out.returncode, out.stderr[-2000:])
    assert 15 < dt < 60
    assert "launch timeout" in out.stderr and "[0, 1]" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_rank_watchdog_names_the_phase():
    """The driver launches the ranks under torch.distributed.run, not under bench.py's launcher: a
    rank stuck on the way to the timed region (rendezvous, ncclCommInitRank, first collective) ends
    ITSELF after --launch-timeout, saying on stderr which rank was stuck where; the launcher then
    sees a failed rank.  Here rank 1 stalls inside its watchdog and rank 0 is run as a torchrun-style
    rank by hand."""
    from nmrfit_amd import _cabi
    if _cabi.device_count() > 0:
        pytest.skip("needs a box without a GPU (rank 0 must fail at context creation, not run)")
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), NMRFIT_BENCH_TEST_STALL="in:1:60", NMRFIT_RDZV_TOKEN="wd%d" % os.getpid())
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--cpu-seconds", "0", "--launch-timeout", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 124, (out.returncode, out.stderr[-2000:])
    assert time.time() - t0 < 30
    assert "nmrfit watchdog: rank 1 still in `test stall` after 2 s" in out.stderr
    assert "HIP device" in out.stderr and "world 2" in out.stderr


def test_watchdog_is_silent_when_the_step_finishes():
    sys.path.insert(0, ROOT)
    from nmrfit_amd import rendezvous
    with rendezvous.Watchdog(5.0, "quick step", rank=0) as dog:
        dog.phase = "still quick"
    with rendezvous.Watchdog(0, "disabled", rank=0):
        time.sleep(0.01)


def _portmode_rank(rank, world, master_port, rdzv_port, out_path):
    """One rank in NMRFIT_RDZV_PORT mode WITHOUT a token, started through an intermediate process
    of its own (so the ranks' parent pids differ, as they do across nodes)."""
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from nmrfit_amd import rendezvous\n"
            "with rendezvous.Channel() as ch:\n"
            "    parts = ch.all_gather(b'p%%d' %% ch.rank)\n"
            "open(%r, 'w').write(repr(parts))\n" % (ROOT, out_path))
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(master_port), NMRFIT_RDZV_PORT=str(rdzv_port), NMRFIT_RDZV_TIMEOUT="60",
               TORCHELASTIC_RUN_ID="job42")
    env.pop("NMRFIT_RDZV_TOKEN", None)
    inner = [sys.executable, "-c", code]
    # the intermediate parent: a different process for every rank
    return subprocess.Popen([sys.executable, "-c", "import subprocess, sys; sys.exit(subprocess.call(%r))" % (inner,)],
                            env=env)


def test_port_mode_without_token_and_different_parents(tmp_path):
    """with NMRFIT_RDZV_PORT (the multi-node mode) the default token must not be the
    parent pid -- ranks on different nodes have different parents.  It is derived from the
    launch-wide values instead (MASTER_ADDR, MASTER_PORT, WORLD_SIZE, TORCHELASTIC_RUN_ID), and rank
    0 listens on MASTER_ADDR only."""
    world, mport, rport = 3, _free_port(), _free_port()
    outs = [str(tmp_path / ("r%d.txt" % r)) for r in range(world)]
    ps = [_portmode_rank(r, world, mport, rport, outs[r]) for r in range(world)]
    for p in ps:
        assert p.wait(timeout=120) == 0
    for o in outs:
        assert open(o).read() == repr([b"p0", b"p1", b"p2"])
    # a different run id is a different launch: its token does not match
    sys.path.insert(0, ROOT)
    from nmrfit_amd import rendezvous
    old = dict(os.environ)
    try:
        os.environ.update(NMRFIT_RDZV_PORT=str(rport), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(mport), WORLD_SIZE="3")
        os.environ.pop("NMRFIT_RDZV_TOKEN", None)
        os.environ["TORCHELASTIC_RUN_ID"] = "job42"
        t1 = rendezvous._token()
        os.environ["TORCHELASTIC_RUN_ID"] = "job43"
        assert rendezvous._token() != t1 and t1.startswith("launch")
        os.environ.pop("NMRFIT_RDZV_PORT")
        assert rendezvous._token() == "ppid%d" % os.getppid()
    finally:
        os.environ.clear()
        os.environ.update(old)


def _precheck_worker(rank, world, port, token, q):
    sys.path.insert(0, ROOT)
    _env(rank, world, port, token, {})
    if rank == 1:
        os.environ["NMRFIT_RCCL_LIB"] = "/nonexistent/librccl.so"     # RCCL "missing" on this rank only
    from nmrfit_amd import _cabi, pso

    class NoGpuEvaluator:          # never reached: the pre-check fails before anything touches a context
        handle, device = None, 0
        _children = set()
    t0 = time.time()
    try:
        pso.RcclExchange(NoGpuEvaluator())
        q.put((rank, "no error", 0.0))
    except _cabi.NmrfitError as e:
        q.put((rank, "%d|%s" % (e.code, e), time.time() - t0))


def test_rccl_missing_on_one_rank_is_an_error_on_every_rank():
    """no rank may enter the collective ncclCommInitRank unless EVERY rank can load RCCL.
    The ranks compare notes over the socket star first (nmrfit_comm_available: dlopen + symbols, no
    GPU work), so RCCL missing on rank 1 raises NMRFIT_E_UNSUPPORTED on ranks 0 AND 1 within seconds,
    naming the rank -- nobody is left waiting inside RCCL."""
    from nmrfit_amd import _cabi
    if _cabi.lib().nmrfit_comm_available() != _cabi.OK:
        pytest.skip("librccl cannot be loaded on this box at all")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, token = _free_port(), "pc%d_%d" % (os.getpid(), time.time_ns())
    ps = [ctx.Process(target=_precheck_worker, args=(r, 2, port, token, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict((r, (msg, dt)) for r, msg, dt in (q.get(timeout=120) for _ in range(2)))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        msg, dt = got[r]
        assert msg.startswith("%d|" % _cabi.E_UNSUPPORTED), msg
        assert "RCCL is not available on rank(s) [1]" in msg and "/nonexistent/librccl.so" in msg
        assert dt < 30


def test_pick_device_over_launcher_policies():
    """VERDICT r3 item 1 (the reference's parallel mode is a process pool, nmrfit/utils.py:182 --
    here one process per GPU): the rank -> GPU mapping over {every device visible, one device per rank by
    HIP_/ROCR_/CUDA_VISIBLE_DEVICES, isolation without a variable, a partial list} x world 8."""
    from nmrfit_amd import rendezvous as rz
    # (1) torch.distributed.run / bench.py's launcher: all 8 devices visible to every rank, nothing set
    for lr in range(8):
        env = {"LOCAL_RANK": str(lr), "RANK": str(lr), "WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "8"}
        assert rz.pick_device(8, env=env) == (lr, "")
    # (1b) the full list spelled out is the same thing
    env = {"LOCAL_RANK": "5", "WORLD_SIZE": "8", "HIP_VISIBLE_DEVICES": "0,1,2,3,4,5,6,7"}
    assert rz.pick_device(8, env=env) == (5, "")
    # (2) per-rank isolation: one visible device, any of the three variables
    for var in rz.VISIBILITY_VARS:
        for lr in range(8):
            env = {"LOCAL_RANK": str(lr), "WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "8", var: str(lr)}
            dev, note = rz.pick_device(1, env=env)
            assert dev == 0
            assert (note == "") == (lr == 0)
            if lr:
                assert "%s=%d" % (var, lr) in note and "isolated" in note
    # (3) one visible device, no variable (a device cgroup): device 0, and the note says what was assumed
    dev, note = rz.pick_device(1, env={"LOCAL_RANK": "6", "WORLD_SIZE": "8"})
    assert dev == 0 and "none of HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES is set" in note
    # (4) a partial list that does not cover the local rank: an error naming the variables, not a guess
    with pytest.raises(RuntimeError, match="LOCAL_RANK=5 of 8 local rank.*4 HIP devices are visible.*HIP_VISIBLE_DEVICES=0,1,2,3"):
        rz.pick_device(4, env={"LOCAL_RANK": "5", "WORLD_SIZE": "8", "HIP_VISIBLE_DEVICES": "0,1,2,3"})
    with pytest.raises(RuntimeError, match="no visibility variable set.*check HIP_VISIBLE_DEVICES"):
        rz.pick_device(2, env={"LOCAL_RANK": "2", "WORLD_SIZE": "4"})
    with pytest.raises(RuntimeError, match="no HIP device is visible"):
        rz.pick_device(0, env={"LOCAL_RANK": "0", "HIP_VISIBLE_DEVICES": ""})
    # LOCAL_RANK absent: RANK stands in (single-node launchers that set only RANK)
    assert rz.pick_device(8, env={"RANK": "3", "WORLD_SIZE": "8"}) == (3, "")
    assert rz.visible_devices_env({"ROCR_VISIBLE_DEVICES": "2", "CUDA_VISIBLE_DEVICES": "0"}) == \
        "ROCR_VISIBLE_DEVICES=2 CUDA_VISIBLE_DEVICES=0"


def test_bench_rank_with_an_impossible_device_fails_before_the_rendezvous():
    """bench.py: LOCAL_RANK beyond the visible devices (and more than one visible) ends the rank with exit 6
    and a message naming the variables -- before any socket is opened.  (No GPU here: device_count() is 0,
    which is the same code path with "no HIP device is visible".)"""
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1",
               HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-seconds", "0"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 6, (out.returncode, out.stderr[-1500:])
    assert "bench.py rank 1/2: no HIP device is visible to local rank 1 (HIP_VISIBLE_DEVICES=)" in out.stderr
    assert not out.stdout.strip()

"""
CPU tier (iv) without torch: the standard-library rendezvous (nmrfit_amd/rendezvous.py) that
hands RCCL's unique id to the ranks, and the sharded generation loop over it
(pso.SocketExchange) with real processes -- world sizes 2, 3 and 8 (the C4 rank count) on
127.0.0.1.  Every rank must end with the single-rank answer bit for bit.  Also: the
self-launching `bench.py --gpus N` must fail fast and loudly when its ranks fail (here: no GPU),
never hang.
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
    from oracle import c_oracle
    sp = synth.make_spectrum(512, 2, seed=5)

    def evaluate(X):
        return c_oracle.objective_batch(X, sp["w"], sp["u"], sp["v"], sp["weights"], threads=1)
    ex = pso.SocketExchange()
    assert (ex.rank, ex.world) == (rank, world)
    seed = ex.broadcast_seed(seed if rank == 0 else 999)        # rank 0's seed wins
    assert float(ex.all_reduce([float(rank)], "max")[0]) == world - 1
    off, n = pso.shard(S, ex.rank, ex.world)
    sw = pso.HostSwarm(evaluate, sp["lower"], sp["upper"], swarmsize=S, offset=off, S_local=n, seed=seed,
                       minfunc=-1.0, minstep=-1.0)
    x, f = pso.run_sharded(sw, ex, maxiter=maxiter)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, f=f, g=sw.g, fg=sw.fg, it=sw.iteration, xs=sw.x)
    ex.barrier()
    ex.close()


@pytest.mark.parametrize("world,S", [(2, 20), (3, 17), (8, 40)])
def test_sharded_swarm_over_sockets_equals_single_rank(tmp_path, world, S):
    sys.path.insert(0, ROOT)
    from nmrfit_amd import pso, synth
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
    sw1 = pso.HostSwarm(lambda X: c_oracle.objective_batch(X, sp["w"], sp["u"], sp["v"], sp["weights"], threads=1),
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
    GPU they fail at context creation; the launcher must notice, end the other rank, and exit
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
    assert "rank" in out.stderr and ("NO_DEVICE" in out.stderr or "error -2" in out.stderr or "hipGetDeviceCount" in out.stderr)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]

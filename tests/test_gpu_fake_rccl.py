"""
GPU tier: the product's own multi-rank RCCL branch with MORE THAN ONE RANK on hardware, as far as a
one-GPU box allows.  RCCL refuses two ranks on one device, so csrc/comm.hip with nranks > 1,
nmrfit_pso_step with a communicator attached on several ranks, bench.py's N > 1 RCCL branch
(`rccl` object, rank count by all-reduce, timing reductions over the communicator) and
fit(options={"exchange": "rccl"}) with world > 1 had never run.  tests/fake_rccl/fake_rccl.cpp
implements the nine RCCL entry points the library dlopens with the same signatures and stream
semantics, moving the data through shared host memory, and is selected through NMRFIT_RCCL_LIB --
by these tests only.  What this covers is OUR code on that path (it says nothing about RCCL itself;
the line it produces says version 99999 and names the library override).
"""
import json
import os
import shutil
import socket
import subprocess
import sys

import numpy as np
import pytest

from nmrfit_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def fake_lib(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.run([HIPCC, "-shared", "-fPIC", "-O2", os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), "-o", out],
                   check=True, capture_output=True, text=True)
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(cmd, env_extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), out.stderr


def test_bench_rccl_branch_with_two_and_four_ranks(fake_lib):
    common = ["--steps", "5", "--warmup", "2", "--cpu-seconds", "0", "--workload", "C2", "--preheat-seconds", "0.1",
              "--no-extras"]
    plain, _ = _bench([sys.executable, "bench.py", "--swarm-per-gpu", "512"] + common, {})
    env = {"NMRFIT_RCCL_LIB": fake_lib, "NMRFIT_BENCH_SHARE_GPU": "1"}
    for n in (2, 4):
        d, err = _bench([sys.executable, "bench.py", "--gpus", str(n), "--swarm-per-gpu", str(512 // n)] + common, env)
        assert d["n_gpus"] == n and d["config"]["swarm_total"] == 512 and "error" not in d
        assert d["rccl"]["nranks"] == n and d["rccl"]["ranks_counted_by_all_reduce"] == n
        assert d["rccl"]["version"] == 99999 and d["rccl"]["library"] == fake_lib      # labelled: not RCCL
        assert "ncclAllGather" in d["config"]["exchange"]
        # sharding and the exchange do not change the trajectory: same swarm best as one rank, bit for bit
        assert d["config"]["swarm_best_f"] == plain["config"]["swarm_best_f"]
        assert d["config"]["generations_done"] == plain["config"]["generations_done"] == 7
        assert d["ranks"]["kernel_ms_mean"]["min"] > 0 and d["ranks"]["timed_region_s"]["max"] >= d["ranks"]["timed_region_s"]["min"]
        for r in range(n):      # every rank announced itself and its communicator
            assert "bench.py rank %d/%d: creating the RCCL communicator" % (r, n) in err
        assert err.count("RCCL communicator ready: rank") == n
    # the driver's form: torch.distributed.run as the launcher
    t, _ = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                   "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--swarm-per-gpu", "256"]
                  + common, env)
    assert t["rccl"]["nranks"] == 2 and t["config"]["swarm_best_f"] == plain["config"]["swarm_best_f"]


_FIT_RANK = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import nmrfit_amd
from nmrfit_amd import synth
sp = synth.make_spectrum(4096, 6, seed=21)
data = synth.SynthData(sp['w'], sp['u'], sp['v'], sp['peaks'])
res = nmrfit_amd.fit(data, list(sp['lower']), list(sp['upper']), summary=False,
                     options={"swarmsize": 301, "maxiter": 60, "exchange": "rccl", "device": 0})    # unseeded
json.dump({"params": [float(v).hex() for v in res.params], "error": float(res.error).hex(), "seed": int(res.seed)},
          open(os.path.join(%(out)r, "fit_rank%%d.json" %% int(os.environ["RANK"])), "w"))
"""


def test_fit_with_the_rccl_exchange_and_three_ranks(fake_lib, tmp_path):
    """nmrfit_amd.fit(options={"exchange": "rccl"}) with world = 3: communicator built from the launcher's
    environment, seed broadcast OVER THE COMMUNICATOR (nmrfit_comm_broadcast_host), sharded swarm run inside
    the library (nmrfit_pso_run with the communicator attached on every rank)."""
    import nmrfit_amd
    port = _free_port()
    code = _FIT_RANK % dict(root=ROOT, out=str(tmp_path))
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NMRFIT_RDZV_TOKEN="ffit%d_%d" % (os.getpid(), port), NMRFIT_RCCL_LIB=fake_lib)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    got = [json.load(open(os.path.join(str(tmp_path), "fit_rank%d.json" % r))) for r in range(3)]
    assert got[1] == got[0] and got[2] == got[0]
    sp = synth.make_spectrum(4096, 6, seed=21)
    data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
    one = nmrfit_amd.fit(data, list(sp["lower"]), list(sp["upper"]), summary=False,
                         options={"swarmsize": 301, "maxiter": 60, "seed": got[0]["seed"]})
    assert [float(v).hex() for v in one.params] == got[0]["params"] and float(one.error).hex() == got[0]["error"]


def test_c4_per_rank_shape_over_the_rccl_branch(fake_lib):
    """C4's per-GPU shape (4096 particles x 65536 points x 24 peaks per rank) with four ranks through the RCCL
    branch (stand-in library, one GPU): same swarm best as the one-rank 16384-particle run, bit for bit."""
    common = ["--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-extras", "--no-other-configs", "--no-pmc",
              "--preheat-seconds", "0.1"]
    four, _ = _bench([sys.executable, "bench.py", "--gpus", "4", "--swarm-per-gpu", "4096"] + common,
                     {"NMRFIT_RCCL_LIB": fake_lib, "NMRFIT_BENCH_SHARE_GPU": "1"})
    one, _ = _bench([sys.executable, "bench.py", "--swarm-per-gpu", "16384"] + common, {})
    assert four["rccl"]["nranks"] == 4 and four["config"]["swarm_total"] == 16384 == one["config"]["swarm_total"]
    assert four["config"]["swarm_best_f"] == one["config"]["swarm_best_f"]
    assert four["config"]["generations_done"] == one["config"]["generations_done"] == 4


def test_watchdog_ends_ranks_stuck_inside_communicator_creation(fake_lib):
    """The failure the first multi-GPU contact is most likely to show: a rank that never comes back from
    ncclCommInitRank.  Here rank 1 of 2 hangs INSIDE the (stand-in) collective creation, so rank 0 waits in
    it too -- both main threads are stuck in a C call.  Every rank's own watchdog thread (the ctypes call
    releases the GIL) says on stderr which rank is stuck where, on which device, and ends the process; the
    launcher then reports non-zero.  Well inside a minute, never a hang."""
    import time
    env = dict(os.environ, NMRFIT_RCCL_LIB=fake_lib, NMRFIT_BENCH_SHARE_GPU="1", FAKE_RCCL_HANG_IN_INIT="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--swarm-per-gpu", "128", "--workload", "C2",
                          "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extras", "--launch-timeout", "10"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode != 0
    assert time.time() - t0 < 90
    assert "nmrfit watchdog: rank" in out.stderr and "RCCL communicator creation" in out.stderr, out.stderr[-3000:]
    assert "HIP device 0 of " in out.stderr and ", PCI " in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]

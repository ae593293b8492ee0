"""
GPU tier (iv), end to end through bench.py: the one-rank path, the RCCL path (torch
"nccl" backend, candidate all-gather and swarm kernels ordered on one HIP stream) forced with a
single rank, and two ranks sharing the one GPU of the test box over gloo.  All three must report
the same swarm best bit for bit: sharding and the exchange path do not change the trajectory.
The 8-GPU RCCL run itself belongs to the driver; this covers its code path as far as one GPU can.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_paths_agree():
    common = ["--steps", "5", "--warmup", "2", "--cpu-seconds", "0", "--workload", "C2"]
    plain = _run([sys.executable, "bench.py", "--swarm-per-gpu", "512"] + common, {})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in plain, key
    assert plain["dtype"] == "f64" and plain["vs_baseline"] is None and plain["n_gpus"] == 1
    r = plain["roofline"]
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    rccl = _run([sys.executable, "bench.py", "--swarm-per-gpu", "512"] + common, {"NMRFIT_BENCH_FORCE_DIST": "1"})
    assert rccl["config"]["swarm_best_f"] == plain["config"]["swarm_best_f"]
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2",
                "--swarm-per-gpu", "256"] + common, {"NMRFIT_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["swarm_total"] == 512
    assert two["config"]["swarm_best_f"] == plain["config"]["swarm_best_f"]
    assert two["config"]["generations_done"] == plain["config"]["generations_done"] == 7

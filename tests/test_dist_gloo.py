"""
CPU tier (iv): the N>1 path with real processes -- world_size 2 (and 3) over gloo on
127.0.0.1.  Each rank runs the sharded generation loop (nmrfit_amd.pso.run_sharded with
TorchExchange) on the numpy mirror with the oracle injected as evaluator; the result must be
bit-identical on every rank and identical to the single-rank run (SURVEY 8(e) determinism).
On the GPU the same loop drives DeviceSwarm and the exchange is RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, S, maxiter, seed, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nmrfit_amd import pso, synth
    from tests import swarm_support
    from oracle import c_oracle
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        sp = synth.make_spectrum(512, 2, seed=5)

        def evaluate(X):
            return c_oracle.objective_batch(X, sp["w"], sp["u"], sp["v"], sp["weights"], threads=1)
        ex = swarm_support.TorchExchange()
        off, n = pso.shard(S, ex.rank, ex.world)
        sw = swarm_support.HostSwarm(evaluate, sp["lower"], sp["upper"], swarmsize=S, offset=off, S_local=n, seed=seed,
                           minfunc=-1.0, minstep=-1.0)
        x, f = pso.run_sharded(sw, ex, maxiter=maxiter)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, f=f, g=sw.g, fg=sw.fg, it=sw.iteration,
                 xs=sw.x, off=off)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _single(S, maxiter, seed):
    sys.path.insert(0, ROOT)
    from nmrfit_amd import pso, synth
    from tests import swarm_support
    from oracle import c_oracle
    sp = synth.make_spectrum(512, 2, seed=5)

    def evaluate(X):
        return c_oracle.objective_batch(X, sp["w"], sp["u"], sp["v"], sp["weights"], threads=1)
    sw = swarm_support.HostSwarm(evaluate, sp["lower"], sp["upper"], swarmsize=S, seed=seed, minfunc=-1.0, minstep=-1.0)
    x, f = pso.run_sharded(sw, pso.LocalExchange(), maxiter=maxiter)
    return x, f, sw


@pytest.mark.parametrize("world,S", [(2, 20), (3, 17)])
def test_sharded_swarm_over_gloo_equals_single_rank(tmp_path, world, S):
    import torch.multiprocessing as mp
    port = _free_port()
    maxiter, seed = 15, 4242
    mp.spawn(_worker, args=(world, port, S, maxiter, seed, str(tmp_path)), nprocs=world, join=True)
    x1, f1, sw1 = _single(S, maxiter, seed)
    xs = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        np.testing.assert_array_equal(d["x"], x1)        # every rank holds the same answer
        assert float(d["f"]) == f1
        np.testing.assert_array_equal(d["g"], sw1.g)
        assert int(d["it"]) == maxiter == sw1.iteration
        xs.append(d["xs"])
    np.testing.assert_array_equal(np.concatenate(xs), sw1.x)   # the shards tile the one-rank swarm

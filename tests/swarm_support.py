"""
Test infrastructure for the swarm loop (NOT part of the nmrfit_amd package; moved out of nmrfit_amd/pso.py in round 5):

``HostSwarm``       numpy mirror of the device swarm (csrc/pso.hip) with an injectable evaluator: what the GPU tests
                    compare DeviceSwarm with bit for bit, and what the CPU tests of the sharding logic run.
``SocketExchange``  the candidate record staged device -> host -> rank 0 -> every rank over nmrfit_amd.rendezvous
                    (standard-library sockets): several ranks on ONE GPU (RCCL refuses two ranks on a device), CPU tests.
``TorchExchange``   the same over a torch.distributed group (gloo): the world-size-2/3 CPU tests under
                    torch.distributed.run.

The product exchanges through RCCL inside libnmrfit_amd.so (nmrfit_amd.pso.RcclExchange).
"""
import numpy as np

from nmrfit_amd.pso import DEFAULTS, uniform2  # noqa: F401


class SocketExchange:
    """Host-staged exchange over a ``rendezvous.Channel`` (standard-library sockets): the
    (D+1)-double record goes device -> host -> rank 0 -> every rank -> device.  For CPU tests of
    the sharding logic and for rehearsing several ranks on ONE GPU (RCCL refuses two ranks on
    the same device); multi-GPU fits use RcclExchange."""

    def __init__(self, channel=None):
        from nmrfit_amd import rendezvous
        self._own_channel = channel is None
        self.channel = rendezvous.Channel() if channel is None else channel
        self.rank, self.world = self.channel.rank, self.channel.world

    def gather_host(self, cand):
        cand = np.ascontiguousarray(cand, dtype=np.float64)
        parts = self.channel.all_gather(cand.tobytes())
        return np.stack([np.frombuffer(p, dtype=np.float64) for p in parts])

    def broadcast_seed(self, seed):
        import struct
        return struct.unpack("<Q", self.channel.broadcast(struct.pack("<Q", int(seed) & 0xFFFFFFFFFFFFFFFF)))[0]

    def barrier(self):
        self.channel.barrier()

    def all_reduce(self, values, op="max"):
        a = np.array(values, dtype=np.float64).reshape(-1)
        parts = np.stack([np.frombuffer(p, dtype=np.float64) for p in self.channel.all_gather(a.tobytes())])
        return {"sum": parts.sum, "max": parts.max, "min": parts.min}[op](axis=0)

    def close(self):
        if self._own_channel and self.channel is not None:
            self.channel.close()
            self.channel = None


class TorchExchange:
    """The same host-staged exchange over a torch.distributed group (gloo): kept for the CPU
    tests that rehearse the N>1 logic under ``torch.distributed.run``.  Not used by the product
    path -- multi-GPU fits exchange through RcclExchange without importing torch."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def gather_host(self, cand):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(cand))
        out = torch.empty(self.world * t.numel(), dtype=t.dtype)   # flat: gloo needs 1-D
        self._dist.all_gather_into_tensor(out, t, group=self.group)
        return out.view(self.world, t.numel()).numpy()

    def broadcast_seed(self, seed):
        import torch
        t = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64)
        self._dist.broadcast(t, src=0, group=self.group)
        return int(t.item())


# ---- numpy mirror -----------------------------------------------------------------------------
class HostSwarm:
    """numpy mirror of the swarm kernels (csrc/pso.hip, pso_update.h) with an injectable ``evaluate(X) -> f``: same
    Philox draws, same update arithmetic, same acceptance / stopping rule -- what the device swarm is compared with bit
    for bit.  NOT a fallback for DeviceSwarm: nothing in the product selects it."""

    def __init__(self, evaluate, lower, upper, swarmsize, offset=0, S_local=None, seed=0,
                 omega=DEFAULTS["omega"], phip=DEFAULTS["phip"], phig=DEFAULTS["phig"],
                 minstep=DEFAULTS["minstep"], minfunc=DEFAULTS["minfunc"]):
        self.lb = np.array(lower, dtype=np.float64)
        self.ub = np.array(upper, dtype=np.float64)
        assert len(self.lb) == len(self.ub), 'Lower- and upper-bounds must be the same length'
        assert np.all(self.ub > self.lb), 'All upper-bound values must be greater than lower-bound values'
        self.evaluate = evaluate
        self.S_global = int(swarmsize)
        self.offset = int(offset)
        self.S = self.S_global if S_local is None else int(S_local)
        self.D = self.lb.size
        self.seed = int(seed)
        self.omega, self.phip, self.phig = omega, phip, phig
        self.minstep, self.minfunc = minstep, minfunc
        self.iteration, self.stop = 0, 0
        self.fg = np.inf
        self.g = np.zeros(self.D)
        self.best_x, self.best_f = np.zeros(self.D), np.inf

    def _select(self):
        if self.S:
            self.fx = np.asarray(self.evaluate(self.x), dtype=np.float64)
            upd = self.fx < self.fp
            self.p[upd, :] = self.x[upd, :]
            self.fp[upd] = self.fx[upd]
            i = int(np.argmin(self.fp))
            # pyswarm seeds g with p[argmin fp] -- or, while no particle has a finite objective yet (every
            # fp still +inf, argmin 0), with x[0]: the record then carries this shard's first position, and
            # the fold's lowest-rank tie-break makes it GLOBAL particle 0's
            row = self.p[i, :] if self.fp[i] < np.inf else self.x[i, :]
            self.cand = np.concatenate(([self.fp[i]], row))
        else:
            self.fx = np.zeros(0)
            self.cand = np.concatenate(([np.inf], np.zeros(self.D)))

    def init(self):
        r0, r1 = uniform2(self.seed, 0, self.S, self.D, self.offset)
        vhigh = np.abs(self.ub - self.lb)
        vlow = -vhigh
        self.x = self.lb + r0 * (self.ub - self.lb)
        self.v = vlow + r1 * (vhigh - vlow)
        self.p = np.zeros_like(self.x)
        self.fp = np.full(self.S, np.inf)
        self.iteration, self.stop, self._seeded = 0, 0, False
        self._select()

    def step_local(self):
        if self.stop:
            return
        rp, rg = uniform2(self.seed, self.iteration + 1, self.S, self.D, self.offset)
        self.v = (self.omega * self.v + (self.phip * rp) * (self.p - self.x)) + (self.phig * rg) * (self.g - self.x)
        x = self.x + self.v
        x = np.where(x < self.lb, self.lb, x)
        x = np.where(x > self.ub, self.ub, x)
        self.x = x
        self._select()

    def candidate(self):
        return self.cand

    def apply_global(self, cands):
        if self.stop:
            return
        cands = np.asarray(cands, dtype=np.float64).reshape(-1, self.D + 1)
        win = int(np.argmin(cands[:, 0]))          # first minimum: lowest rank wins ties
        fc, pc = cands[win, 0], cands[win, 1:]
        if not self._seeded:
            self.g, self.fg = pc.copy(), fc
            self.best_x, self.best_f = pc.copy(), fc
            self._seeded = True
            self.iteration = 0
            return
        if fc < self.fg:
            stepsize = np.sqrt(np.sum((self.g - pc) ** 2))
            if np.abs(self.fg - fc) <= self.minfunc:
                self.stop, self.best_x, self.best_f = 1, pc.copy(), fc
            elif stepsize <= self.minstep:
                self.stop, self.best_x, self.best_f = 2, pc.copy(), fc
            else:
                self.g, self.fg = pc.copy(), fc
                self.best_x, self.best_f = pc.copy(), fc
        self.iteration += 1

    def status(self):
        return dict(iteration=self.iteration, stop=self.stop, fg=self.fg)

    def best(self):
        return self.best_x.copy(), float(self.best_f)



"""
Batched particle-swarm driver: the replacement for ``pyswarm.pso`` as the reference calls it
(nmrfit/utils.py:176-182).  pyswarm (github.com/tisimst/pyswarm, unpinned master; not
vendored in the reference) evaluates one particle per Python call; here a generation is one
batched objective launch and the swarm state never leaves the GPU.

``DeviceSwarm`` is the product path: the nmrfit_pso_* entry points of libnmrfit_amd.so (csrc/pso.hip), state in HBM,
Philox4x32-10 counter RNG on the device.  (Its numpy mirror and the host-staged exchanges that the CPU tests and
one-GPU rehearsals drive it with live in tests/swarm_support.py: test infrastructure, not part of this package.)

Sharding: rank q of G owns particles [offset, offset+S_local); the
only cross-rank traffic is one all-gather of a (D+1)-double candidate record per generation
(``RcclExchange``: ncclAllGather inside libnmrfit_amd.so, on the kernels' stream, no Python in
the generation loop).  Random numbers are
a function of (seed, generation, dimension, GLOBAL particle index), so the trajectory does
not depend on the number of ranks.

Stopping rule, restated from pyswarm: with (fc, pc) the best personal best after a
generation, if fc < fg: stop when |fg - fc| <= minfunc, else stop when |g - pc| <= minstep,
else accept (g, fg) = (pc, fc).  On a stop pyswarm returns (pc, fc).
"""
import ctypes

import numpy as np

from . import _cabi

DEFAULTS = dict(swarmsize=204, maxiter=2000, omega=-0.2134, phip=-0.3344, phig=2.3259,   # utils.py:177-181
                minstep=1e-8, minfunc=1e-8)                                                 # pyswarm defaults

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (counter words c0..c3 as uint64 arrays holding 32-bit values)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & _MASK
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniform2(seed, gen, S, D, offset):
    """Two U[0,1) matrices [S, D] for generation ``gen`` -- identical to csrc/pso.hip uniform2."""
    part = (np.arange(S, dtype=np.uint64) + np.uint64(offset))[:, None] + np.zeros((1, D), dtype=np.uint64)
    dim = np.zeros((S, 1), dtype=np.uint64) + np.arange(D, dtype=np.uint64)[None, :]
    gen_a = np.full((S, D), gen, dtype=np.uint64)
    o0, o1, o2, o3 = philox4x32_10(gen_a, dim, part & _MASK, part >> np.uint64(32),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    ua = (o1 << np.uint64(32)) | o0
    ub = (o3 << np.uint64(32)) | o2
    scale = 2.0 ** -53
    return (ua >> np.uint64(11)).astype(np.float64) * scale, (ub >> np.uint64(11)).astype(np.float64) * scale


def shard(S_global, rank, world):
    """Contiguous split of the swarm axis; the first S_global % world ranks get one extra."""
    base, extra = divmod(S_global, world)
    n = base + (1 if rank < extra else 0)
    off = rank * base + min(rank, extra)
    return off, n


# ---- candidate exchange ---------------------------------------------------------------------
class LocalExchange:
    """Single rank: the gathered set is the rank's own candidate."""
    world = 1
    rank = 0

    def gather_host(self, cand):
        return cand[None, :]

    def broadcast_seed(self, seed):
        return int(seed)


class RcclExchange:
    """The product exchange for multi-GPU fits: an RCCL communicator created through the C-ABI
    (``nmrfit_comm_*``, csrc/comm.hip).  Attached to a DeviceSwarm, every generation's
    ncclAllGather of the (D+1)-double record and the fold run inside ``nmrfit_pso_step`` on the
    context's HIP stream -- no host synchronisation, no Python, no PyTorch in the loop.

    ``channel`` is a ``rendezvous.Channel`` (or anything with rank / world / broadcast /
    all_gather) used before RCCL exists: the ranks first tell each other whether RCCL can be
    loaded at all (so that a rank without it is an error on EVERY rank, not a hang of the others
    inside the collective ncclCommInitRank), then rank 0's 128-byte unique id goes round.
    Collective: every rank constructs it, with its own Evaluator (one GPU per process).

    ``init_timeout`` (seconds; default NMRFIT_COMM_INIT_TIMEOUT or 300, 0 disables): deadline
    for the communicator creation, the one step that cannot time out by itself -- on expiry the
    process says on stderr which rank was stuck on which device and exits (rendezvous.Watchdog),
    so that the launcher sees a failed rank.  ``verbose``: one stderr line per rank once the
    communicator exists (rank, HIP device, PCI bus id, RCCL version)."""

    def __init__(self, evaluator, channel=None, init_timeout=None, verbose=False):
        import os
        import sys
        from . import rendezvous
        # first contact with a new node is where RCCL fails if it fails: let it say why
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        # every rank on this node (the launcher says MASTER_ADDR is loopback): RCCL's bootstrap sockets
        # may as well use the interface the rendezvous already works over, whatever else the container has
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        self._lib = _cabi.lib()
        self.ev = evaluator
        self._own_channel = channel is None
        self.channel = rendezvous.Channel() if channel is None else channel
        self.rank, self.world = self.channel.rank, self.channel.world
        if init_timeout is None:
            init_timeout = float(os.environ.get("NMRFIT_COMM_INIT_TIMEOUT", "300"))
        # 1. can every rank load RCCL?  (dlopen + symbols only: nothing collective yet)
        rc_av = self._lib.nmrfit_comm_available()
        msg_av = "" if rc_av == _cabi.OK else self._lib.nmrfit_last_error().decode("utf-8", "replace")
        notes = self.channel.all_gather(msg_av.encode("utf-8", "replace") if rc_av != _cabi.OK else b"")
        bad = [(r, n.decode("utf-8", "replace")) for r, n in enumerate(notes) if n]
        if bad:
            self._close_channel()
            raise _cabi.NmrfitError(_cabi.E_UNSUPPORTED, "RCCL is not available on rank(s) %s: %s"
                                    % ([r for r, _ in bad], bad[0][1]))
        uid = ctypes.create_string_buffer(_cabi.UNIQUE_ID_BYTES)
        rc0, err0 = 0, ""
        if self.rank == 0:
            rc0 = self._lib.nmrfit_comm_unique_id(uid)
            if rc0 != _cabi.OK:
                err0 = self._lib.nmrfit_last_error().decode("utf-8", "replace")
        # rank 0 ALWAYS answers (an empty id when it could not make one), so that no rank is left
        # waiting in the rendezvous for an id that will never come
        raw = self.channel.broadcast((uid.raw if rc0 == _cabi.OK else b"") if self.rank == 0 else b"")
        if len(raw) != _cabi.UNIQUE_ID_BYTES:
            self._close_channel()
            if self.rank == 0:
                raise _cabi.NmrfitError(rc0, err0)
            raise _cabi.NmrfitError(_cabi.E_COMM, "rank 0 could not create an RCCL unique id")
        uid = ctypes.create_string_buffer(raw, _cabi.UNIQUE_ID_BYTES)
        self._h = ctypes.c_void_p()
        dev = getattr(evaluator, "device", -1)

        try:      # looked up now: the watchdog thread must not make HIP calls while this one is stuck in one
            pci = _cabi.device_pci_bus_id(dev)
        except Exception:
            pci = "unknown"

        def where():
            return "(HIP device %s, PCI %s, world %d)" % (dev, pci, self.world)
        # 2. the collective creation, under a deadline
        with rendezvous.Watchdog(init_timeout, "ncclCommInitRank (nmrfit_comm_create)", rank=self.rank, describe=where):
            rc = self._lib.nmrfit_comm_create(evaluator.handle, self.rank, self.world, uid, ctypes.byref(self._h))
        if rc != _cabi.OK:
            self._close_channel()
            _cabi.check(rc)
        evaluator._children.add(self)
        if verbose:
            sys.stderr.write("nmrfit: RCCL communicator ready: %s\n" % self.describe())
            sys.stderr.flush()

    def _close_channel(self):
        if getattr(self, "_own_channel", False) and self.channel is not None:
            self.channel.close()
            self.channel = None

    def describe(self):
        buf = ctypes.create_string_buffer(256)
        _cabi.check(self._lib.nmrfit_comm_describe(self._h, buf, 256))
        return buf.value.decode("utf-8", "replace")

    @property
    def handle(self):
        return self._h

    def info(self):
        r, n, v = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _cabi.check(self._lib.nmrfit_comm_info(self._h, ctypes.byref(r), ctypes.byref(n), ctypes.byref(v)))
        return dict(rank=r.value, world=n.value, rccl_version=v.value)

    def barrier(self):
        _cabi.check(self._lib.nmrfit_comm_barrier(self._h))

    def all_reduce(self, values, op="max"):
        a = np.array(values, dtype=np.float64).reshape(-1)
        _cabi.check(self._lib.nmrfit_comm_all_reduce_host(self._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                          a.size, {"sum": 0, "max": 1, "min": 2}[op]))
        return a

    def broadcast_seed(self, seed):
        buf = (ctypes.c_uint64 * 1)(int(seed) & 0xFFFFFFFFFFFFFFFF)
        _cabi.check(self._lib.nmrfit_comm_broadcast_host(self._h, buf, 8, 0))
        return int(buf[0])

    def gather_host(self, cand):
        """Host-array form of the exchange (tests): through device buffers and ncclAllGather."""
        cand = _cabi.f64(cand)
        n = cand.size
        d_s, d_r = self.ev.dev_alloc(n * 8), self.ev.dev_alloc(n * 8 * self.world)
        try:
            self.ev.upload(d_s, cand)
            _cabi.check(self._lib.nmrfit_comm_all_gather_dev(self._h, d_s, d_r, n))
            return self.ev.download(d_r, (self.world, n))
        finally:
            self.ev.dev_free(d_s)
            self.ev.dev_free(d_r)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _cabi.check(self._lib.nmrfit_comm_destroy(self._h))   # E_STATE while a swarm is still attached
            self._h = ctypes.c_void_p()
        self._close_channel()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the product path ---------------------------------------------------------------------------
class DeviceSwarm:
    """Device-resident swarm over an ``equations.Evaluator`` (csrc/pso.hip)."""

    def __init__(self, evaluator, lower, upper, swarmsize, offset=0, S_local=None, seed=0,
                 omega=DEFAULTS["omega"], phip=DEFAULTS["phip"], phig=DEFAULTS["phig"],
                 minstep=DEFAULTS["minstep"], minfunc=DEFAULTS["minfunc"]):
        self._lib = _cabi.lib()
        self.ev = evaluator
        lb, ub = _cabi.f64(lower), _cabi.f64(upper)
        assert len(lb) == len(ub), 'Lower- and upper-bounds must be the same length'
        assert np.all(ub > lb), 'All upper-bound values must be greater than lower-bound values'
        self.D = int(lb.size)
        if self.D < 4 or (self.D - 4) % 3:
            raise ValueError("bounds must have 4 + 3P entries")
        self.P = (self.D - 4) // 3
        self.S_global = int(swarmsize)
        self.offset = int(offset)
        self.S = self.S_global if S_local is None else int(S_local)
        self._minstep, self._minfunc = minstep, minfunc
        prm = _cabi.PsoParams(omega, phip, phig, minstep, minfunc, int(seed) & 0xFFFFFFFFFFFFFFFF)
        self._h = ctypes.c_void_p()
        _cabi.check(self._lib.nmrfit_pso_create(evaluator.handle, self.S, self.S_global, self.offset, self.P,
                                                _cabi.ptr(lb), _cabi.ptr(ub), ctypes.byref(prm),
                                                ctypes.byref(self._h)))
        evaluator._children.add(self)

    def close(self):
        if getattr(self, "_comm", None) is not None and getattr(self, "_h", None) is not None and self._h.value:
            try:
                self.set_comm(None)
            except Exception:
                pass
        if getattr(self, "_d_gather", None) is not None:
            try:
                self.ev.dev_free(self._d_gather)
            except Exception:
                pass
            self._d_gather = None
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nmrfit_pso_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init(self):
        _cabi.check(self._lib.nmrfit_pso_init(self._h))

    def step_local(self):
        _cabi.check(self._lib.nmrfit_pso_step_local(self._h))

    def set_comm(self, exchange):
        """Attach an RcclExchange (None detaches): ``step`` and ``run`` then include the
        all-gather of the candidate records and the fold, all enqueued by one C call."""
        _cabi.check(self._lib.nmrfit_pso_set_comm(self._h, exchange.handle if exchange is not None else None))
        self._comm = exchange

    def step(self):
        """One whole generation in one C call (update, objective, personal bests, candidate,
        exchange over the attached communicator, fold).  The first call after ``init`` only
        folds generation 0."""
        _cabi.check(self._lib.nmrfit_pso_step(self._h))

    def set_handover(self, mode):
        """How the personal-best / argmin kernel's workgroups hand their results to the one that finishes (swarms of up
        to 1024 particles whose generation is not finished inside the objective launch): "two_launch" (default since
        round 5: nothing handed over inside a launch), "fast" (fence-free agent-scope stores) or "fenced" (release /
        acquire fences).  Bit-identical results; an A/B knob."""
        if isinstance(mode, str):
            mode = {"fast": _cabi.HANDOVER_FAST, "fenced": _cabi.HANDOVER_FENCED,
                    "two_launch": _cabi.HANDOVER_TWO_LAUNCH}[mode.lower()]
        _cabi.check(self._lib.nmrfit_pso_set_handover(self._h, int(mode)))

    def set_fused_pbest(self, enable=True):
        """Personal bests inside the objective launch when one workgroup holds a whole particle (default on;
        A/B knob: off restores the separate personal-best / argmin kernel; bit-identical results)."""
        _cabi.check(self._lib.nmrfit_pso_set_fused_pbest(self._h, 1 if enable else 0))

    def set_fused_tail(self, enable=True):
        """The whole generation as one launch (single rank, up to 1024 particles, one workgroup per particle): the
        objective launch ends with the personal bests and the fold is deferred into the next launch's prologue
        (csrc/pso_update.h).  Default on; off restores the separate one-workgroup launch for the candidate record
        and the fold (A/B knob, bit-identical results)."""
        _cabi.check(self._lib.nmrfit_pso_set_fused_tail(self._h, 1 if enable else 0))

    def last_launches(self):
        """Kernel launches of the last generation's evaluate-and-select part (1, 2 or 3)."""
        n = ctypes.c_int32(0)
        _cabi.check(self._lib.nmrfit_pso_last_launches(self._h, ctypes.byref(n)))
        return n.value

    def candidate_dev(self):
        p = ctypes.c_void_p()
        _cabi.check(self._lib.nmrfit_pso_candidate_dev(self._h, ctypes.byref(p)))
        return p

    def set_candidate_dev(self, dptr):
        _cabi.check(self._lib.nmrfit_pso_set_candidate_dev(self._h, ctypes.c_void_p(dptr) if dptr else None))

    @property
    def minfunc(self):
        return self._minfunc

    @property
    def minstep(self):
        return self._minstep

    def candidate(self):
        return self.ev.download(self.candidate_dev(), (self.D + 1,))

    def apply_global_dev(self, d_cands, nranks):
        _cabi.check(self._lib.nmrfit_pso_apply_global_dev(self._h, d_cands, int(nranks)))

    def apply_global(self, cands):
        """Host-array form: uploads the gathered records (used with gloo / tests)."""
        cands = _cabi.f64(cands).reshape(-1, self.D + 1)
        if getattr(self, "_d_gather", None) is None or self._gather_rows < cands.shape[0]:
            if getattr(self, "_d_gather", None) is not None:
                self.ev.dev_free(self._d_gather)
            self._d_gather = self.ev.dev_alloc(cands.nbytes)
            self._gather_rows = cands.shape[0]
        self.ev.upload(self._d_gather, cands)
        self.apply_global_dev(self._d_gather, cands.shape[0])

    def status(self):
        it, stop, fg = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_double(0.0)
        _cabi.check(self._lib.nmrfit_pso_status(self._h, ctypes.byref(it), ctypes.byref(stop), ctypes.byref(fg)))
        return dict(iteration=it.value, stop=stop.value, fg=fg.value)

    def best(self):
        x = np.empty(self.D)
        f = ctypes.c_double(0.0)
        _cabi.check(self._lib.nmrfit_pso_best(self._h, _cabi.ptr(x), ctypes.byref(f)))
        return x, f.value

    def run(self, maxiter, check_every=64):
        _cabi.check(self._lib.nmrfit_pso_run(self._h, int(maxiter), int(check_every)))

    def state(self):
        x = np.empty((self.S, self.D)); v = np.empty_like(x); p = np.empty_like(x)
        fx = np.empty(self.S); fp = np.empty(self.S)
        _cabi.check(self._lib.nmrfit_pso_get_state(self._h, _cabi.ptr(x), _cabi.ptr(v), _cabi.ptr(p),
                                                   _cabi.ptr(fx), _cabi.ptr(fp)))
        return dict(x=x, v=v, p=p, fx=fx, fp=fp)


STOP_MESSAGES = {
    1: 'Stopping search: Swarm best objective change less than {minfunc}',
    2: 'Stopping search: Swarm best position change less than {minstep}',
}


def run_sharded(swarm, exchange, maxiter, check_every=1, verbose=False):
    """Generation loop for a (possibly sharded) swarm; every rank must call it.  Returns
    (x_best, f_best).  A DeviceSwarm with an RcclExchange runs entirely inside the library
    (``nmrfit_pso_run`` with the communicator attached: one ncclAllGather per generation on
    the kernels' stream); with any other exchange object (``gather_host(cand) -> rows``, ``rank``: the host-staged
    exchanges of tests/swarm_support.py) the (D+1)-double record is staged through the host."""
    if isinstance(swarm, DeviceSwarm) and isinstance(exchange, RcclExchange):
        swarm.set_comm(exchange)
        try:
            swarm.run(maxiter, check_every)
            stopped = swarm.status()["stop"]
            best = swarm.best()
        finally:
            swarm.set_comm(None)
    else:
        swarm.init()
        swarm.apply_global(exchange.gather_host(swarm.candidate()))
        it = 0
        stopped = 0
        while it < maxiter:
            it += 1
            swarm.step_local()
            swarm.apply_global(exchange.gather_host(swarm.candidate()))
            if it % check_every == 0 or it == maxiter:
                stopped = swarm.status()["stop"]
                if stopped:
                    break
        best = swarm.best()
    if verbose and exchange.rank == 0:
        if stopped:
            print(STOP_MESSAGES[stopped].format(minfunc=getattr(swarm, "minfunc", 1e-8),
                                                minstep=getattr(swarm, "minstep", 1e-8)))
        else:
            print('Stopping search: maximum iterations reached --> {:}'.format(maxiter))
    return best


def pso(evaluator, lb, ub, swarmsize=100, omega=0.5, phip=0.5, phig=0.5, maxiter=100, minstep=1e-8,
        minfunc=1e-8, seed=0, check_every=64, verbose=True):
    """pyswarm.pso-shaped entry point over a GPU ``Evaluator`` (single rank):
    returns (xopt, fopt) like pyswarm does."""
    sw = DeviceSwarm(evaluator, lb, ub, swarmsize, seed=seed, omega=omega, phip=phip, phig=phig,
                     minstep=minstep, minfunc=minfunc)
    try:
        sw.run(maxiter, check_every)
        st = sw.status()
        if verbose:
            if st["stop"]:
                print(STOP_MESSAGES[st["stop"]].format(minfunc=minfunc, minstep=minstep))
            else:
                print('Stopping search: maximum iterations reached --> {:}'.format(maxiter))
        return sw.best()
    finally:
        sw.close()

"""
Process bootstrap for a sharded swarm, standard library only (no torch, no MPI).

One process per GPU (launched by ``python -m torch.distributed.run``, by ``bench.py --gpus N``
itself, or by anything else that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR /
MASTER_PORT).  The ranks need exactly two things from each other before RCCL takes over:
the 128-byte RCCL unique id that rank 0 generates (``nmrfit_comm_unique_id`` ->
``nmrfit_comm_create``, include/nmrfit_amd.h) and, for CPU tests and one-GPU rehearsals, a
host-staged all-gather of the (D+1)-double candidate record.  Both go over a star of TCP
connections to rank 0.

Where rank 0 listens:

* ``NMRFIT_RDZV_PORT`` set: on MASTER_ADDR (the address rank 0 binds -- not every interface)
  at that port (multi-node capable; the port must be free, so it cannot be MASTER_PORT under
  torchrun, whose agent keeps its own store listening there);
* otherwise (one node, the default): on an ephemeral port of 127.0.0.1, published through a
  small file in the temporary directory whose name is derived from MASTER_ADDR, MASTER_PORT
  and the launch token.  No fixed port is needed, so nothing can collide with the launcher's
  own rendezvous.

The launch token is what the ranks of ONE launch share and nobody else knows:
``NMRFIT_RDZV_TOKEN`` if set; else, in the one-node mode, the parent process id (torchrun's
agent, or bench.py's launcher, is the parent of every rank); else, with ``NMRFIT_RDZV_PORT``
(ranks on several nodes have different parents), a digest of the launch-wide values
MASTER_ADDR, MASTER_PORT, WORLD_SIZE and TORCHELASTIC_RUN_ID -- guessable by anyone who can
read the job's environment, so export ``NMRFIT_RDZV_TOKEN`` (any shared secret) on a network
you do not trust.

A rank announces itself with a magic word, the token and its rank number; rank 0 rejects
anything else, so a stray connection cannot join the group.

``Watchdog`` is the deadline for the steps that cannot time out by themselves (a collective
``ncclCommInitRank`` in which some rank never arrives): on expiry it says on stderr which rank
was where and ends the process, so that the launcher sees a failed rank instead of a hang.
"""
import hashlib
import hmac
import os
import socket
import struct
import sys
import tempfile
import threading
import time

_MAGIC = b"NMRFITv1"
DEFAULT_TIMEOUT = float(os.environ.get("NMRFIT_RDZV_TIMEOUT", "300"))


def env_rank_world():
    """(rank, local_rank, world) from the launcher's environment (defaults: a single rank)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            int(os.environ.get("WORLD_SIZE", "1")))


VISIBILITY_VARS = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")


def visible_devices_env(env=None):
    """The device-visibility variables that are set, as one string ("HIP_VISIBLE_DEVICES=3"), "" if none."""
    env = os.environ if env is None else env
    return " ".join("%s=%s" % (k, env[k]) for k in VISIBILITY_VARS if env.get(k) is not None)


def pick_device(device_count, local_rank=None, local_world=None, env=None):
    """Which HIP device a rank of a one-process-per-GPU launch works on -> (device, note).

    The reference's parallel mode hands particles to a process pool (nmrfit/utils.py:182); here it
    is one process per GPU, and launchers disagree about how a process finds its GPU:

    * every device visible to every rank (torch.distributed.run, bench.py's own launcher): the
      device is LOCAL_RANK;
    * one device visible per rank (a launcher that isolates each rank with HIP_VISIBLE_DEVICES /
      ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES, or with a device cgroup): the rank's only
      device is number 0 whatever LOCAL_RANK says.

    So: LOCAL_RANK when that index exists; device 0 when exactly one device is visible (the note
    says which variable isolated it, or that none is set -- then RCCL's own duplicate-device check
    is what catches a launcher that did NOT isolate the ranks); anything else is an error that
    names the variables, raised before the rendezvous so that every rank fails alike."""
    env = os.environ if env is None else env
    if local_rank is None:
        local_rank = int(env.get("LOCAL_RANK", env.get("RANK", "0")))
    if local_world is None:
        local_world = int(env.get("LOCAL_WORLD_SIZE", env.get("WORLD_SIZE", "1")))
    device_count, local_rank, local_world = int(device_count), int(local_rank), int(local_world)
    vis = visible_devices_env(env)
    if device_count < 1:
        raise RuntimeError("no HIP device is visible to local rank %d%s" % (local_rank, (" (%s)" % vis) if vis else ""))
    if 0 <= local_rank < device_count:
        return local_rank, ""
    if device_count == 1:
        if vis:
            return 0, "LOCAL_RANK=%d but one HIP device is visible (%s): this rank is isolated, using device 0" % (
                local_rank, vis)
        return 0, ("LOCAL_RANK=%d but one HIP device is visible and none of %s is set: assuming the launcher isolates "
                   "each rank's device some other way, using device 0 (if it does not, RCCL refuses the duplicate "
                   "device)" % (local_rank, " / ".join(VISIBILITY_VARS)))
    raise RuntimeError("LOCAL_RANK=%d of %d local rank(s) but %d HIP devices are visible (%s): expected either every "
                       "device visible to every rank (device = LOCAL_RANK) or exactly one per rank; check %s"
                       % (local_rank, local_world, device_count, vis or "no visibility variable set",
                          " / ".join(VISIBILITY_VARS)))


def _token():
    t = os.environ.get("NMRFIT_RDZV_TOKEN")
    if t:
        return t
    if os.environ.get("NMRFIT_RDZV_PORT"):
        # several nodes: the ranks' parents differ, the launcher's launch-wide values do not
        key = "|".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE",
                                                        "TORCHELASTIC_RUN_ID"))
        return "launch" + hashlib.sha256(key.encode()).hexdigest()[:32]
    return "ppid%d" % os.getppid()


class Watchdog:
    """Deadline for a step that cannot time out by itself.  ``with Watchdog(seconds, what):`` --
    if the block is still running after ``seconds`` the watchdog thread writes one line to
    stderr (rank, what it was waiting in, extra ``describe()`` text) and ends the process with
    ``exit_code`` (os._exit: the main thread is stuck inside a C call and cannot be unwound).
    ``seconds <= 0`` disables it.  ``phase`` may be updated while the block runs."""

    def __init__(self, seconds, phase, rank=None, exit_code=124, describe=None):
        self.seconds = float(seconds)
        self.phase = phase
        self.rank = env_rank_world()[0] if rank is None else rank
        self.exit_code = exit_code
        self.describe = describe
        self._done = threading.Event()
        self._thread = None

    def _run(self):
        if self._done.wait(self.seconds):
            return
        extra = ""
        try:
            extra = self.describe() if self.describe else ""
        except Exception as e:       # diagnostics must not hide the time-out
            extra = "(describe failed: %r)" % (e,)
        sys.stderr.write("nmrfit watchdog: rank %d still in `%s` after %.0f s%s -- ending this process (exit %d)\n"
                         % (self.rank, self.phase, self.seconds, (" " + extra) if extra else "", self.exit_code))
        sys.stderr.flush()
        os._exit(self.exit_code)

    def __enter__(self):
        if self.seconds > 0:
            self._thread = threading.Thread(target=self._run, name="nmrfit-watchdog", daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._done.set()
        if self._thread is not None:
            self._thread.join(1.0)
        return False


def _rdzv_file(token):
    key = "%s|%s|%s|%d" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"), token,
                           os.getuid())
    h = hashlib.sha256(key.encode()).hexdigest()[:24]
    return os.path.join(os.environ.get("NMRFIT_RDZV_DIR", tempfile.gettempdir()), "nmrfit_rdzv_%s" % h)


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf.extend(chunk)
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > (1 << 26):
        raise ConnectionError("rendezvous message too large")
    return _recv_exact(sock, n)


class Channel:
    """A star of TCP connections centred on rank 0: broadcast, all-gather and barrier of small
    byte strings.  Every rank must make the same sequence of calls."""

    def __init__(self, rank=None, world=None, timeout=DEFAULT_TIMEOUT):
        r, _, w = env_rank_world()
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        self.timeout = timeout
        self._peers = {}       # rank 0: rank -> socket
        self._sock = None      # other ranks: the connection to rank 0
        self._listener = None
        self._file = None
        if self.world > 1:
            self._connect()

    # -- set-up ------------------------------------------------------------------------------
    def _connect(self):
        token = _token()
        tok = token.encode()
        port_env = os.environ.get("NMRFIT_RDZV_PORT")
        deadline = time.monotonic() + self.timeout
        if self.rank == 0:
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            if port_env:
                # the address the other ranks connect to, not every interface of the node
                host = os.environ.get("MASTER_ADDR", "127.0.0.1")
                # A MASTER_ADDR hostname that rank 0's own /etc/hosts maps to loopback (Debian's 127.0.1.1
                # line) would make rank 0 listen where the ranks of other nodes cannot reach it: with ranks
                # on several nodes bind every interface then (the token handshake rejects strangers).
                try:
                    resolved = socket.gethostbyname(host)
                except OSError:
                    resolved = host
                multi_node = self.world > int(os.environ.get("LOCAL_WORLD_SIZE", self.world))
                if multi_node and resolved.startswith("127."):
                    # rank 0 cannot listen where MASTER_ADDR points: NMRFIT_RDZV_BIND names the address to bind; failing
                    # that every interface -- but then only behind a secret the launcher set (the derived token is made
                    # of values anybody on the network can guess)
                    bind_to = os.environ.get("NMRFIT_RDZV_BIND")
                    if bind_to:
                        host = bind_to
                    elif not os.environ.get("NMRFIT_RDZV_TOKEN"):
                        ls.close()
                        raise OSError("rendezvous: MASTER_ADDR=%s resolves to %s on rank 0 but the ranks span several "
                                      "nodes.  Set NMRFIT_RDZV_BIND to the address of the interface the other nodes "
                                      "reach, or NMRFIT_RDZV_TOKEN (a shared secret) to listen on every interface"
                                      % (host, resolved))
                    else:
                        host = ""
                    sys.stderr.write("nmrfit rendezvous: MASTER_ADDR=%s resolves to %s on rank 0 but the ranks span "
                                     "several nodes: listening on %s, port %s\n"
                                     % (os.environ.get("MASTER_ADDR"), resolved, host or "every interface", port_env))
                elif multi_node:
                    sys.stderr.write("nmrfit rendezvous: rank 0 listening on %s (%s) port %s\n" % (host, resolved, port_env))
                try:
                    ls.bind((host, int(port_env)))
                except OSError as e:
                    ls.close()
                    raise OSError("rendezvous: rank 0 cannot listen on MASTER_ADDR=%s port %s (%s); rank 0 must "
                                  "run on the MASTER_ADDR host and NMRFIT_RDZV_PORT must be free" % (host, port_env, e))
            else:
                ls.bind(("127.0.0.1", 0))
            ls.listen(max(16, self.world))
            self._listener = ls
            if not port_env:
                path = _rdzv_file(token)
                tmp = "%s.%d.tmp" % (path, os.getpid())
                with open(tmp, "w") as fh:
                    fh.write("127.0.0.1:%d\n" % ls.getsockname()[1])
                os.replace(tmp, path)        # atomic: a reader sees the whole line or no file
                self._file = path
            ls.settimeout(1.0)
            while len(self._peers) < self.world - 1:
                if time.monotonic() > deadline:
                    raise TimeoutError("rendezvous: %d of %d ranks joined within %.0f s"
                                       % (len(self._peers) + 1, self.world, self.timeout))
                try:
                    conn, _ = ls.accept()
                except socket.timeout:
                    continue
                try:
                    conn.settimeout(10.0)
                    hello = _recv(conn)
                    ok = hello.startswith(_MAGIC) and hmac.compare_digest(hello[len(_MAGIC) + 4:], tok)
                    peer = struct.unpack("<i", hello[len(_MAGIC):len(_MAGIC) + 4])[0] if ok else -1
                    if not ok or not (0 < peer < self.world) or peer in self._peers:
                        conn.close()
                        continue
                    conn.settimeout(self.timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._peers[peer] = conn
                except (OSError, struct.error, ConnectionError):
                    conn.close()
            if self._file:
                try:
                    os.unlink(self._file)
                except OSError:
                    pass
                self._file = None
        else:
            addr = None
            if port_env:
                addr = (os.environ.get("MASTER_ADDR", "127.0.0.1"), int(port_env))
            path = _rdzv_file(token)
            last_err = None
            while True:
                if time.monotonic() > deadline:
                    raise TimeoutError("rendezvous: rank %d could not reach rank 0 within %.0f s (%s)"
                                       % (self.rank, self.timeout, last_err))
                target = addr
                if target is None:
                    try:
                        with open(path) as fh:
                            host, port = fh.read().strip().rsplit(":", 1)
                        target = (host, int(port))
                    except (OSError, ValueError) as e:
                        last_err = e
                        time.sleep(0.05)
                        continue
                try:
                    s = socket.create_connection(target, timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send(s, _MAGIC + struct.pack("<i", self.rank) + tok)
                    s.settimeout(self.timeout)
                    self._sock = s
                    break
                except OSError as e:
                    last_err = e
                    time.sleep(0.1)
        self.barrier()     # everyone is connected (and a rejected rank finds out here)

    # -- collectives over the star ---------------------------------------------------------------
    def all_gather(self, payload):
        """list of every rank's byte string, in rank order, on every rank."""
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [None] * (self.world - 1)
            for r, s in self._peers.items():
                parts[r] = _recv(s)
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for s in self._peers.values():
                _send(s, blob)
            return parts
        _send(self._sock, payload)
        blob = _recv(self._sock)
        parts, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, off)
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def broadcast(self, payload, root=0):
        """root's byte string on every rank (the others pass anything, e.g. b"")."""
        return self.all_gather(payload if self.rank == root else b"")[root]

    def barrier(self):
        self.all_gather(b"")

    def close(self):
        for s in list(self._peers.values()) + [self._sock, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._sock, self._listener = {}, None, None
        if self._file:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

"""Row-sharded SGPR / SVGP across the GPUs of one node (new; the reference is single-process).

Every N-dependent quantity of the collapsed bound is a sum over rows, so rank g reduces its contiguous block
``X[lo:hi]`` to the packed statistics ``[Phi | psi | kappa | yy | n | n_whitened | n_parts]`` (M^2 + M + 5 doubles; the
last two count the shards summed in that whitened their rows / in total, so a sum of mixed-route shards is rejected) on
its own GPU and the only exchange is one sum-all-reduce of that vector -- RCCL reduce-scatter + all-gather over xGMI
inside ``liboak_hip`` -- plus the D + R + 2 scalars of the gradient record.  The O(M^3) tail is replicated, so every
rank sees bit-identical objective values and gradients and the BFGS iterations of ``oak_model.fit`` simply run replicated.

This module is the process-level plumbing, with no dependency beyond NumPy and the standard library:

* :class:`HostPlane` -- a small TCP star (rank 0 listens on ``MASTER_ADDR:MASTER_PORT``) for the control plane: broadcast of
  the RCCL unique id, barriers, and -- when the data exchange itself is set to ``"host"`` -- the sums;
* :class:`Communicator` / :func:`init_from_env` / :func:`current` -- what the model classes look at: under
  ``torchrun``-style environment variables (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT) ``gpflow_lite.SGPR`` /
  ``SVGP`` keep only their row shard on the device and attach the communicator to their context, ``oak_model.predict`` and
  ``get_sobol`` shard their work with one gather;
* :class:`ShardedSGPR`, :func:`sharded_predict`, :func:`sharded_sobol` -- the same pieces for callers of the raw C ABI.
"""
from __future__ import annotations

import os
import socket
import struct
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_bounds(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one row and tile [0, n_rows)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside [0, {world})")
    return (n_rows * rank) // world, (n_rows * (rank + 1)) // world


def stats_len(M: int) -> int:
    return M * M + M + 5


def pack_stats(Phi: np.ndarray, psi: np.ndarray, kappa: float, yy: float, n_rows: float, whitened: bool = False) -> np.ndarray:
    """One shard's packed vector (the layout of ``oak_sgpr_get_stats``)."""
    M = Phi.shape[0]
    out = np.empty(stats_len(M))
    out[:M * M] = np.asarray(Phi, dtype=np.float64).reshape(-1)
    out[M * M:M * M + M] = np.asarray(psi, dtype=np.float64).reshape(-1)
    out[M * M + M:] = (kappa, yy, n_rows, 1.0 if whitened else 0.0, 1.0)
    return out


def unpack_stats(packed: np.ndarray, M: int):
    """(Phi, psi, kappa, yy, n_rows) of a (summed) packed vector; raises if the sum mixes whitened and raw shards."""
    packed = np.asarray(packed, dtype=np.float64)
    if packed.size != stats_len(M):
        raise ValueError("packed statistics have the wrong length")
    n_white, n_parts = float(packed[-2]), float(packed[-1])
    if n_parts < 1 or n_white not in (0.0, n_parts):
        raise ValueError(f"packed statistics mix solve routes: {n_white:g} of {n_parts:g} shards whitened")
    return (packed[:M * M].reshape(M, M), packed[M * M:M * M + M].copy(), float(packed[M * M + M]),
            float(packed[M * M + M + 1]), float(packed[M * M + M + 2]))


def stats_whitened(packed: np.ndarray) -> bool:
    return float(np.asarray(packed)[-2]) > 0.0


def choose_route(n_total: int, M: int) -> str:
    """All ranks must agree on the solve route; decide it from GLOBAL sizes (same rule as the library's auto)."""
    return "whitened" if n_total * M <= (1 << 24) else "phi"


# ---------------------------------------------------------------------------------------------------------------------
# control plane
# ---------------------------------------------------------------------------------------------------------------------
MAX_FRAME = 1 << 33          # 8 GiB: no collective of this package moves more (a length prefix beyond it is a protocol error, not an allocation)
HELLO_MAGIC = b"OAKP"


def _send(sock: socket.socket, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv(sock: socket.socket, limit: int = MAX_FRAME) -> bytes:
    def exactly(n):
        chunks, got = [], 0
        while got < n:
            c = sock.recv(min(n - got, 1 << 20))
            if not c:
                raise ConnectionError("control plane: peer closed the connection")
            chunks.append(c); got += len(c)
        return b"".join(chunks)
    (n,) = struct.unpack("<Q", exactly(8))
    if n > limit:
        raise ConnectionError(f"control plane: frame of {n} bytes exceeds the limit of {limit}")
    return exactly(n)


def _job_token(world: int, port: int) -> bytes:
    """What a peer must present in its hello frame: $OAK_JOB_TOKEN if the launcher exported one (any shared secret), else a
    value derived from what every rank of THIS job already agrees on (run id, world size, port).  It keeps a stray or stale
    process that happens to hit the port from being taken for a rank; it is not an authentication scheme (the plane binds to
    the loopback interface unless told otherwise)."""
    import hashlib
    secret = os.environ.get("OAK_JOB_TOKEN") or "|".join([os.environ.get("TORCHELASTIC_RUN_ID", ""), str(world), str(port)])
    return hashlib.sha256(secret.encode()).digest()[:16]


class HostPlane:
    """TCP star over the ranks of one job: rank 0 listens, ranks 1..world-1 connect (retrying until ``timeout``).

    Collectives are two hops through rank 0 (gather, combine, scatter back): fine for a control plane -- a 128-byte id, a
    few scalars, a barrier -- and for the debug / single-GPU ``"host"`` data exchange; the production exchange is RCCL.
    Every rank receives the bytes rank 0 computed, so the results are bit-identical everywhere.

    Handshake: a peer sends ``OAKP | job token (16 bytes) | rank``; rank 0 keeps ONE deadline for the whole rendezvous, gives
    every accepted connection the remaining time to say hello, and drops (without failing the job) connections that send
    nothing, a wrong token, or a rank that is out of range or already taken.  Rank 0 binds the address it is given
    (MASTER_ADDR; 127.0.0.1 for a one-node job)."""

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29533, timeout: float = 120.0):
        self.rank, self.world = int(rank), int(world)
        self._peers: List[Optional[socket.socket]] = [None] * self.world
        self._root: Optional[socket.socket] = None
        if self.world == 1:
            return
        token = _job_token(self.world, port)
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world + 8)
            deadline = time.time() + timeout
            missing = self.world - 1
            try:
                while missing > 0:
                    left = deadline - time.time()
                    if left <= 0:
                        raise TimeoutError(f"control plane: {missing} of {self.world - 1} ranks did not join within {timeout:.0f} s")
                    srv.settimeout(left)
                    try:
                        conn, _a = srv.accept()
                    except socket.timeout:
                        continue
                    try:
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        conn.settimeout(max(0.05, min(10.0, deadline - time.time())))      # a silent connection cannot stall the rendezvous
                        hello = _recv(conn, limit=64)
                        ok = len(hello) == 24 and hello[:4] == HELLO_MAGIC and hello[4:20] == token
                        r = struct.unpack("<I", hello[20:24])[0] if ok else -1
                        if not ok or not (0 < r < self.world) or self._peers[r] is not None:
                            raise ConnectionError("not a rank of this job")
                    except (OSError, ConnectionError, struct.error):
                        conn.close()                                                       # dropped; keep waiting for the real ranks
                        continue
                    conn.settimeout(None)
                    self._peers[r] = conn
                    missing -= 1
            finally:
                srv.close()
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(None)
            _send(s, HELLO_MAGIC + token + struct.pack("<I", self.rank))
            self._root = s

    # -- primitives ------------------------------------------------------------------------------------------------
    def _through_root(self, payload: bytes, combine: Callable[[List[bytes]], bytes]) -> bytes:
        """Every rank contributes ``payload``; rank 0 combines the list (rank order) and everyone gets the result."""
        if self.world == 1:
            return combine([payload])
        if self.rank == 0:
            parts = [payload] + [_recv(self._peers[r]) for r in range(1, self.world)]
            out = combine(parts)
            for r in range(1, self.world):
                _send(self._peers[r], out)
            return out
        _send(self._root, payload)
        return _recv(self._root)

    def barrier(self):
        self._through_root(b"", lambda parts: b"")

    def broadcast(self, data: Optional[bytes] = None, src: int = 0) -> Optional[bytes]:
        """The byte string ``data`` of rank ``src`` on every rank (``None`` travels as ``None``).  Plain bytes, no pickling: what
        goes over this socket is never executed."""
        mine = b"\x00" if (self.rank != src or data is None) else b"\x01" + bytes(data)
        out = self._through_root(mine, lambda parts: parts[src])
        return None if out[:1] == b"\x00" else out[1:]

    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        """Sum over ranks of a float64 array, added in rank order on rank 0 (deterministic, identical on every rank)."""
        a = np.ascontiguousarray(a, dtype=np.float64)

        def combine(parts):
            acc = np.frombuffer(parts[0], dtype=np.float64).copy()
            for p in parts[1:]:
                acc += np.frombuffer(p, dtype=np.float64)
            return acc.tobytes()
        return np.frombuffer(self._through_root(a.tobytes(), combine), dtype=np.float64).reshape(a.shape).copy()

    def allgather(self, a: np.ndarray) -> List[np.ndarray]:
        """Every rank's 1-D float64 array (any lengths), in rank order."""
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)

        def combine(parts):
            return struct.pack(f"<{len(parts)}Q", *[len(p_) for p_ in parts]) + b"".join(parts)
        out = self._through_root(a.tobytes(), combine)
        lens = struct.unpack(f"<{self.world}Q", out[:8 * self.world])
        res, off = [], 8 * self.world
        for n in lens:
            res.append(np.frombuffer(out[off:off + n], dtype=np.float64).copy())
            off += n
        return res

    def close(self):
        for s in self._peers + [self._root]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._root = [None] * self.world, None


class Communicator:
    """What one rank knows about the job: its rank, the world size, the control plane, and how device contexts exchange
    their sums -- ``exchange="rccl"`` (reduce-scatter + all-gather over xGMI inside liboak_hip; one GPU per rank) or
    ``"host"`` (the same collectives through the control plane: several ranks on one GPU, no fabric, tests)."""

    def __init__(self, rank: int, world: int, plane: Optional[HostPlane] = None, exchange: str = "rccl"):
        if exchange not in ("rccl", "host"):
            raise ValueError("exchange must be 'rccl' or 'host'")
        self.rank, self.world, self.plane, self.exchange = int(rank), int(world), plane, exchange
        if self.world > 1 and plane is None:
            raise ValueError("a multi-rank communicator needs a control plane")

    @property
    def active(self) -> bool:
        return self.world > 1

    def bounds(self, n_rows: int) -> Tuple[int, int]:
        return shard_bounds(n_rows, self.rank, self.world)

    def attach(self, ctx):
        """Give a device context (oak._capi.HipContext) this job's communicator.  Collective: every rank calls it, for its
        contexts in the same order."""
        if not self.active:
            return
        if self.exchange == "host":
            ctx.comm_init_host(self.world, self.rank, self.plane.allreduce_sum)
            ctx._oak_comm_attached = self
            return
        # ncclCommInitRank is collective: first make sure every rank can load librccl at all, then ship rank 0's id
        ok, uid = 1, None
        try:
            ctx.comm_info()                                 # dlopen + version check of librccl on THIS rank
            if self.rank == 0:
                uid = ctx.comm_unique_id()
        except Exception:                                   # noqa: BLE001
            ok = 0
        oks = self.plane.allgather(np.array([float(ok)]))
        if min(int(x[0]) for x in oks) == 0:
            raise RuntimeError("RCCL is not usable on every rank (set the exchange to 'host' to run without it)")
        uid = self.plane.broadcast(uid, src=0)
        ctx.comm_init(uid, self.world, self.rank)
        ctx._oak_comm_attached = self

    def allgatherv(self, ctx, local: np.ndarray) -> np.ndarray:
        """Concatenation (rank order) of every rank's 1-D float64 block, through the context's communicator."""
        local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1)
        if not self.active:
            return local
        counts = [int(c[0]) for c in self.plane.allgather(np.array([float(local.size)]))]
        return ctx.comm_allgatherv(local, counts)


_current: Optional[Communicator] = None


def current() -> Optional[Communicator]:
    """The communicator model classes shard under (None or world 1: single process, nothing changes)."""
    return _current


def set_current(comm: Optional[Communicator]):
    global _current
    _current = comm


def init_from_env(exchange: Optional[str] = None, timeout: float = 120.0) -> Communicator:
    """Communicator from the launcher's environment (``torchrun`` / ``python -m torch.distributed.run`` / mpirun wrappers
    export RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT) and make it the current one.  The control plane listens
    on MASTER_PORT + 1 (the launcher's own store owns MASTER_PORT).  ``exchange`` defaults to $OAK_EXCHANGE or "rccl"."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    exchange = exchange or os.environ.get("OAK_EXCHANGE", "rccl")
    plane = None
    if world > 1:
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("OAK_CONTROL_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")          # one node: RCCL's bootstrap over loopback, data over xGMI
        plane = HostPlane(rank, world, addr, port, timeout)
    comm = Communicator(rank, world, plane, exchange)
    set_current(comm)
    return comm


def shutdown():
    """Close the control plane of the current communicator and forget it."""
    global _current
    if _current is not None and _current.plane is not None:
        _current.plane.close()
    _current = None


# ---------------------------------------------------------------------------------------------------------------------
# raw-ABI helpers
# ---------------------------------------------------------------------------------------------------------------------
class ShardedSGPR:
    """One rank's view of a row-sharded SGPR for callers of the raw binding.

    ``ctx`` is this rank's :class:`oak._capi.HipContext`, already attached to a communicator (``Communicator.attach``) --
    then ``elbo`` / ``elbo_grad`` are the library's fused entry points, whose sums are collective -- or not attached, with
    ``reducer`` (a callable summing a packed vector over the ranks) doing the exchange outside the library (forward only).
    """

    def __init__(self, ctx, X_local, y_local, Z, n_total: int, reducer: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                 route: Optional[str] = None):
        self.ctx, self.reducer = ctx, reducer
        self.M = int(np.asarray(Z).shape[0])
        self.n_total = int(n_total)
        ctx.sgpr_set_data(X_local, y_local)
        ctx.sgpr_set_inducing(Z)
        ctx.sgpr_set_global_rows(self.n_total)       # the library's own auto rule then also sees the global size
        ctx.sgpr_set_route(route or choose_route(self.n_total, self.M))

    def elbo(self, desc, noise_var: float, jitter: float = 1e-6) -> float:
        if self.reducer is None:
            return self.ctx.sgpr_elbo(desc, noise_var, jitter)       # local stats + all-reduce + tail
        self.ctx.sgpr_local_stats(desc, jitter)
        total = self.reducer(self.ctx.sgpr_get_stats())
        self.ctx.sgpr_set_stats(total, stats_whitened(total))    # set_stats rejects a sum of mixed-route shards
        e, _ = self.ctx.sgpr_tail(desc, noise_var, jitter)
        return e

    def elbo_grad(self, desc, noise_var: float, jitter: float = 1e-6):
        """(ELBO, gradient): the backward pass sums its per-shard record inside the library, so the context must carry a
        communicator (RCCL or host exchange); an outside ``reducer`` cannot serve it."""
        if self.reducer is not None:
            raise RuntimeError("ShardedSGPR.elbo_grad needs the context attached to a communicator (Communicator.attach); "
                               "a reducer outside the library only serves the forward statistics")
        return self.ctx.sgpr_elbo_grad(desc, noise_var, jitter)


def _gather_blocks(ctx, local: np.ndarray, comm: Optional[Communicator], gather):
    if gather is not None:                       # caller-supplied gather (tests): list of per-rank arrays
        return np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1) for p in gather(local)])
    comm = comm or current()
    if comm is None or not comm.active:
        return np.asarray(local, dtype=np.float64).reshape(-1)
    return comm.allgatherv(ctx, local)


def sharded_sobol(ctx, desc, Xc, alpha, subsets, rank: int, world: int, gather: Optional[Callable[[np.ndarray], list]] = None,
                  comm: Optional[Communicator] = None, **kwargs) -> np.ndarray:
    """All len(subsets) Sobol terms on every rank, in the order of ``subsets``.

    With the context attached to a communicator (``Communicator.attach``) this is ONE collective call of the library
    (``oak_sobol_collective``): the index-pair rows of the Gram of products -- or, for the per-term kernel, blocks of terms --
    are sharded over the ranks and summed on the device.  With a caller-supplied ``gather`` (a control plane outside the
    library) rank g evaluates the contiguous block ``subsets[lo:hi]`` and the scalars are gathered."""
    if gather is None:
        comm = comm or current()
        if comm is not None and comm.active:
            if getattr(ctx, "_oak_comm_attached", None) is not comm:
                raise RuntimeError("sharded_sobol: the context has not joined the communicator (Communicator.attach)")
            return np.asarray(ctx.sobol(desc, Xc, alpha, subsets, collective=True, **kwargs), dtype=np.float64)
        return np.asarray(ctx.sobol(desc, Xc, alpha, subsets, **kwargs), dtype=np.float64)
    lo, hi = shard_bounds(len(subsets), rank, world)
    local = np.asarray(ctx.sobol(desc, Xc, alpha, list(subsets[lo:hi]), **kwargs), dtype=np.float64) if hi > lo else np.empty(0)
    out = _gather_blocks(ctx, local, comm, gather)
    if out.size != len(subsets):
        raise RuntimeError("sharded_sobol: gathered blocks do not tile the term list")
    return out


def sharded_predict(ctx, desc, Xs, rank: int, world: int, gather: Optional[Callable[[np.ndarray], list]] = None,
                    comm: Optional[Communicator] = None):
    """Predictions are independent per test row: rank g predicts ``Xs[lo:hi]`` from its (replicated) posterior; mean and
    variance are gathered.  The caller has run ``elbo`` on every rank first (that leaves the posterior in each context)."""
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    lo, hi = shard_bounds(len(Xs), rank, world)
    if hi > lo:
        m, v = ctx.sgpr_predict(desc, Xs[lo:hi])
        local = np.stack([m, v], axis=1)
    else:
        local = np.empty((0, 2))
    out = _gather_blocks(ctx, local.reshape(-1), comm, gather).reshape(-1, 2)
    if out.shape[0] != len(Xs):
        raise RuntimeError("sharded_predict: gathered blocks do not tile the test rows")
    return out[:, 0].copy(), out[:, 1].copy()

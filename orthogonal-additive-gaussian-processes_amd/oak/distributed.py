"""Row-sharded SGPR across the GPUs of one node (new; the reference is single-process).

Every N-dependent quantity of the collapsed bound is a sum over rows, so rank g reduces its contiguous block
``X[lo:hi]`` to the packed statistics ``[Phi | psi | kappa | yy | n | n_whitened | n_parts]`` (M^2 + M + 5 doubles; the
last two count the shards summed in that whitened their rows / in total, so a sum of mixed-route shards is rejected) on its own GPU and the
only exchange is one sum-all-reduce of that vector -- RCCL reduce-scatter + all-gather over xGMI inside
``liboak_hip`` (``oak_comm_allreduce_stats``).  The O(M^3) tail is then replicated.  This module holds the
process-level plumbing: shard arithmetic, the packed layout, the communicator bootstrap (unique id broadcast over
the torch.distributed control plane) and a reducer interface so the exchange can be exercised on CPU with gloo.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

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


def torch_allreduce(packed: np.ndarray) -> np.ndarray:
    """Sum over the default torch.distributed group (gloo on CPU) -- the test/CPU stand-in for the RCCL exchange."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(packed, dtype=np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


def choose_route(n_total: int, M: int) -> str:
    """All ranks must agree on the solve route; decide it from GLOBAL sizes (same rule as the library's auto)."""
    return "whitened" if n_total * M <= (1 << 24) else "phi"


class ShardedSGPR:
    """One rank's view of a row-sharded SGPR.

    ``ctx`` is this rank's :class:`oak._capi.HipContext`.  With ``reducer=None`` the packed statistics are summed on
    the device by RCCL (``init_rccl`` must have been called); passing a callable (e.g. :func:`torch_allreduce`) routes
    the exchange through the host instead, which is how the N>1 logic is tested without GPUs' xGMI links.
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

    @staticmethod
    def init_rccl(ctx, rank: int, world: int, broadcast: Callable[[Optional[bytes]], bytes]):
        """Bootstrap the RCCL communicator: rank 0 draws the unique id, `broadcast` ships it to everyone."""
        uid = broadcast(ctx.comm_unique_id() if rank == 0 else None)
        ctx.comm_init(uid, world, rank)

    def elbo(self, desc, noise_var: float, jitter: float = 1e-6) -> float:
        if self.reducer is None:
            return self.ctx.sgpr_elbo(desc, noise_var, jitter)       # local stats + RCCL all-reduce + tail
        self.ctx.sgpr_local_stats(desc, jitter)
        total = self.reducer(self.ctx.sgpr_get_stats())
        self.ctx.sgpr_set_stats(total, stats_whitened(total))    # set_stats rejects a sum of mixed-route shards
        e, _ = self.ctx.sgpr_tail(desc, noise_var, jitter)
        return e


# ---- embarrassingly parallel pieces: no data-path collective, one gather of the results -------------------
def torch_allgather(local: np.ndarray) -> list:
    """Every rank's array, in rank order, over the default torch.distributed group (control plane, gloo)."""
    import torch.distributed as dist
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, np.ascontiguousarray(local, dtype=np.float64))
    return parts


def sharded_sobol(ctx, desc, Xc, alpha, subsets, rank: int, world: int, gather: Callable[[np.ndarray], list] = torch_allgather,
                  **kwargs) -> np.ndarray:
    """Sobol terms are independent: rank g evaluates the contiguous block ``subsets[lo:hi]`` on its GPU (``ctx.sobol``),
    the scalars are gathered.  Returns all len(subsets) values on every rank, in the order of ``subsets``."""
    lo, hi = shard_bounds(len(subsets), rank, world)
    local = np.asarray(ctx.sobol(desc, Xc, alpha, list(subsets[lo:hi]), **kwargs), dtype=np.float64) if hi > lo else np.empty(0)
    out = np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1) for p in gather(local)])
    if out.size != len(subsets):
        raise RuntimeError("sharded_sobol: gathered blocks do not tile the term list")
    return out


def sharded_predict(ctx, desc, Xs, rank: int, world: int, gather: Callable[[np.ndarray], list] = torch_allgather):
    """Predictions are independent per test row: rank g predicts ``Xs[lo:hi]`` from its (replicated) posterior; mean and
    variance are gathered.  The caller has run ``elbo`` on every rank first (that leaves the posterior in each context)."""
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    lo, hi = shard_bounds(len(Xs), rank, world)
    if hi > lo:
        m, v = ctx.sgpr_predict(desc, Xs[lo:hi])
        local = np.stack([m, v], axis=1)
    else:
        local = np.empty((0, 2))
    out = np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in gather(local)], axis=0)
    if out.shape[0] != len(Xs):
        raise RuntimeError("sharded_predict: gathered blocks do not tile the test rows")
    return out[:, 0].copy(), out[:, 1].copy()

"""Oracle-backed stand-in for ``oak._capi.HipContext`` -- TEST INFRASTRUCTURE ONLY.

There is no GPU in the CPU suite and the product has no CPU fallback, so the multi-process logic of the MODEL API
(``gpflow_lite.SGPR`` sharding its rows, ``oak_model.fit`` running BFGS replicated, sharded prediction and Sobol terms)
is exercised against this class instead: the same method names and argument meaning as the binding, the arithmetic from
``oracle/oak_oracle.py`` (statistics per row shard, summed through whatever host communicator was attached -- exactly the
seam the library has), gradients by central differences.  Continuous (RBF) sub-kernels only.  ``install()`` swaps it in.
"""
from __future__ import annotations

import copy

import numpy as np
import scipy.linalg as sla

from oracle import oak_oracle as o


class SpecDesc:
    """What the fake needs of a kernel description: the plain-data spec (oak._capi.KernelDesc keeps only the packed arrays)."""

    def __init__(self, spec):
        from oak import _capi
        self._real = _capi._RealKernelDesc(spec)              # the real class still validates the spec
        self.spec = copy.deepcopy(spec)
        self.D = self._real.D
        self.order_var = self._real.order_var
        self.cat_blocks = self._real.cat_blocks


class FakeContext:
    def __init__(self, device: int = 0):
        self.device = int(device)
        self._allreduce = None
        self._world, self._rank = 1, 0
        self._route = "auto"
        self._n_global = 0
        self._post = None

    # -- communicator ----------------------------------------------------------------------------------------------
    def comm_init_host(self, nranks, rank, allreduce):
        self._world, self._rank, self._allreduce = int(nranks), int(rank), allreduce

    def comm_rank(self):
        return self._rank

    def _sum(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a if self._allreduce is None else np.asarray(self._allreduce(a.reshape(-1).copy())).reshape(a.shape)

    def comm_allgatherv(self, local, counts):
        counts = [int(c) for c in counts]
        buf = np.zeros(sum(counts))
        off = sum(counts[:self._rank])
        buf[off:off + len(local)] = np.asarray(local, dtype=np.float64).reshape(-1)
        return self._sum(buf)

    def comm_destroy(self):
        self._allreduce, self._world, self._rank = None, 1, 0

    # -- SGPR ------------------------------------------------------------------------------------------------------
    def sgpr_set_data(self, X, Y):
        self.X, self.Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64).reshape(len(X), 1)
        self._sel = 0

    def sgpr_set_targets(self, y):
        y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
        if len(y) != len(self.X):
            raise ValueError("oak_sgpr_set_targets: row count differs from the data on the device")
        self.Y = np.concatenate([y, self.Y[:, 1:]], axis=1)

    def sgpr_set_extra_targets(self, Y_extra):
        if Y_extra is None or np.asarray(Y_extra).size == 0:
            self.Y = self.Y[:, :1]
            return
        Y_extra = np.asarray(Y_extra, dtype=np.float64).reshape(len(self.X), -1)
        self.Y = np.concatenate([self.Y[:, :1], Y_extra], axis=1)

    def sgpr_select_output(self, p):
        self._sel = int(p)

    def sgpr_set_inducing(self, Z):
        self.Z = np.asarray(Z, dtype=np.float64)

    def sgpr_set_route(self, route):
        self._route = route

    def sgpr_set_global_rows(self, n):
        self._n_global = int(n)

    def sgpr_stats_whitened(self):
        return False

    def sync(self):
        pass

    def close(self):
        pass

    def _elbo(self, spec, s2, jitter):
        M = len(self.Z)
        kuf = o.oak_K(spec, self.Z, self.X)
        P = self.Y.shape[1]
        yy_loc = (self.Y ** 2).sum(axis=0)
        local = np.concatenate([(kuf @ kuf.T).reshape(-1), (kuf @ self.Y).T.reshape(-1),
                                [float(o.oak_K_diag(spec, self.X).sum()), float(yy_loc[0]), float(len(self.X))], yy_loc[1:]])
        tot = self._sum(local)
        Phi, psi = tot[:M * M].reshape(M, M), tot[M * M:M * M + P * M].reshape(P, M).T
        kappa, yy0, n = tot[M * M + P * M:M * M + P * M + 3]
        yy = np.concatenate([[yy0], tot[M * M + P * M + 3:]])
        from oak import _capi
        try:
            L = np.linalg.cholesky(o.oak_K(spec, self.Z) + jitter * np.eye(M))
            W = sla.solve_triangular(L, sla.solve_triangular(L, Phi, lower=True).T, lower=True)
            LB = np.linalg.cholesky(np.eye(M) + W / s2)
        except np.linalg.LinAlgError as ex:                   # what the binding raises for OAK_E_NOTPD
            raise _capi.NotPositiveDefiniteError(str(ex)) from ex
        c = np.stack([sla.solve_triangular(LB, sla.solve_triangular(L, psi[:, p], lower=True), lower=True) / s2 for p in range(P)], axis=1)
        e = sum(-0.5 * n * np.log(2 * np.pi) - np.sum(np.log(np.diag(LB))) - 0.5 * n * np.log(s2) - 0.5 * yy[p] / s2
                + 0.5 * c[:, p] @ c[:, p] - 0.5 * kappa / s2 + 0.5 * np.trace(W) / s2 for p in range(P))       # one bound per output
        return float(e), (L, LB, c)

    def sgpr_elbo(self, desc, noise_var, jitter=1e-6):
        e, self._post = self._elbo(desc.spec, float(noise_var), jitter)
        self._sel = 0
        return e

    def grad_len(self, desc):
        return 2 * desc.D + desc.order_var.size + 1

    def sgpr_elbo_grad(self, desc, noise_var, jitter=1e-6):
        """Central differences in the constrained parameters, public layout [lengthscale (D) | base_var (D) | order_var | noise]."""
        spec, s2 = desc.spec, float(noise_var)
        e = self.sgpr_elbo(desc, s2, jitter)
        D, nov = desc.D, desc.order_var.size
        g = np.zeros(2 * D + nov + 1)

        def fd(setter, x0):
            h = 1e-4 * max(1.0, abs(x0))
            sp, sm = copy.deepcopy(spec), copy.deepcopy(spec)
            s2p = setter(sp, x0 + h); s2m = setter(sm, x0 - h)
            return (self._elbo(sp, s2p, jitter)[0] - self._elbo(sm, s2m, jitter)[0]) / (2 * h)
        for d in range(D):
            def set_l(sp, v, d=d):
                sp["dims"][d]["lengthscale"] = v
                return s2
            g[d] = fd(set_l, float(spec["dims"][d]["lengthscale"]))
        for r in range(nov):
            def set_v(sp, v, r=r):
                sp["order_variances"][r] = v
                return s2
            g[2 * D + r] = fd(set_v, float(spec["order_variances"][r]))
        g[2 * D + nov] = fd(lambda sp, v: v, s2)
        return e, g

    def sgpr_alpha(self, M):
        L, LB, c = self._post
        return np.linalg.solve(L.T, np.linalg.solve(LB.T, c[:, self._sel]))

    def sgpr_predict(self, desc, Xs):
        L, LB, c = self._post
        Kus = o.oak_K(desc.spec, self.Z, Xs)
        t1 = sla.solve_triangular(L, Kus, lower=True)
        t2 = sla.solve_triangular(LB, t1, lower=True)
        return t2.T @ c[:, self._sel], o.oak_K_diag(desc.spec, Xs) + np.sum(t2 * t2, 0) - np.sum(t1 * t1, 0)

    def sgpr_last_terms(self):
        return {}

    # -- GPR (just what get_sobol of a full model touches) --------------------------------------------------------------
    def gpr_set_data(self, X, y):
        self.gX, self.gy = np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64).reshape(-1, 1)

    def gpr_set_targets(self, y):
        self.gy = np.asarray(y, dtype=np.float64).reshape(-1, 1)

    def gpr_log_marginal(self, desc, noise_var):
        self._gpost = (desc.spec, float(noise_var))
        return o.gpr_log_marginal_likelihood(desc.spec, self.gX, self.gy, noise_var)

    def gpr_alpha(self, n):
        spec, s2 = self._gpost
        return o.gpr_alpha(spec, self.gX, self.gy, s2)[:, 0]

    # -- Sobol / preprocessing ---------------------------------------------------------------------------------------
    def sobol(self, desc, Xc, alpha, subsets, use_order_var=True, delta=1.0, mu=0.0, collective=False):
        """``collective``: the real library shards the index-pair rows over the ranks and sums partial Gram matrices through the
        communicator; the double keeps the exchange (a share of every term per rank, summed by the host all-reduce) so that a
        rank that skips the call, or calls it with different arguments, still fails the job."""
        all_subsets, vals = o.compute_sobol_oak(desc.spec, np.asarray(Xc), np.asarray(alpha).reshape(-1, 1), delta, mu, use_order_var)
        look = {tuple(s): v for s, v in zip(all_subsets, vals)}
        out = np.array([look[tuple(s)] for s in subsets])
        if collective and self._world > 1:
            out = self._sum(out / self._world)
        return out

    def flow_forward(self, X, kind, params):
        X = np.asarray(X, dtype=np.float64).copy()
        for d, k in enumerate(kind):
            if k == 3:
                X[:, d] = (X[:, d] - params[d][0]) / params[d][1]
            elif k != 0:
                raise NotImplementedError("the fake context has no normalising flows")
        return X


_default = None


def install():
    """Route every context the host package creates to the fake (call in a fresh process)."""
    from oak import _capi
    if not hasattr(_capi, "_RealKernelDesc"):
        _capi._RealKernelDesc = _capi.KernelDesc
    _capi.KernelDesc = SpecDesc
    _capi.HipContext = FakeContext

    def default_context():
        global _default
        if _default is None:
            _default = FakeContext(0)
        return _default
    _capi.default_context = default_context

"""Multi-core CPU restatement of the reference's SGPR ELBO: C/OpenMP Gram (gram_oracle.c) + BLAS/LAPACK
solve path in GPflow's op order (A = L^-1 Kuf, AAT = A A^T, oak/utils.py:187-195), chunked over N because the
reference's D live Kuf-sized matrices do not fit at the benchmark sizes.

TEST INFRASTRUCTURE ONLY: used by tests/ for larger parity cases and by bench.py's ``cpu_baseline`` leg.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.linalg as sla

from . import build as _build
from . import oak_oracle as o

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = _build.build()
        _lib = C.CDLL(str(path))
        _lib.oak_oracle_max_threads.restype = C.c_int
    return _lib


def max_threads() -> int:
    return int(_load().oak_oracle_max_threads())


def _featurize(spec, X):
    """Per-dim x/l (or category index) and cov_X_s vectors, dimension-major, via the NumPy oracle formulas."""
    X = np.asarray(X, dtype=np.float64)
    D = len(spec["dims"])
    xs = np.zeros((D, X.shape[0]))
    cs = np.zeros((D, X.shape[0]))
    for d, dim in enumerate(spec["dims"]):
        col = X[:, o.active_col(spec, d)]
        if dim["type"] == "rbf":
            xs[d] = col / dim["lengthscale"]
            if dim["measure"] is not None:
                cs[d] = o.cov_X_s(col.reshape(-1, 1), dim)[:, 0]
        else:
            xs[d] = col.astype(np.int32)
    return np.ascontiguousarray(xs), np.ascontiguousarray(cs)


def _tables(spec):
    D = len(spec["dims"])
    typ = np.zeros(D, np.int32); var = np.ones(D); inv_v = np.zeros(D)
    ncat = np.zeros(D, np.int32); off = np.zeros(D, np.int32)
    tabs = []
    pos = 0
    for d, dim in enumerate(spec["dims"]):
        var[d] = dim["variance"]
        if dim["type"] == "rbf":
            typ[d] = 0
            inv_v[d] = 0.0 if dim["measure"] is None else 1.0 / o.var_s(dim)
        else:
            typ[d] = 1
            B = o.binary_table(dim) if dim["type"] == "binary" else o.categorical_table(dim)
            Bd = o.binary_diag_table(dim) if dim["type"] == "binary" else o.categorical_diag_table(dim)
            ncat[d] = B.shape[0]; off[d] = pos
            tabs += [B.reshape(-1), Bd.reshape(-1)]
            pos += B.size + Bd.size
    tables = np.ascontiguousarray(np.concatenate(tabs)) if tabs else np.zeros(1)
    R = spec["max_interaction_depth"]
    v = spec["order_variances"]
    w = np.array(v if spec.get("share_var_across_orders", True) else [v[0]] + [1.0] * R, dtype=np.float64)
    return typ, var, inv_v, ncat, off, tables, w


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def gram(spec, X, X2=None, threads: int = 0) -> np.ndarray:
    lib = _load()
    typ, var, inv_v, ncat, off, tables, w = _tables(spec)
    x1, c1 = _featurize(spec, X)
    x2, c2 = (x1, c1) if X2 is None else _featurize(spec, X2)
    n1, n2 = x1.shape[1], x2.shape[1]
    out = np.empty((n1, n2))
    D, R = len(spec["dims"]), spec["max_interaction_depth"]
    lib.oak_oracle_gram(C.c_int(D), C.c_int(R), _p(typ, C.c_int), _p(var, C.c_double), _p(inv_v, C.c_double),
                        _p(ncat, C.c_int), _p(off, C.c_int), _p(tables, C.c_double), _p(w, C.c_double),
                        _p(x1, C.c_double), _p(c1, C.c_double), C.c_int64(n1), C.c_int64(n1),
                        _p(x2, C.c_double), _p(c2, C.c_double), C.c_int64(n2), C.c_int64(n2),
                        _p(out, C.c_double), C.c_int(threads))
    return out


def gram_diag(spec, X) -> np.ndarray:
    lib = _load()
    typ, var, inv_v, ncat, off, tables, w = _tables(spec)
    x1, c1 = _featurize(spec, X)
    n1 = x1.shape[1]
    out = np.empty(n1)
    D, R = len(spec["dims"]), spec["max_interaction_depth"]
    lib.oak_oracle_gram_diag(C.c_int(D), C.c_int(R), _p(typ, C.c_int), _p(var, C.c_double), _p(inv_v, C.c_double),
                             _p(ncat, C.c_int), _p(off, C.c_int), _p(tables, C.c_double), _p(w, C.c_double),
                             _p(x1, C.c_double), _p(c1, C.c_double), C.c_int64(n1), C.c_int64(n1), _p(out, C.c_double))
    return out


def sgpr_elbo_chunked(spec, X, Y, Z, noise_variance, jitter=o.JITTER, chunk=8192, threads: int = 0, return_parts=False):
    """gpflow SGPR.elbo in GPflow's op order, summed over row chunks of X (all N-dependence is a sum)."""
    X = np.asarray(X, dtype=np.float64); Y = np.asarray(Y, dtype=np.float64).reshape(-1, 1)
    Z = np.asarray(Z, dtype=np.float64)
    N, M = X.shape[0], Z.shape[0]
    sigma = np.sqrt(noise_variance)
    kuu = gram(spec, Z, None, threads) + jitter * np.eye(M)
    L = np.linalg.cholesky(kuu)
    AAT = np.zeros((M, M)); Aerr = np.zeros((M, 1)); kdiag_sum = 0.0
    for a0 in range(0, N, chunk):
        Xc, Yc = X[a0:a0 + chunk], Y[a0:a0 + chunk]
        kuf = gram(spec, Z, Xc, threads)                               # [M, nc]  (Kuf = kernel(Z, X))
        A = sla.solve_triangular(L, kuf, lower=True, check_finite=False) / sigma
        AAT += A @ A.T
        Aerr += A @ Yc
        kdiag_sum += gram_diag(spec, Xc).sum()
    B = AAT + np.eye(M)
    LB = np.linalg.cholesky(B)
    c = sla.solve_triangular(LB, Aerr, lower=True, check_finite=False) / sigma
    bound = -0.5 * N * np.log(2 * np.pi)
    bound += -np.sum(np.log(np.diag(LB)))
    bound -= 0.5 * N * np.log(noise_variance)
    bound += -0.5 * np.sum(np.square(Y)) / noise_variance
    bound += 0.5 * np.sum(np.square(c))
    bound += -0.5 * kdiag_sum / noise_variance
    bound += 0.5 * np.trace(AAT)
    if return_parts:
        # the kernel-dependent pieces of the bound one by one (same names as HipContext.sgpr_last_terms)
        terms = dict(sum_log_diag_LB=float(np.sum(np.log(np.diag(LB)))), cTc=float(np.sum(np.square(c))),
                     tr_AAT=float(np.trace(AAT)), kappa=float(kdiag_sum), yy=float(np.sum(np.square(Y))), n_rows=float(N),
                     logdet_Kuu=float(2.0 * np.sum(np.log(np.diag(L)))))
        return float(bound), dict(L=L, LB=LB, c=c, AAT=AAT, terms=terms)
    return float(bound)

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


def effective_cpus() -> int:
    """CPUs this process can actually use: the scheduler affinity mask, capped by the cgroup CPU quota (a container may see
    256 logical CPUs in os.cpu_count() and be entitled to a handful of them -- timing 256 threads on those is not a
    256-thread baseline)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:                                   # pragma: no cover
        n = os.cpu_count() or 1
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()              # cgroup v2: "<quota|max> <period>"
        if txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except OSError:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except OSError:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


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


def _blas_pointer(name):
    """Raw address of SciPy's Fortran-ABI BLAS routine ``name`` (scipy.linalg.cython_blas exports them as capsules)."""
    import scipy.linalg.cython_blas as cb
    cap = cb.__pyx_capi__[name]
    C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
    C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
    C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
    C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    return C.c_void_p(C.pythonapi.PyCapsule_GetPointer(cap, C.pythonapi.PyCapsule_GetName(cap)))


def sgpr_elbo_manycore(spec, X, Y, Z, noise_variance, jitter=o.JITTER, chunk=32768, threads=0, blas_threads=0, timing=None, return_parts=False):
    """The same bound with the row-dependent part in ONE C routine (gram_oracle.c::oak_oracle_sgpr_rows): Kuf chunks by all
    OpenMP threads with a vectorised pair loop, then A = L^-1 Kuf / sigma, AAT += A A^T, Aerr += A y by single-threaded BLAS
    calls on 512-row blocks, ``blas_threads`` blocks at a time (default: min(OpenMP threads, 64) -- OpenBLAS serves at most
    2 x its build-time thread count concurrent callers).  This is bench.py's many-core CPU baseline."""
    import time as _time
    from threadpoolctl import threadpool_limits
    lib = _load()
    X = np.asarray(X, dtype=np.float64); Y = np.ascontiguousarray(np.asarray(Y, dtype=np.float64).reshape(-1))
    Z = np.asarray(Z, dtype=np.float64)
    N, M = X.shape[0], Z.shape[0]
    sigma = float(np.sqrt(noise_variance))
    t_all = _time.perf_counter()
    typ, var, inv_v, ncat, off, tables, w = _tables(spec)
    kuu = gram(spec, Z, None) + jitter * np.eye(M)
    L = np.linalg.cholesky(kuu)
    Lf = np.asfortranarray(L)
    x1, c1 = _featurize(spec, X)
    x2, c2 = _featurize(spec, Z)
    AAT = np.zeros((M, M), order="F"); Aerr = np.zeros(M); kd = np.zeros(1); secs = np.zeros(3)
    nthr = threads if threads > 0 else min(max_threads(), effective_cpus())
    bt = blas_threads if blas_threads > 0 else min(nthr, 64)
    D, R = len(spec["dims"]), spec["max_interaction_depth"]
    lib.oak_oracle_sgpr_rows.restype = C.c_int
    with threadpool_limits(limits=1, user_api="blas"):
        rc = lib.oak_oracle_sgpr_rows(C.c_int(D), C.c_int(R), _p(typ, C.c_int), _p(var, C.c_double), _p(inv_v, C.c_double),
                                      _p(ncat, C.c_int), _p(off, C.c_int), _p(tables, C.c_double), _p(w, C.c_double),
                                      _p(x1, C.c_double), _p(c1, C.c_double), C.c_int64(N), C.c_int64(N),
                                      _p(x2, C.c_double), _p(c2, C.c_double), C.c_int64(M), C.c_int64(M),
                                      _p(Y, C.c_double), _p(Lf, C.c_double), C.c_double(1.0 / sigma), C.c_int64(chunk), C.c_int(bt),
                                      _blas_pointer("dtrsm"), _blas_pointer("dsyrk"), _blas_pointer("dgemv"),
                                      _p(AAT, C.c_double), _p(Aerr, C.c_double), _p(kd, C.c_double), _p(secs, C.c_double), C.c_int(nthr))
    if rc != 0:
        raise MemoryError("oak_oracle_sgpr_rows: out of memory")
    AAT = np.tril(AAT) + np.tril(AAT, -1).T
    B = AAT + np.eye(M)
    LB = np.linalg.cholesky(B)
    c = sla.solve_triangular(LB, Aerr.reshape(-1, 1), lower=True, check_finite=False) / sigma
    bound = -0.5 * N * np.log(2 * np.pi)
    bound += -np.sum(np.log(np.diag(LB)))
    bound -= 0.5 * N * np.log(noise_variance)
    bound += -0.5 * np.sum(np.square(Y)) / noise_variance
    bound += 0.5 * np.sum(np.square(c))
    bound += -0.5 * float(kd[0]) / noise_variance
    bound += 0.5 * np.trace(AAT)
    if timing is not None:
        total = _time.perf_counter() - t_all
        timing.update(gram=float(secs[0]), trsm_syrk_gemv=float(secs[1]), reduce=float(secs[2]),
                      rest=total - float(secs.sum()), omp_threads=nthr, blas_callers=bt)
    if return_parts:
        terms = dict(sum_log_diag_LB=float(np.sum(np.log(np.diag(LB)))), cTc=float(np.sum(np.square(c))),
                     tr_AAT=float(np.trace(AAT)), kappa=float(kd[0]), yy=float(np.sum(np.square(Y))), n_rows=float(N),
                     logdet_Kuu=float(2.0 * np.sum(np.log(np.diag(L)))))
        return float(bound), dict(L=L, LB=LB, c=c, AAT=AAT, terms=terms)
    return float(bound)


def blas_info():
    """What the BLAS / OpenMP pools of this process are (bench.py prints it next to the CPU baseline)."""
    try:
        from threadpoolctl import threadpool_info
        return [{k: d.get(k) for k in ("user_api", "internal_api", "num_threads", "version", "threading_layer")} for d in threadpool_info()]
    except Exception as ex:                                            # pragma: no cover
        return [{"error": repr(ex)}]


def sgpr_elbo_chunked(spec, X, Y, Z, noise_variance, jitter=o.JITTER, chunk=8192, threads: int = 0, return_parts=False, timing=None):
    """gpflow SGPR.elbo in GPflow's op order, summed over row chunks of X (all N-dependence is a sum).

    Per chunk: Kuf by the C/OpenMP Gram, A = L^-1 Kuf / sigma by BLAS dtrsm in place, AAT += A A^T by dsyrk, Aerr += A y
    (oak/utils.py:187-195).  The Gram is produced as K(X_c, Z), rows = data points, C-contiguous: that IS Kuf = K(Z, X_c) in
    column-major order, so the BLAS calls work on it without a copy or a transpose.  ``timing`` (a dict) receives the seconds
    spent in gram / trsm / syrk / rest."""
    import time as _time
    from scipy.linalg import blas as _blas
    X = np.asarray(X, dtype=np.float64); Y = np.asarray(Y, dtype=np.float64).reshape(-1, 1)
    Z = np.asarray(Z, dtype=np.float64)
    N, M = X.shape[0], Z.shape[0]
    sigma = np.sqrt(noise_variance)
    tm = dict(gram=0.0, trsm=0.0, syrk=0.0, rest=0.0)
    t_all = _time.perf_counter()
    kuu = gram(spec, Z, None, threads) + jitter * np.eye(M)
    L = np.linalg.cholesky(kuu)
    Lf = np.asfortranarray(L)
    AAT = np.zeros((M, M), order="F"); Aerr = np.zeros((M, 1)); kdiag_sum = 0.0
    for a0 in range(0, N, chunk):
        Xc, Yc = X[a0:a0 + chunk], Y[a0:a0 + chunk]
        t0 = _time.perf_counter()
        kfu = gram(spec, Xc, Z, threads)                               # [nc, M] C-order == Kuf [M, nc] column-major
        t1 = _time.perf_counter()
        A = _blas.dtrsm(1.0 / sigma, Lf, kfu.T, side=0, lower=1, trans_a=0, diag=0, overwrite_b=1)   # A = L^-1 Kuf / sigma
        t2 = _time.perf_counter()
        AAT = _blas.dsyrk(1.0, A, beta=1.0, c=AAT, trans=0, lower=1, overwrite_c=1)                 # lower triangle of AAT += A A^T
        t3 = _time.perf_counter()
        Aerr += A @ Yc
        kdiag_sum += gram_diag(spec, Xc).sum()
        tm["gram"] += t1 - t0; tm["trsm"] += t2 - t1; tm["syrk"] += t3 - t2
    AAT = np.tril(AAT) + np.tril(AAT, -1).T
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
    tm["rest"] = (_time.perf_counter() - t_all) - tm["gram"] - tm["trsm"] - tm["syrk"]
    if timing is not None:
        timing.update(tm)
    if return_parts:
        # the kernel-dependent pieces of the bound one by one (same names as HipContext.sgpr_last_terms)
        terms = dict(sum_log_diag_LB=float(np.sum(np.log(np.diag(LB)))), cTc=float(np.sum(np.square(c))),
                     tr_AAT=float(np.trace(AAT)), kappa=float(kdiag_sum), yy=float(np.sum(np.square(Y))), n_rows=float(N),
                     logdet_Kuu=float(2.0 * np.sum(np.log(np.diag(L)))))
        return float(bound), dict(L=L, LB=LB, c=c, AAT=AAT, terms=terms)
    return float(bound)

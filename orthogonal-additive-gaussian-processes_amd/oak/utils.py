"""Sobol indices, sufficient statistics and per-component predictions: host mirror of oak/utils.py:116-574.

``compute_sobol_oak`` and ``get_prediction_component`` run on the device (``oak_sobol`` / ``oak_component_predict``):
per-dimension L_d matrices are generated once and shared by every subset instead of being rebuilt per term.
The closed forms f1..f4 are kept as NumPy helpers because the reference's tests call them directly.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from . import _capi
from . import gpflow_lite as gpflow
from .gpflow_lite import TensorLike
from .input_measures import EmpiricalMeasure, MOGMeasure
from .oak_kernel import KernelComponenent, OAKKernel, get_list_representation, kernel_to_spec
from .ortho_binary_kernel import OrthogonalBinary
from .ortho_categorical_kernel import OrthogonalCategorical
from .ortho_rbf_kernel import OrthogonalRBFKernel


# ---- closed-form Gaussian-measure integrals, eq. (44)-(47) of the paper (oak/utils.py:116-165) -------------
def f1(x, y, sigma, lengthscales, delta, mu):
    l2, d2 = lengthscales ** 2, delta ** 2
    return (sigma ** 4 * lengthscales / np.sqrt(l2 + 2 * d2) * np.exp(-((x - y) ** 2) / (4 * l2))
            * np.exp(-((mu - (x + y) / 2) ** 2) / (2 * d2 + l2)))


def f2(x, y, sigma, lengthscales, delta, mu):
    l2, d2 = lengthscales ** 2, delta ** 2
    M = 1 / l2 + 1 / (l2 + d2)
    m = (mu / (l2 + d2) + x / l2) / M
    C = x ** 2 / l2 + mu ** 2 / (l2 + d2) - m ** 2 * M
    return (sigma ** 4 * lengthscales * np.sqrt((l2 + 2 * d2) / (d2 * M + 1)) * np.exp(-C / 2) / (l2 + d2)
            * np.exp(-((y - mu) ** 2) / (2 * (l2 + d2))) * np.exp(-((m - mu) ** 2) / (2 * (1 / M + d2))))


def f3(x, y, sigma, lengthscales, delta, mu):
    return f2(y, x, sigma, lengthscales, delta, mu)


def f4(x, y, sigma, lengthscales, delta, mu):
    l2, d2 = lengthscales ** 2, delta ** 2
    return (sigma ** 4 * l2 * (l2 + 2 * d2) * np.sqrt((l2 + d2) / (l2 + 3 * d2)) / ((l2 + d2) ** 2)
            * np.exp(-((x - mu) ** 2 + (y - mu) ** 2) / (2 * (l2 + d2))))


# ---- per-dimension L matrices (oak/utils.py:221-335), evaluated on the device --------------------------------
def _one_dim_desc(dim_spec):
    return _capi.KernelDesc(dict(dims=[dim_spec], order_variances=[0.0, 1.0], max_interaction_depth=1,
                                 share_var_across_orders=True))


def compute_L(X, lengthscale: float, variance: float, dim: int, delta: float, mu: float) -> np.ndarray:
    X = np.asarray(X, dtype=np.float64)
    d = dict(type="rbf", lengthscale=float(lengthscale), variance=1.0, measure=("gaussian", 0.0, 1.0), active_dim=int(dim))
    return _capi.default_context().sobol_L(_one_dim_desc(d), 0, float(variance), float(delta), float(mu), X)


def compute_L_binary_kernel(X, p0: float, variance: float, dim: int) -> np.ndarray:
    assert 0 <= p0 <= 1
    X = np.asarray(X, dtype=np.float64)
    d = dict(type="binary", p0=float(p0), variance=1.0, active_dim=int(dim))
    return _capi.default_context().sobol_L(_one_dim_desc(d), 0, float(variance), 1.0, 0.0, X)


def compute_L_categorical_kernel(X, W, kappa, p, variance: float, dim: int) -> np.ndarray:
    p = np.asarray(p, dtype=np.float64).reshape(-1, 1)
    assert np.abs(p.sum() - 1) < 1e-6
    X = np.asarray(X, dtype=np.float64)
    d = dict(type="categorical", p=p, W=np.asarray(W, dtype=np.float64), kappa=np.asarray(kappa, dtype=np.float64),
             variance=1.0, active_dim=int(dim))
    return _capi.default_context().sobol_L(_one_dim_desc(d), 0, float(variance), 1.0, 0.0, X)


def compute_L_empirical_measure(x, w, kernel: OrthogonalRBFKernel, z) -> np.ndarray:
    """L = Kxu^T diag(w) Kxu with Kxu = kernel.K(x, z) (oak/utils.py:312-335)."""
    z = np.asarray(z, dtype=np.float64).reshape(-1, 1)
    d = kernel.dim_spec(0)
    d["measure"] = ("empirical", np.asarray(x, dtype=np.float64).reshape(-1, 1), np.asarray(w, dtype=np.float64).reshape(-1, 1))
    return _capi.default_context().sobol_L(_one_dim_desc(d), 0, 1.0, 1.0, 0.0, z)


# ---- sufficient statistics (oak/utils.py:168-218) ------------------------------------------------------------
def get_model_sufficient_statistics(m, get_L=True):
    """alpha such that the predictive mean is K(x*, Xc) alpha and, with ``get_L`` (the reference's default), the matrix L
    of oak/utils.py:168-218: the "effective" factor inv(L^-1 - LB^-1 L^-1) of a sparse model, chol(K + noise I) of a full
    one."""
    if not isinstance(m, (gpflow.SGPR, gpflow.GPR, gpflow.SVGP)):
        raise NotImplementedError
    alpha = m.alpha()
    if not get_L:
        return alpha
    return alpha, TensorLike(m.effective_L())


def _sobol_inputs(model):
    assert isinstance(model.kernel, OAKKernel), "only work for OAK kernel"
    num_dims = model.data[0].shape[1]
    selected, kernel_list = get_list_representation(model.kernel, num_dims=num_dims)
    return num_dims, selected[1:], kernel_list


def _first_output(alpha) -> np.ndarray:
    """alpha as a vector.  With several output columns the reference's expressions pick output 0
    (``...numpy()[0][0]`` at oak/utils.py:428-430, ``predictive_component_mean[:, 0]`` at :529) -- so does this."""
    a = np.asarray(alpha, dtype=np.float64)
    return np.ascontiguousarray(a.reshape(len(a), -1)[:, 0])


def compute_sobol_oak(model, delta: float, mu: float, share_var_across_orders: Optional[bool] = True
                      ) -> Tuple[List[List[int]], List[float]]:
    """Sobol index alpha^T (prod_{d in S} L_d) alpha of every non-constant term (oak/utils.py:338-435)."""
    num_dims, subsets, _ = _sobol_inputs(model)
    for k in model.kernel.kernels:   # same support matrix as the reference (:386-427)
        if isinstance(k, OrthogonalRBFKernel):
            if isinstance(k.measure, MOGMeasure):
                raise NotImplementedError
        elif not isinstance(k, (OrthogonalBinary, OrthogonalCategorical)):
            raise NotImplementedError
    Xc = model.inducing_variable.Z.numpy() if isinstance(model, (gpflow.SGPR, gpflow.SVGP)) else model.data[0]
    alpha = get_model_sufficient_statistics(model, get_L=False)
    desc = _capi.KernelDesc(kernel_to_spec(model.kernel))
    comm = getattr(model, "_comm", None)
    hip = getattr(model, "_hip", None)
    kw = dict(use_order_var=bool(share_var_across_orders), delta=delta, mu=mu)
    if comm is not None and hip is not None and getattr(hip, "_oak_comm_attached", None) is comm and len(subsets) >= 8 * comm.world:
        # the model's context has joined the job's communicator (row-sharded SGPR / SVGP): one collective call, the index-pair
        # rows of the Gram of products (or blocks of terms) sharded over the ranks and summed on the device.  A model whose
        # context never joined (a full GPR under a multi-rank job, an SVGP that has not seen its data yet) evaluates every
        # term on its own: replicated, identical on every rank, no exchange.
        sobol = hip.sobol(desc, Xc, _first_output(alpha), subsets, collective=True, **kw)
    else:
        sobol = _capi.default_context().sobol(desc, Xc, _first_output(alpha), subsets, **kw)
    assert len(subsets) == len(sobol)
    return subsets, [float(s) for s in sobol]


def get_prediction_component(m, alpha, X: np.ndarray = None, share_var_across_orders: Optional[bool] = True) -> list:
    """Predictive mean of every additive term (oak/utils.py:491-530)."""
    if X is None:
        X = m.data[0]
    X = np.asarray(X, dtype=np.float64)
    subsets = get_list_representation(m.kernel, num_dims=X.shape[1])[0][1:]
    Xc = m.data[0] if isinstance(m, gpflow.GPR) else m.inducing_variable.Z.numpy()
    desc = _capi.KernelDesc(kernel_to_spec(m.kernel))
    out = _capi.default_context().component_predict(desc, X, Xc, _first_output(alpha), subsets,
                                                    use_order_var=bool(share_var_across_orders))
    return [TensorLike(row) for row in out]


# ---- inducing-point initialisation (oak/utils.py:533-574) -------------------------------------------------------------
KMEANS_SEED_SAMPLE = 1 << 24     # rows the device k-means++ accepts; above it the seeds come from a fixed random subsample


def kmeans_centres(X, n_clusters: int, random_state: int = 0, max_iter: int = 300, tol: float = 1e-4) -> np.ndarray:
    """``KMeans(n_clusters, random_state=random_state).fit(X).cluster_centers_`` on the device.

    Same pipeline as scikit-learn's ``KMeans.fit`` (one k-means++ initialisation, the default since scikit-learn 1.4):
    centre the data, draw the greedy k-means++ seeds from ``RandomState(random_state)`` (``oak_kmeans_plusplus``; the
    random numbers are drawn on the host in scikit-learn's order), run Lloyd (``oak_kmeans``) with the absolute tolerance
    ``tol * mean(var(X, axis=0))``, add the mean back.  Reproduces the reference call to 1e-9
    (tests/test_gpu_kmeans.py); above ``KMEANS_SEED_SAMPLE`` rows the seeding sees a fixed random subsample."""
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    if X.ndim != 2:
        raise ValueError("X must be two-dimensional")
    mean = X.mean(axis=0)
    Xc = X - mean
    rs = np.random.RandomState(random_state)
    pool = Xc
    if Xc.shape[0] > KMEANS_SEED_SAMPLE:
        pool = Xc[np.random.RandomState(random_state).choice(Xc.shape[0], KMEANS_SEED_SAMPLE, replace=False)]
    ctx = _capi.default_context()
    seeds, _ = ctx.kmeans_plusplus(pool, n_clusters, random_state=rs)
    abs_tol = float(np.mean(np.var(X, axis=0)) * tol)
    centres, _, _, _ = ctx.kmeans(Xc, seeds, max_iter, abs_tol)
    return centres + mean


def initialize_kmeans_with_binary(X, binary_index: list, continuous_index: Optional[list] = None, n_clusters: Optional[int] = 200):
    from sklearn.cluster import KMeans
    X = np.asarray(X)
    Z = np.zeros([n_clusters, X.shape[1]])
    for index in binary_index:
        km = KMeans(n_clusters=n_clusters, random_state=0).fit(X[:, index][:, None])
        Z[:, index] = km.cluster_centers_.astype(int)[:, 0]
    if continuous_index is not None:
        Z[:, continuous_index] = kmeans_centres(X[:, continuous_index], n_clusters, random_state=0)
    return Z


def initialize_kmeans_with_categorical(X, binary_index: list, categorical_index: list, continuous_index: list,
                                       n_clusters: Optional[int] = 200):
    from sklearn.cluster import KMeans
    X = np.asarray(X)
    Z = np.zeros([n_clusters, X.shape[1]])
    for index in binary_index + categorical_index:
        km = KMeans(n_clusters=n_clusters, random_state=0).fit(X[:, index][:, None])
        Z[:, index] = km.cluster_centers_.astype(int)[:, 0]
    Z[:, continuous_index] = kmeans_centres(X[:, continuous_index], n_clusters, random_state=0)
    return Z

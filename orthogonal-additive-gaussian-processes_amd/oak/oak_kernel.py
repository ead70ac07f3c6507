"""Composite orthogonal additive kernel: host mirror of oak/oak_kernel.py:24-364.

The class keeps the reference's constructor, attributes (``kernels``, ``variances``) and methods
(``K``, ``K_diag``, ``compute_additive_terms``) so model/utility code written against the reference keeps
working; evaluation is delegated to the fused HIP Gram kernel, which never materialises the D per-dimension
matrices nor the Newton-Girard intermediates.
"""
from __future__ import annotations

import itertools
from typing import List, Optional, Tuple, Type, Sequence

import numpy as np

from . import _capi
from . import gpflow_lite as gpflow
from .gpflow_lite import Parameter, TensorLike, _as_value
from .input_measures import EmpiricalMeasure, GaussianMeasure, MOGMeasure
from .ortho_binary_kernel import OrthogonalBinary
from .ortho_categorical_kernel import OrthogonalCategorical
from .ortho_rbf_kernel import OrthogonalRBFKernel


def bounded_param(low: float, high: float, param: float) -> Parameter:
    """Parameter constrained to (low, high) by a scaled sigmoid (oak/oak_kernel.py:24-33)."""
    return Parameter(param, transform=gpflow.Sigmoid(low, high))


def _columns(k) -> List[int]:
    """Columns of X a sub-kernel reads.  One for every constrained kernel (they are one-dimensional); an unconstrained RBF may
    read a group (OAKKernel(active_dims=[[0, 1], ...]), oak_kernel.py:74-82,199-210)."""
    dims = k.active_dims
    if isinstance(dims, slice):
        return [0]
    return [int(c) for c in np.asarray(dims).reshape(-1)]


def _first_col(k) -> int:
    cols = _columns(k)
    if len(cols) != 1:
        raise NotImplementedError(f"sub-kernel with {len(cols)} active columns {cols}: only an unconstrained RBF reads several")
    return cols[0]


def _has_trainable_base_variance(k) -> bool:
    base = getattr(k, "base_kernel", k)
    return isinstance(getattr(base, "variance", None), Parameter)


def _sub_kernel_spec(k, col: int) -> dict:
    if isinstance(k, (OrthogonalRBFKernel, OrthogonalBinary, OrthogonalCategorical)):
        return k.dim_spec(col)
    if isinstance(k, gpflow.RBF):   # unconstrained additive model, oak_kernel.py:199-210
        ls = np.asarray(_as_value(k.lengthscales)).reshape(-1)
        if ls.size != 1:
            raise NotImplementedError("ARD lengthscales inside one sub-kernel are not on the OAK path (the reference builds "
                                      "base_kernels[d](active_dims=...) with the default scalar lengthscale)")
        out = dict(type="rbf", lengthscale=float(ls[0]), variance=float(np.asarray(_as_value(k.variance)).reshape(-1)[0]),
                   measure=None, active_dim=col)
        if len(_columns(k)) > 1:
            out["active_dims"] = _columns(k)        # one RBF over the group's columns
        return out
    raise NotImplementedError(f"no HIP description for sub-kernel {type(k).__name__}")


def kernel_to_spec(kernel) -> dict:
    """Plain-data description (see _capi.KernelDesc) of an OAKKernel or of a single constrained sub-kernel."""
    if isinstance(kernel, OAKKernel):
        dims = [_sub_kernel_spec(k, _columns(k)[0] if isinstance(k, gpflow.RBF) else _first_col(k)) for k in kernel.kernels]
        # OAKKernel pins the base variance to a constant 1 only for Gaussian-measure / binary / categorical dims under
        # share_var_across_orders (oak_kernel.py:163-166,179,187); empirical- and MOG-measure dims keep a trainable
        # base_kernel.variance, whose gradient needs the pair-kernel contribution (grad_base_var)
        return dict(dims=dims, order_variances=[float(np.asarray(_as_value(v)).reshape(-1)[0]) for v in kernel.variances],
                    max_interaction_depth=int(kernel.max_interaction_depth),
                    share_var_across_orders=bool(kernel.share_var_across_orders),
                    base_var_grad=(not kernel.share_var_across_orders)
                    or any(_has_trainable_base_variance(k) for k in kernel.kernels))
    if isinstance(kernel, (OrthogonalRBFKernel, OrthogonalBinary, OrthogonalCategorical, gpflow.RBF)):
        if isinstance(kernel, gpflow.RBF) and not isinstance(kernel.active_dims, slice) and len(kernel.active_dims) != 1:
            raise NotImplementedError("multi-column stand-alone RBF inside a model is not on the OAK path")
        return dict(dims=[_sub_kernel_spec(kernel, _first_col(kernel))], order_variances=[0.0, 1.0],
                    max_interaction_depth=1, share_var_across_orders=True, base_var_grad=True)
    raise NotImplementedError(f"kernel {type(kernel).__name__} is not supported by the HIP path")


class OAKKernel(gpflow.Kernel):
    """Orthogonal additive kernel (oak/oak_kernel.py:36-278); arguments as in the reference (:59-73)."""

    def __init__(
        self,
        base_kernels: List[Type[gpflow.Kernel]],
        num_dims: int,
        max_interaction_depth: int,
        active_dims: Optional[List[List[int]]] = None,
        constrain_orthogonal: bool = False,
        p0: Optional[List[float]] = None,
        p: Optional[List[float]] = None,
        lengthscale_bounds: Optional[List[float]] = None,
        empirical_locations: Optional[List[float]] = None,
        empirical_weights: Optional[List[float]] = None,
        gmm_measures: Optional[List[MOGMeasure]] = None,
        share_var_across_orders: Optional[bool] = True,
    ):
        super().__init__(active_dims=range(num_dims))
        if active_dims is None:
            active_dims = [[d] for d in range(num_dims)]
        flat = [d for group in active_dims for d in group]
        assert max(flat) <= num_dims, "Active dims exceeding num dims."                    # :79 (sic: <=)
        assert len(flat) == len(np.unique(flat)), "Active dims contains duplicates."       # :80-82
        n_sub = len(active_dims)
        self.base_kernels, self.max_interaction_depth = base_kernels, max_interaction_depth
        self.share_var_across_orders = share_var_across_orders
        p0 = [None] * n_sub if p0 is None else p0
        p = [None] * n_sub if p is None else p
        one = np.ones(1)   # constant (non-trainable) unit variance, as tf.ones(1) in :164-166,179,187,209
        self.kernels = []
        if constrain_orthogonal:
            if empirical_locations is None:
                assert empirical_weights is None, "Cannot have weights without locations"
                empirical_locations, empirical_weights = [None] * n_sub, [None] * n_sub
            elif empirical_weights is not None:
                n_loc = [None if empirical_locations[d] is None else len(empirical_locations[d]) for d in range(n_sub)]
                n_w = [None if empirical_weights[d] is None else len(empirical_weights[d]) for d in range(n_sub)]
                assert n_loc == n_w, f"Shape of empirical measure locations {n_loc} do not match weights {n_w}"
            gmm_measures = [None] * n_sub if gmm_measures is None else gmm_measures
            delta2 = 1   # variance of the Gaussian input measure, hard-coded in the reference (:84)
            for d in range(n_sub):
                if empirical_locations[d] is not None and gmm_measures[d] is not None:
                    raise ValueError(f"Both empirical and GMM measure defined for input {d}")      # :132-138
                if p0[d] is None and p[d] is None:
                    if empirical_locations[d] is not None:
                        meas = EmpiricalMeasure(empirical_locations[d], empirical_weights[d])
                        k = OrthogonalRBFKernel(base_kernels[d](), meas, active_dims=active_dims[d])
                    elif gmm_measures[d] is not None:
                        k = OrthogonalRBFKernel(base_kernels[d](), measure=gmm_measures[d], active_dims=active_dims[d])
                    else:
                        k = OrthogonalRBFKernel(base_kernels[d](), GaussianMeasure(0, delta2), active_dims=active_dims[d])
                        if share_var_across_orders:
                            k.base_kernel.variance = one
                    if lengthscale_bounds is not None:
                        k.base_kernel.lengthscales = bounded_param(lengthscale_bounds[0], lengthscale_bounds[1], 1)
                elif p[d] is not None:
                    assert base_kernels[d] is None
                    k = OrthogonalCategorical(p=p[d], active_dims=active_dims[d])
                    if share_var_across_orders:
                        k.variance = one
                else:
                    assert base_kernels[d] is None
                    k = OrthogonalBinary(p0=p0[d], active_dims=active_dims[d])
                    if share_var_across_orders:
                        k.variance = one
                self.kernels.append(k)
        else:   # unconstrained kernel with the additive structure (:191-210)
            assert empirical_locations is None, "Cannot have empirical locations without orthogonal constraint"
            assert empirical_weights is None, "Cannot have empirical weights without orthogonal constraint"
            for d in range(n_sub):
                if p0[d] is None:
                    k = base_kernels[d](active_dims=active_dims[d])
                else:
                    assert base_kernels[d] is None
                    k = OrthogonalBinary(p0=p0[d], active_dims=active_dims[d])
                if share_var_across_orders:
                    k.variance = one
                self.kernels.append(k)
        n_var = max_interaction_depth + 1 if share_var_across_orders else 1       # :212-221
        self.variances = [Parameter(1.0, transform=gpflow.positive()) for _ in range(n_var)]

    # -- evaluation ------------------------------------------------------------------------------------
    def _desc(self) -> _capi.KernelDesc:
        return _capi.KernelDesc(kernel_to_spec(self))

    def compute_additive_terms(self, kernel_matrices):
        """[e_0 .. e_R] of the given same-shaped arrays (oak/oak_kernel.py:223-249), on the device."""
        mats = [np.asarray(m, dtype=np.float64) for m in kernel_matrices]
        shape = mats[0].shape
        stacked = np.ascontiguousarray(np.stack([m.reshape(-1) for m in mats]))
        out = _capi.default_context().additive_terms(stacked, int(self.max_interaction_depth))
        return [TensorLike(o.reshape(shape)) for o in out]

    def K(self, X, X2=None):
        return TensorLike(_capi.default_context().gram(self._desc(), X, X2))          # :251-265

    def K_diag(self, X):
        return TensorLike(_capi.default_context().gram_diag(self._desc(), X))          # :267-278


class KernelComponenent(gpflow.Kernel):
    """One additive term sigma2_|S| * prod_{d in S} k_d (oak/oak_kernel.py:281-335; the reference's spelling)."""

    def __init__(self, oak_kernel: OAKKernel, iComponent_list: List[int], share_var_across_orders: Optional[bool] = True):
        super().__init__(active_dims=oak_kernel.active_dims)
        self.oak_kernel = oak_kernel
        self.iComponent_list = iComponent_list
        self.share_var_across_orders = share_var_across_orders
        self.kernels = [k for i, k in enumerate(oak_kernel.kernels) if i in iComponent_list]

    def _subset(self):
        return sorted(int(i) for i in self.iComponent_list)

    def K(self, X, X2=None):
        ctx = _capi.default_context()
        return TensorLike(ctx.gram_component(self.oak_kernel._desc(), self._subset(), bool(self.share_var_across_orders), X, X2))

    def K_diag(self, X):
        ctx = _capi.default_context()
        return TensorLike(ctx.gram_component_diag(self.oak_kernel._desc(), self._subset(), bool(self.share_var_across_orders), X))


class _ComponentList(list):
    """``kernel_list`` of get_list_representation: the KernelComponenent of term i, built when it is asked for.  A depth-4 kernel
    over 32 inputs has 41 449 terms; the Sobol pass wants their index subsets, not 41 449 Python objects (0.4 s to build, against
    15 ms for all the indices on the device).  It IS a list, as in the reference (``isinstance(x, list)``, ``kernel_list + [...]``,
    ``.append``): indexing builds and caches single components (``kernel_list[i] is kernel_list[i]``), anything that needs the whole
    list (iteration, concatenation, mutation, comparison) fills the underlying list first."""

    def __init__(self, kernel, subsets, share_var_across_orders):
        super().__init__()
        self._kernel, self._subsets, self._share0 = kernel, subsets, share_var_across_orders
        self._cache, self._full = {}, False

    def _component(self, i):
        c = self._cache.get(i)
        if c is None:
            # as in the reference (:362) only the constant component takes ``share_var_across_orders`` from the caller
            c = KernelComponenent(self._kernel, self._subsets[i], share_var_across_orders=self._share0) if i == 0 \
                else KernelComponenent(self._kernel, self._subsets[i])
            self._cache[i] = c
        return c

    def _fill(self):
        if not self._full:
            self._full = True
            list.extend(self, (self._component(i) for i in range(len(self._subsets))))
            self._cache = {}
        return self

    def __len__(self):
        return list.__len__(self) if self._full else len(self._subsets)

    def __getitem__(self, i):
        if self._full or isinstance(i, slice):
            return list.__getitem__(self._fill(), i)
        n = len(self._subsets)
        j = i + n if i < 0 else i
        if not 0 <= j < n:
            raise IndexError("list index out of range")
        return self._component(j)

    def __iter__(self):
        return list.__iter__(self._fill())

    def __reversed__(self):
        return list.__reversed__(self._fill())

    def __contains__(self, x):
        return list.__contains__(self._fill(), x)

    def __add__(self, other):
        return list(self._fill()) + list(other)

    def __radd__(self, other):
        return list(other) + list(self._fill())

    def __mul__(self, k):
        return list(self._fill()) * k

    __rmul__ = __mul__

    def __eq__(self, other):
        return list.__eq__(self._fill(), other)

    def __ne__(self, other):
        return list.__ne__(self._fill(), other)

    __hash__ = None

    def __repr__(self):
        return list.__repr__(self._fill())

    def __reduce__(self):
        return (list, (list(self._fill()),))


def _fill_first(name):
    def method(self, *a, **k):
        return getattr(list, name)(self._fill(), *a, **k)
    method.__name__ = name
    return method


for _n in ("append", "extend", "insert", "pop", "remove", "clear", "index", "count", "sort", "reverse", "copy",
           "__setitem__", "__delitem__", "__iadd__", "__imul__"):
    setattr(_ComponentList, _n, _fill_first(_n))
del _n


def get_list_representation(kernel: OAKKernel, num_dims: int, share_var_across_orders: Optional[bool] = True
                            ) -> Tuple[List[List[int]], List]:
    """All interaction subsets up to the kernel's depth, constant term first (oak/oak_kernel.py:338-364).
    As in the reference (:362) non-constant components ignore ``share_var_across_orders``."""
    assert isinstance(kernel, OAKKernel)
    selected_dims: List[List[int]] = [[]]
    for order in range(1, kernel.max_interaction_depth + 1):
        selected_dims.extend(list(combo) for combo in itertools.combinations(range(num_dims), order))
    return selected_dims, _ComponentList(kernel, selected_dims, share_var_across_orders)


# --------------------------------------------------------------------------------------------------------
# gradient scatter: packed d objective / d (constrained parameter) from the HIP backward pass -> Parameter objects
# layout of gvec: [lengthscale (D) | base_var (D) | order_var (n_order_var) | noise_var | dTable (meas_data_len)]
# --------------------------------------------------------------------------------------------------------
def scatter_gradient(kernel, likelihood, desc: _capi.KernelDesc, gvec: np.ndarray, variables):
    D = desc.D
    g_ls, g_bv = gvec[:D], gvec[D:2 * D]
    n_ov = desc.order_var.size
    g_ov = gvec[2 * D:2 * D + n_ov]
    g_noise = gvec[2 * D + n_ov]
    g_tab = gvec[2 * D + n_ov + 1:]
    lookup = {id(likelihood.variance): g_noise} if hasattr(likelihood, "variance") else {}
    subs = kernel.kernels if isinstance(kernel, OAKKernel) else [kernel]
    if isinstance(kernel, OAKKernel):
        for r, v in enumerate(kernel.variances):
            lookup[id(v)] = g_ov[r]
    for d, k in enumerate(subs):
        base = getattr(k, "base_kernel", k)
        if hasattr(base, "lengthscales") and isinstance(base.lengthscales, Parameter):
            lookup[id(base.lengthscales)] = g_ls[d]
        var = getattr(base, "variance", None)
        if isinstance(var, Parameter):
            lookup[id(var)] = g_bv[d]
        if isinstance(k, OrthogonalCategorical):
            off, C = desc.cat_blocks[d]
            GB = g_tab[off:off + C * C].reshape(C, C)      # d obj / d B (unit-variance table)
            gW, gk = _categorical_chain(k.W.numpy(), k.kappa.numpy(), np.asarray(k.p, dtype=np.float64).reshape(-1, 1), GB)
            lookup[id(k.W)] = gW
            lookup[id(k.kappa)] = gk
    return [lookup.get(id(p)) for p in variables]


def _categorical_chain(W, kappa, p, GB):
    """Back-propagate d/dB through B = A - (Ap)(Ap)^T / (p^T A p), A = W W^T + diag(kappa)."""
    A = W @ W.T + np.diag(kappa)
    u = A @ p
    s = float((p.T @ u)[0, 0])
    Gs = 0.5 * (GB + GB.T)
    # dB = dA - (dA p u^T + u p^T dA)/s + u u^T (p^T dA p)/s^2
    GA = GB - (GB @ u @ p.T + p @ (u.T @ GB)) / s + p @ p.T * float((u.T @ GB @ u)[0, 0]) / (s * s)
    del Gs
    gW = (GA + GA.T) @ W
    gk = np.diag(GA).copy()
    return gW, gk

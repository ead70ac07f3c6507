"""Constrained coregion kernel for {0..C-1} inputs (host mirror of oak/ortho_categorical_kernel.py:14-74)."""
from __future__ import annotations

from typing import List

import numpy as np

from . import _capi
from . import gpflow_lite as gpflow
from .gpflow_lite import TensorLike, _as_value
from .ortho_rbf_kernel import _col


class OrthogonalCategorical(gpflow.Kernel):
    def __init__(self, p: List, rank: int = 2, active_dims: int = None):
        super().__init__(active_dims=active_dims)
        self.num_cat = len(p)
        self.p = p
        self.variance = gpflow.Parameter(1.0, transform=gpflow.positive())
        self.W = gpflow.Parameter(np.random.uniform(size=(self.num_cat, rank)))     # tf.random.uniform (:28)
        self.kappa = gpflow.Parameter(np.ones(self.num_cat), transform=gpflow.positive())

    def _var(self) -> float:
        return float(np.asarray(_as_value(self.variance)).reshape(-1)[0])

    def output_covariance(self):
        """variance * (A - (Ap)(Ap)^T / p^T A p), A = W W^T + diag(kappa)  (ortho_categorical_kernel.py:34-42)."""
        B, _ = _capi.categorical_table_unit(self.W.numpy(), self.kappa.numpy(), self.p)
        return B * self._var()

    def output_variance(self):
        return np.diag(self.output_covariance()).copy()

    def dim_spec(self, active_dim: int = 0) -> dict:
        return dict(type="categorical", p=np.asarray(self.p, dtype=np.float64).reshape(-1, 1), W=self.W.numpy(),
                    kappa=self.kappa.numpy(), variance=self._var(), active_dim=active_dim)

    def _spec(self) -> dict:
        return dict(dims=[self.dim_spec(0)], order_variances=[0.0, 1.0], max_interaction_depth=1,
                    share_var_across_orders=True)

    def K(self, X, X2=None):
        X = _col(X)
        X2 = None if X2 is None else _col(X2, "X2")
        return TensorLike(_capi.default_context().gram(_capi.KernelDesc(self._spec()), X, X2))

    def K_diag(self, X):
        return TensorLike(_capi.default_context().gram_diag(_capi.KernelDesc(self._spec()), _col(X)))

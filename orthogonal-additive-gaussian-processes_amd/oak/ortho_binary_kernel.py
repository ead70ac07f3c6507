"""Constrained kernel for {0,1} inputs (host mirror of oak/ortho_binary_kernel.py:13-59): a 2x2 table lookup."""
from __future__ import annotations

import numpy as np

from . import _capi
from . import gpflow_lite as gpflow
from .gpflow_lite import TensorLike, _as_value
from .ortho_rbf_kernel import _col


class OrthogonalBinary(gpflow.Kernel):
    def __init__(self, p0: float = 0.5, active_dims: int = None):
        super().__init__(active_dims=active_dims)
        self.variance = gpflow.Parameter(1.0, transform=gpflow.positive())
        self.p0 = p0

    def _var(self) -> float:
        return float(np.asarray(_as_value(self.variance)).reshape(-1)[0])

    def output_covariance(self):
        """variance * [[p1^2, -p0 p1], [-p0 p1, p0^2]]  (ortho_binary_kernel.py:29-33)."""
        q0, q1 = self.p0, 1.0 - self.p0
        return self._var() * np.array([[q1 * q1, -q0 * q1], [-q0 * q1, q0 * q0]])

    def output_variance(self):
        q0, q1 = self.p0, 1.0 - self.p0
        return self._var() * np.array([q1 * q1, q0 * q0])

    def dim_spec(self, active_dim: int = 0) -> dict:
        return dict(type="binary", p0=float(self.p0), variance=self._var(), active_dim=active_dim)

    def _spec(self) -> dict:
        return dict(dims=[self.dim_spec(0)], order_variances=[0.0, 1.0], max_interaction_depth=1,
                    share_var_across_orders=True)

    def K(self, X, X2=None):
        X = _col(X)
        X2 = None if X2 is None else _col(X2, "X2")
        return TensorLike(_capi.default_context().gram(_capi.KernelDesc(self._spec()), X, X2))

    def K_diag(self, X):
        return TensorLike(_capi.default_context().gram_diag(_capi.KernelDesc(self._spec()), _col(X)))

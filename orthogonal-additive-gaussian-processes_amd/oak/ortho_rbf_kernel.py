"""Constrained squared-exponential kernel (host mirror of oak/ortho_rbf_kernel.py:20-177).

K(x, z) = k(x, z) - c(x) c(z) / v with c = cov_X_s, v = var_s under the input measure; all evaluation happens in the
HIP library (featurize + fused Gram kernels), this class only describes the kernel.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _capi
from . import gpflow_lite as gpflow
from .gpflow_lite import TensorLike, _as_value
from .input_measures import EmpiricalMeasure, GaussianMeasure, Measure, MOGMeasure, UniformMeasure


def _col(X, name="X"):
    X = np.asarray(X, dtype=np.float64)
    if X.ndim != 2 or X.shape[1] != 1:
        raise ValueError(f"{name} must have shape [N, 1], got {X.shape}")   # tf.debugging.assert_shapes analogue (:50,83)
    return X


class OrthogonalRBFKernel(gpflow.Kernel):
    def __init__(self, base_kernel: gpflow.RBF, measure: Measure, active_dims=None):
        super().__init__(active_dims=active_dims)
        self.base_kernel, self.measure = base_kernel, measure
        if not isinstance(base_kernel, gpflow.RBF):
            raise NotImplementedError          # ortho_rbf_kernel.py:34-35
        if not isinstance(measure, (UniformMeasure, GaussianMeasure, EmpiricalMeasure, MOGMeasure)):
            raise NotImplementedError          # ortho_rbf_kernel.py:36-45

    # -- description ---------------------------------------------------------------------------------
    def dim_spec(self, active_dim: int = 0) -> dict:
        return dict(type="rbf",
                    lengthscale=float(np.asarray(_as_value(self.base_kernel.lengthscales)).reshape(-1)[0]),
                    variance=float(np.asarray(_as_value(self.base_kernel.variance)).reshape(-1)[0]),
                    measure=self.measure.as_tuple(), active_dim=active_dim)

    def _spec(self) -> dict:
        # a single sub-kernel: K = 0*e_0 + 1*e_1
        return dict(dims=[self.dim_spec(0)], order_variances=[0.0, 1.0], max_interaction_depth=1,
                    share_var_across_orders=True)

    # -- measure integrals (ortho_rbf_kernel.py:47-152) ----------------------------------------------
    def cov_X_s(self, X):
        X = _col(X)
        c, _ = _capi.default_context().measure_cov(_capi.KernelDesc(self._spec()), 0, X)
        return TensorLike(c[:, None])

    def var_s(self):
        _, v = _capi.default_context().measure_cov(_capi.KernelDesc(self._spec()), 0, np.zeros((1, 1)))
        return np.float64(v)

    # -- Gram (ortho_rbf_kernel.py:157-177) -----------------------------------------------------------
    def K(self, X: np.ndarray, X2: Optional[np.ndarray] = None) -> np.ndarray:
        X = _col(X)
        X2 = None if X2 is None else _col(X2, "X2")
        return TensorLike(_capi.default_context().gram(_capi.KernelDesc(self._spec()), X, X2))

    def K_diag(self, X):
        return TensorLike(_capi.default_context().gram_diag(_capi.KernelDesc(self._spec()), _col(X)))

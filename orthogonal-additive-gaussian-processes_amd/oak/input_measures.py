"""Input measures of the constrained kernels (mirror of oak/input_measures.py:16-78): plain data holders."""
from __future__ import annotations

import numpy as np


class Measure:
    """Base class; concrete measures only carry their parameters."""

    kind = None


class UniformMeasure(Measure):
    """Uniform density on [a, b] (oak/input_measures.py:20-28)."""

    kind = "uniform"

    def __init__(self, a: float, b: float):
        self.a, self.b = a, b

    def as_tuple(self):
        return ("uniform", float(self.a), float(self.b))


class GaussianMeasure(Measure):
    """N(mu, var) (oak/input_measures.py:31-39)."""

    kind = "gaussian"

    def __init__(self, mu: float, var: float):
        self.mu, self.var = mu, var

    def as_tuple(self):
        return ("gaussian", float(self.mu), float(self.var))


class EmpiricalMeasure(Measure):
    """Weighted Dirac measure on `location` [K, 1]; uniform weights by default; weights must sum to one
    (oak/input_measures.py:42-57)."""

    kind = "empirical"

    def __init__(self, location: np.ndarray, weights: np.ndarray = None):
        self.location = location
        if weights is None:
            weights = np.full((location.shape[0], 1), 1.0 / len(location))
        total = np.sum(weights)
        assert np.isclose(total, 1.0, atol=1e-6), f"not close to 1 {total}"
        self.weights = weights

    def as_tuple(self):
        return ("empirical", np.asarray(self.location, dtype=np.float64).reshape(-1, 1),
                np.asarray(self.weights, dtype=np.float64).reshape(-1, 1))


class MOGMeasure(Measure):
    """Mixture of K one-dimensional Gaussians; all three arrays have shape (K,) (oak/input_measures.py:60-78)."""

    kind = "mog"

    def __init__(self, means: np.ndarray, variances: np.ndarray, weights: np.ndarray):
        means, variances, weights = np.asarray(means), np.asarray(variances), np.asarray(weights)
        if not (means.ndim == variances.ndim == weights.ndim == 1 and len(means) == len(variances) == len(weights)):
            raise ValueError("means, variances and weights must all have shape (K,)")
        total = weights.sum()
        assert np.isclose(total, 1.0, atol=1e-6), f"Weights not close to 1 {total}"
        self.means, self.variances, self.weights = means.astype(float), variances.astype(float), weights

    def as_tuple(self):
        return ("mog", self.means, self.variances, np.asarray(self.weights, dtype=np.float64))

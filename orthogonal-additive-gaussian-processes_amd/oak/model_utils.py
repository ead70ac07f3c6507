"""Model construction, fitting, prediction and attribution: host mirror of oak/model_utils.py:31-770.

Same call surface as the reference (``create_model_oak``, ``oak_model.fit/optimise/predict/get_loglik/get_sobol``,
``save_model/load_model``); every Gram, Cholesky, solve and Sobol reduction is executed by the HIP library.
``oak_model.plot`` (matplotlib, model_utils.py:526-700) is out of scope.
"""
from __future__ import annotations

import os
import time
from pathlib import Path
from typing import Callable, List, Optional, Union

import numpy as np

from . import _capi
from . import gpflow_lite as gpflow
from .gpflow_lite import GPR, SGPR, GPModel, InducingPoints, set_trainable
from .input_measures import MOGMeasure
from .normalising_flow import Normalizer
from .oak_kernel import OAKKernel, get_list_representation
from .utils import compute_sobol_oak, initialize_kmeans_with_categorical, kmeans_centres


def get_kmeans_centers(X: np.ndarray, K: int = 500) -> np.ndarray:
    """K-means centres of X (oak/model_utils.py:31-41): k-means++ seeds + Lloyd iterations on the device."""
    np.random.seed(44)
    return kmeans_centres(X, K, random_state=0)


def save_model(model: GPModel, filename: Path) -> None:
    """npz with a single ``hyperparams`` entry listing the trainable parameter values -- every parameter for an SVGP model
    (oak/model_utils.py:44-64)."""
    values = [p.numpy() for p in (model.parameters if isinstance(model, gpflow.models.SVGP) else model.trainable_parameters)]
    filename = Path(filename)
    os.makedirs(filename.parents[0], exist_ok=True)
    arr = np.empty(len(values), dtype=object)
    for i, v in enumerate(values):
        arr[i] = v
    np.savez(filename, hyperparams=arr)


def load_model(model: GPModel, filename: Path, load_all_parameters=False) -> None:
    """Positional assignment back into the model's parameters (oak/model_utils.py:67-87)."""
    stored = np.load(str(filename), allow_pickle=True)["hyperparams"]
    targets = model.parameters if load_all_parameters else model.trainable_parameters
    for i, p in enumerate(targets):
        p.assign(stored[i])


def create_model_oak(
    data,
    max_interaction_depth: int = 2,
    constrain_orthogonal: bool = True,
    inducing_pts: np.ndarray = None,
    optimise=False,
    zfixed=True,
    p0=None,
    p=None,
    lengthscale_bounds=None,
    empirical_locations: Optional[List[float]] = None,
    empirical_weights: Optional[List[float]] = None,
    use_sparsity_prior: bool = True,
    gmm_measures: Optional[List[MOGMeasure]] = None,
    share_var_across_orders: Optional[bool] = True,
) -> GPModel:
    """OAK kernel wrapped in SGPR (inducing points given) or GPR (oak/model_utils.py:90-176)."""
    num_dims = np.asarray(data[0]).shape[1]
    p0 = [None] * num_dims if p0 is None else p0
    p = [None] * num_dims if p is None else p
    base_kernels = [gpflow.RBF if (p0[d] is None and p[d] is None) else None for d in range(num_dims)]
    k = OAKKernel(
        base_kernels,
        num_dims=num_dims,
        max_interaction_depth=max_interaction_depth,
        constrain_orthogonal=constrain_orthogonal,
        p0=p0,
        p=p,
        lengthscale_bounds=lengthscale_bounds,
        empirical_locations=empirical_locations,
        empirical_weights=empirical_weights,
        gmm_measures=gmm_measures,
        share_var_across_orders=share_var_across_orders,
    )
    if inducing_pts is not None:
        model = SGPR(data, mean_function=None, kernel=k, inducing_variable=InducingPoints(inducing_pts))
        if zfixed:
            set_trainable(model.inducing_variable, False)
    else:
        model = GPR(data, mean_function=None, kernel=k)
    if use_sparsity_prior:
        print("Using sparsity prior")
        if share_var_across_orders:
            for v in model.kernel.variances:
                v.prior = gpflow.Gamma(1.0, 0.2)          # :161-165
    model.likelihood.variance.assign(0.01)               # :167
    if optimise:
        t_start = time.time()
        gpflow.Scipy().minimize(model.training_loss_closure(), model.trainable_variables, method="BFGS")
        gpflow.print_summary(model, fmt="notebook")
        print(f"Training took {time.time() - t_start:.1f} seconds.")
    return model


def apply_normalise_flow(X, input_flows: List[Normalizer]):
    """Column-wise application of the fitted flows; columns without a flow pass through (oak/model_utils.py:179-191)."""
    X = np.asarray(X, dtype=np.float64)
    kind, params = _flow_columns(X.shape[1], input_flows)
    return _capi.default_context().flow_forward(X, kind, params)


def _flow_columns(D: int, input_flows, affine=None):
    """(kind[D], params[D, 5]) of oak_flow_forward for a list of fitted flows; ``affine`` maps column -> (mean, std)."""
    kind = np.zeros(D, dtype=np.int32)
    params = np.zeros((D, 5))
    for ii in range(D):
        f = input_flows[ii]
        if f is not None:
            b = f.bijector
            kind[ii] = 2 if b.log else 1
            params[ii] = (b.offset, float(f.scale.numpy()), float(f.shift.numpy()), float(f.skewness.numpy()), float(f.tailweight.numpy()))
    for ii, (mean, std) in (affine or {}).items():
        kind[ii] = 3
        params[ii] = (mean, std, 0.0, 0.0, 0.0)
    return kind, params


class _Standardizer:
    """sklearn.preprocessing.StandardScaler semantics (population std), NumPy only."""

    def fit(self, A):
        A = np.asarray(A, dtype=np.float64)
        self.mean_ = A.mean(axis=0)
        self.var_ = A.var(axis=0)
        self.scale_ = np.where(self.var_ > 0, np.sqrt(self.var_), 1.0)
        return self

    def transform(self, A):
        return (np.asarray(A, dtype=np.float64) - self.mean_) / self.scale_

    def inverse_transform(self, A):
        return np.asarray(A, dtype=np.float64) * self.scale_ + self.mean_


class oak_model:
    """Scikit-style wrapper (oak/model_utils.py:194-524); constructor arguments as in the reference (:195-208)."""

    def __init__(
        self,
        max_interaction_depth=2,
        num_inducing=200,
        lengthscale_bounds=[1e-3, 1e3],
        binary_feature: Optional[List[int]] = None,
        categorical_feature: Optional[List[int]] = None,
        empirical_measure: Optional[List[int]] = None,
        use_sparsity_prior: bool = True,
        gmm_measure: Optional[List[int]] = None,
        sparse: bool = False,
        use_normalising_flow: bool = True,
        share_var_across_orders: bool = True,
    ):
        self.max_interaction_depth = max_interaction_depth
        self.num_inducing = num_inducing
        self.lengthscale_bounds = lengthscale_bounds
        self.binary_feature = binary_feature
        self.categorical_feature = categorical_feature
        self.use_sparsity_prior = use_sparsity_prior
        self.empirical_measure = empirical_measure
        self.gmm_measure = gmm_measure
        self.sparse = sparse
        self.use_normalising_flow = use_normalising_flow
        self.share_var_across_orders = share_var_across_orders
        # filled in by fit()
        self.input_flows = None
        self.scaler_y = None
        self.Y_scaled = None
        self.X_scaled = None
        self.alpha = None
        self.continuous_index = None
        self.binary_index = None
        self.categorical_index = None
        self.empirical_locations = None
        self.empirical_weights = None
        self.estimated_gmm_measures = None

    def fit(self, X, Y, optimise: bool = True, initialise_inducing_points: bool = True):
        X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
        self.xmin, self.xmax = X.min(0), X.max(0)
        self.num_dims = X.shape[1]
        (self.continuous_index, self.binary_index, self.categorical_index, p0, p) = _calculate_features(
            X, categorical_feature=self.categorical_feature, binary_feature=self.binary_feature)
        if self.empirical_measure is not None and not set(self.empirical_measure).issubset(self.continuous_index):
            raise ValueError(f"Empirical measure={self.empirical_measure} should only be used on non-binary/categorical "
                             f"inputs {self.continuous_index}")
        if self.gmm_measure is not None:
            if len(self.gmm_measure) != self.num_dims:
                return ValueError(f"Must specify number of components for each inputs dimension 1..{X.shape[0]}")  # sic (:283)
            idx_gmm = np.flatnonzero(self.gmm_measure)
            if not set(idx_gmm).issubset(self.continuous_index):
                raise ValueError(f"GMM measure on inputs {idx_gmm} should only be used on continuous inputs {self.continuous_index}")

        self.estimated_gmm_measures = [None] * self.num_dims
        if self.gmm_measure is not None:
            for i_dim in np.flatnonzero(self.gmm_measure):
                self.estimated_gmm_measures[i_dim] = estimate_one_dim_gmm(K=self.gmm_measure[i_dim], X=X[:, i_dim])
        self.empirical_locations = [None] * self.num_dims
        self.empirical_weights = [None] * self.num_dims

        # input scaling: one flow per continuous feature without an empirical / GMM measure (:305-317)
        self.input_flows = [None] * self.num_dims
        for i in self.continuous_index:
            if self.empirical_measure is not None and i in self.empirical_measure:
                continue
            if self.estimated_gmm_measures[i] is not None:
                continue
            if self.use_normalising_flow:
                n = Normalizer(X[:, i])
                gpflow.Scipy().minimize(n.KL_objective, n.trainable_variables)
                self.input_flows[i] = n

        self.alpha = None
        self.scaler_y = _Standardizer().fit(Y)
        self.Y_scaled = self.scaler_y.transform(Y)
        if self.empirical_measure is not None:
            self.scaler_X_empirical = _Standardizer().fit(X[:, self.empirical_measure])
        if not self.use_normalising_flow:
            self.scaler_X_continuous = _Standardizer().fit(X[:, self.continuous_index])
        self.X_scaled = self._transform_x(X)

        if self.empirical_measure is not None:   # locations / weights after scaling (:334-344)
            for ii in self.empirical_measure:
                loc, cnt = np.unique(self.X_scaled[:, ii], return_counts=True)
                self.empirical_weights[ii] = (cnt / cnt.sum()).reshape(-1, 1)
                self.empirical_locations[ii] = loc.reshape(-1, 1)

        self._check_untransformed_columns(X)

        Z = None
        if X.shape[0] > 1000 or self.sparse:     # sparse GP above 1000 rows (:374)
            if initialise_inducing_points:
                if (p0 is None) and (p is None):
                    print("all features are continuous")
                    # reference: KMeans(n_clusters=num_inducing, random_state=0).fit(X_scaled) (:377-383); Lloyd on the device
                    Z = kmeans_centres(self.X_scaled, self.num_inducing, random_state=0)
                else:
                    Z = initialize_kmeans_with_categorical(
                        self.X_scaled, binary_index=self.binary_index, categorical_index=self.categorical_index,
                        continuous_index=self.continuous_index, n_clusters=self.num_inducing)
            else:
                Z = self.X_scaled[: self.num_inducing, :]

        self.m = create_model_oak(
            (self.X_scaled, self.Y_scaled),
            max_interaction_depth=self.max_interaction_depth,
            inducing_pts=Z,
            optimise=optimise,
            p0=p0,
            p=p,
            lengthscale_bounds=self.lengthscale_bounds,
            use_sparsity_prior=self.use_sparsity_prior,
            empirical_locations=self.empirical_locations,
            empirical_weights=self.empirical_weights,
            gmm_measures=self.estimated_gmm_measures,
            share_var_across_orders=self.share_var_across_orders,
        )

    def _check_untransformed_columns(self, X):
        """The input scaling must leave discrete and measure-carrying columns alone (the checks of oak/model_utils.py:346-371):
        binary, categorical and GMM-measure columns pass through unchanged; empirical-measure columns are only standardised,
        so their inverse transformer maps them back."""
        passthrough = {"binary": list(self.binary_index), "categorical": list(self.categorical_index),
                       "GMM measure": [] if self.gmm_measure is None else [int(i) for i in np.flatnonzero(self.gmm_measure)]}
        for what, cols in passthrough.items():
            if cols and not np.allclose(self.X_scaled[:, cols], X[:, cols]):
                raise AssertionError(f"Flow applied to {what} inputs")
        for i in (self.empirical_measure or []):
            if not np.allclose(self._get_x_inverse_transformer(i)(self.X_scaled[:, i]), X[:, i]):
                raise AssertionError("Flow applied to empirical measure inputs")

    def optimise(self, compile: bool = True):
        print("Model prior to optimisation")
        gpflow.print_summary(self.m, fmt="notebook")
        self.alpha = None
        t_start = time.time()
        gpflow.Scipy().minimize(self.m.training_loss_closure(), self.m.trainable_variables, method="BFGS", compile=compile)
        gpflow.print_summary(self.m, fmt="notebook")
        print(f"Training took {time.time() - t_start:.1f} seconds.")

    def predict(self, X, clip=False):
        X = np.asarray(X, dtype=np.float64)
        X_scaled = self._transform_x(np.clip(X, self.xmin, self.xmax) if clip else X)
        try:
            y_pred = self.m.predict_f(X_scaled)[0].numpy()
            return self.scaler_y.inverse_transform(y_pred)[:, 0]
        except ValueError:
            print("test X is outside the range of training input, try clipping X.")

    def get_loglik(self, X, y, clip=False):
        X = np.asarray(X, dtype=np.float64)
        X_scaled = self._transform_x(np.clip(X, self.xmin, self.xmax) if clip else X)
        return float(np.mean(self.m.predict_log_density((X_scaled, self.scaler_y.transform(y)))))

    def _transform_x(self, X):
        """Flows, then the standard scalers of the empirical-measure / no-flow columns (oak/model_utils.py:462-476): one
        elementwise pass on the device (a column has a flow or a scaler, never both)."""
        X = np.asarray(X, dtype=np.float64)
        affine = {}
        if self.empirical_measure is not None:
            for j, col in enumerate(self.empirical_measure):
                affine[col] = (float(self.scaler_X_empirical.mean_[j]), float(self.scaler_X_empirical.scale_[j]))
        if not self.use_normalising_flow:
            for j, col in enumerate(self.continuous_index):
                affine[col] = (float(self.scaler_X_continuous.mean_[j]), float(self.scaler_X_continuous.scale_[j]))
        for col in affine:
            assert self.input_flows[col] is None, "a column has either a flow or a scaler"
        kind, params = _flow_columns(X.shape[1], self.input_flows, affine)
        return _capi.default_context().flow_forward(X, kind, params)

    def _get_x_inverse_transformer(self, i: int) -> Optional[Union[Normalizer, Callable]]:
        assert i in self.continuous_index
        if self.empirical_measure is not None and i in self.empirical_measure:
            j = self.empirical_measure.index(i)
            mean_i, std_i = self.scaler_X_empirical.mean_[j], np.sqrt(self.scaler_X_empirical.var_[j])
            return lambda x: x * std_i + mean_i
        if self.gmm_measure is not None and i in self.gmm_measure:
            return None
        return self.input_flows[i].bijector.inverse

    def get_sobol(self, likelihood_variance=False):
        """Normalised Sobol index of every additive term (oak/model_utils.py:499-524)."""
        selected_dims, _ = get_list_representation(self.m.kernel, num_dims=self.num_dims)
        model_indices, sobols = compute_sobol_oak(self.m, 1, 0, share_var_across_orders=self.share_var_across_orders)
        sobols = np.asarray(sobols)
        total = np.sum(sobols) + (self.m.likelihood.variance.numpy() if likelihood_variance else 0.0)
        self.normalised_sobols = sobols / total
        self.tuple_of_indices = selected_dims[1:]
        return self.normalised_sobols

    def plot(self, transformer_y=None, X_columns=None, X_lists=None, top_n=None, likelihood_variance=False, semilogy=True, save_fig=None,
             tikz_path=None, ylim=None, quantile_range=None, log_axis=(False, False), grid_range=None, log_bin=None, num_bin=100):
        """Arguments as in the reference (oak/model_utils.py:526-545); the figures themselves are outside this build's scope."""
        raise NotImplementedError("plotting (oak/model_utils.py:526-700, plotting_utils.py) is outside this build's scope")


def _calculate_features(X, categorical_feature: List[int], binary_feature: List[int]):
    """Column typing and the empirical class probabilities the discrete sub-kernels are built from
    (behaviour of oak/model_utils.py:703-750): returns (continuous_index, binary_index, categorical_index, p0, p) where
    ``p0[j]`` = P(x_j = 0) for a binary column, ``p[j]`` = the class frequencies (C x 1, classes in sorted order) for a
    categorical one, None elsewhere; both are None altogether when no discrete column was declared."""
    X = np.asarray(X)
    D = X.shape[1]
    binary = set(binary_feature or [])
    categorical = set(categorical_feature or [])
    if binary & categorical:
        raise ValueError(f"Overlapping feature set {binary & categorical}")
    kind = np.array(["binary" if j in binary else ("categorical" if j in categorical else "continuous") for j in range(D)])
    binary_index, categorical_index, continuous_index = ([int(j) for j in np.flatnonzero(kind == k)]
                                                         for k in ("binary", "categorical", "continuous"))
    if binary_feature is None and categorical_feature is None:
        p0 = p = None
    else:
        p0, p = [None] * D, [None] * D
        for j in binary_index:
            p0[j] = 1 - X[:, j].mean()
        for j in categorical_index:
            freq = np.unique(X[:, j], return_counts=True)[1] / X.shape[0]
            if abs(freq.sum() - 1) >= 1e-6:
                raise AssertionError("class frequencies do not sum to one")
            p[j] = freq.reshape(-1, 1)
    print("indices of binary feature ", binary_index)
    print("indices of continuous feature ", continuous_index)
    print("indices of categorical feature ", categorical_index)
    return continuous_index, binary_index, categorical_index, p0, p


def estimate_one_dim_gmm(K: int, X: np.ndarray) -> MOGMeasure:
    """Spherical K-component GMM of a 1-D sample (oak/model_utils.py:753-770)."""
    from sklearn.mixture import GaussianMixture
    X = np.asarray(X)
    if X.ndim != 1:
        raise ValueError("X must be one-dimensional")
    assert K > 0
    gm = GaussianMixture(n_components=K, random_state=0, covariance_type="spherical").fit(X.reshape(-1, 1))
    assert np.allclose(gm.weights_.sum(), 1.0)
    return MOGMeasure(weights=gm.weights_, means=gm.means_.reshape(-1), variances=gm.covariances_)

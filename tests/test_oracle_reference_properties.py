"""The reference's own test properties for the hot path, re-expressed against the CPU oracle.

Every test cites the reference test it restates.  Monte-Carlo checks of the reference are replaced by numerical
quadrature of the same integral (tighter, deterministic).
"""
from functools import reduce
from itertools import combinations

import numpy as np
import pytest
from scipy import integrate, stats

from oracle import oak_oracle as o


def _rbf_dim(measure, l=1.0, var=1.0):
    return dict(type="rbf", lengthscale=l, variance=var, measure=measure)


ONE_D_KERNELS = [
    _rbf_dim(None),
    _rbf_dim(("gaussian", 0.0, 1.0)),
    dict(type="binary", p0=0.5, variance=1.0),
    _rbf_dim(("uniform", 0.0, 1.0)),
    _rbf_dim(("empirical", np.array([[0.1], [0.5], [0.5]]), np.full((3, 1), 1 / 3))),
    _rbf_dim(("mog", np.array([3.0, 2.0]), np.array([3.0, 10.0]), np.array([0.6, 0.4]))),
]


@pytest.mark.parametrize("dim", ONE_D_KERNELS)
def test_kernel_1d_diag_matches_full(dim):
    """tests/test_kernel_properties.py:57-66 -- diag(K(X,X)) == K_diag(X)."""
    X = np.array([[0.1], [0.5], [0.5]])
    if dim["type"] == "binary":
        X = np.array([[0.0], [1.0], [1.0]])
    np.testing.assert_allclose(np.diag(o.base_K(X, X, dim)), o.base_K_diag(X, dim), rtol=1e-12)
    np.testing.assert_allclose(o.base_K(X, None, dim), o.base_K(X, X, dim), rtol=1e-12)


@pytest.mark.parametrize("num_dims", [3, 4])
def test_newton_girard(num_dims):
    """tests/test_kernel_properties.py:69-86 -- Newton-Girard == brute-force sums of products."""
    rng = np.random.default_rng(num_dims)
    xx = [rng.standard_normal((2, 2)) for _ in range(num_dims)]
    result = o.compute_additive_terms(xx, num_dims)
    hard = [np.ones((2, 2))] + [reduce(np.add, map(lambda x: np.prod(x, axis=0), combinations(xx, i))) for i in range(1, num_dims + 1)]
    assert len(result) == len(hard)
    for r1, r2 in zip(result, hard):
        np.testing.assert_allclose(r1, r2, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("active_dims", [[0], [1]])
@pytest.mark.parametrize("measure", [("gaussian", 0, 1), ("uniform", 0, 1),
                                     ("empirical", np.array([[0.1], [0.5]]), np.full((2, 1), 0.5)),
                                     ("mog", np.array([3.0, 2.0]), np.array([3.0, 10.0]), np.array([0.6, 0.4]))])
def test_orthogonal_rbf_kernel_2d_with_active_dims(active_dims, measure):
    """tests/test_kernel_properties.py:89-116 -- slicing active_dims == evaluating on the sliced column."""
    X = np.array([[0.1, 0.2], [0.5, 0.5], [0.5, 0.7]])
    dim = _rbf_dim(measure, l=10.0)
    dim["active_dim"] = active_dims[0]
    spec = dict(dims=[dim], order_variances=[0.0, 1.0], max_interaction_depth=1, share_var_across_orders=True)
    Ks = o.base_K(X[:, active_dims], X[:, active_dims], dim)
    np.testing.assert_allclose(np.diag(Ks), o.base_K_diag(X[:, active_dims], dim), rtol=1e-12)
    np.testing.assert_allclose(o.oak_K(spec, X, X), Ks, rtol=1e-12)


def test_mog_equals_gaussian():
    """tests/test_orthogonality.py:152-165."""
    k_gmm = _rbf_dim(("mog", np.array([3.0, 3.0]), np.array([5.0, 5.0]), np.array([0.2, 0.8])), l=10.0)
    k_gauss = _rbf_dim(("gaussian", 3.0, 5.0), l=10.0)
    xx = np.array([[-2], [2.0], [3.0]])
    np.testing.assert_allclose(o.base_K(xx, None, k_gauss), o.base_K(xx, None, k_gmm), rtol=1e-7)


def _density(measure):
    if measure[0] == "gaussian":
        return lambda s: stats.norm.pdf(s, measure[1], np.sqrt(measure[2])), (-np.inf, np.inf)
    if measure[0] == "uniform":
        return lambda s: 1.0 / (measure[2] - measure[1]), (measure[1], measure[2])
    if measure[0] == "mog":
        mu, var, w = measure[1:4]
        return lambda s: float(np.sum(w * stats.norm.pdf(s, mu, np.sqrt(var)))), (-np.inf, np.inf)
    raise ValueError


@pytest.mark.parametrize("measure", [("gaussian", 0.0, 1.0), ("gaussian", 0.4, 2.5), ("uniform", 0.0, 1.0), ("uniform", -2.0, 3.0),
                                     ("mog", np.array([-1.0, 1.5]), np.array([0.5, 2.0]), np.array([0.3, 0.7]))])
@pytest.mark.parametrize("l", [0.7, 10.0])
def test_cov_var_and_orthogonality_by_quadrature(measure, l):
    """tests/test_orthogonality.py:27-149: cov_X_s(x) = E_s k(x,s), var_s = E_s cov_X_s(s), and the constrained kernel
    integrates to zero against the measure -- here by adaptive quadrature instead of 10^4-sample Monte Carlo."""
    dim = _rbf_dim(measure, l=l, var=1.3)
    pdf, (a, b) = _density(measure)
    base = lambda x, s: 1.3 * np.exp(-0.5 * (x - s) ** 2 / l ** 2)
    for x in (-0.8, 0.0, 0.6):
        cov_num = integrate.quad(lambda s: base(x, s) * pdf(s), a, b, epsabs=1e-13, epsrel=1e-12)[0]
        np.testing.assert_allclose(o.cov_X_s(np.array([[x]]), dim)[0, 0], cov_num, rtol=1e-9)
        k_int = integrate.quad(lambda s: o.base_K(np.array([[x]]), np.array([[s]]), dim)[0, 0] * pdf(s), a, b, epsabs=1e-13, epsrel=1e-12)[0]
        assert abs(k_int) < 1e-10
    var_num = integrate.quad(lambda s: o.cov_X_s(np.array([[s]]), dim)[0, 0] * pdf(s), a, b, epsabs=1e-13, epsrel=1e-12)[0]
    np.testing.assert_allclose(o.var_s(dim), var_num, rtol=1e-9)


def test_empirical_measure_orthogonality():
    """tests/test_orthogonality.py:98-125 -- sum_j w_j K(x, loc_j) == 0 for the empirical measure (exact)."""
    rng = np.random.default_rng(44)
    loc = np.linspace(0, 1, 10).reshape(-1, 1)
    w = rng.standard_normal((10, 1)); w /= w.sum()
    dim = _rbf_dim(("empirical", loc, w), l=10.0)
    K = o.base_K(rng.standard_normal((5, 1)), loc, dim)
    np.testing.assert_allclose(K @ w, 0.0, atol=1e-12)


def test_categorical_orthogonality():
    """tests/test_categorical_kernel.py:13-23 -- B p == 0 (samples have zero mean under p)."""
    rng = np.random.default_rng(44)
    p = np.array([0.2, 0.5, 0.3]).reshape(-1, 1)
    dim = dict(W=rng.uniform(size=(3, 2)), kappa=np.ones(3), p=p, variance=1.7)
    B = o.categorical_table(dim)
    np.testing.assert_allclose(B @ p, 0.0, atol=1e-14)
    np.testing.assert_allclose(np.diag(B), o.categorical_diag_table(dim), rtol=1e-13)


@pytest.mark.parametrize("order,D", [(0, 1), (1, 1), (1, 2), (2, 2)])
def test_kernel_components_sum_to_kernel(order, D, concrete_normalised_10_rows_data):
    """tests/test_oak_kernel.py:32-117 -- K == sum of KernelComponenent terms, incl. K_diag."""
    X, _ = concrete_normalised_10_rows_data
    x_try = X[:, 1:1 + D]
    ov = [1.3, 3.3, 4.3][: order + 1]
    spec = o.make_spec(D, order, order_variances=ov)
    subsets = o.list_representation(D, order)
    np.testing.assert_allclose(o.oak_K(spec, x_try), sum(o.component_K(spec, S, x_try) for S in subsets), rtol=1e-10)
    np.testing.assert_allclose(o.oak_K_diag(spec, x_try), sum(o.component_K_diag(spec, S, x_try) for S in subsets), rtol=1e-10)


def test_list_representation_order(concrete_normalised_10_rows_data):
    """tests/test_oak_kernel.py:120-144."""
    assert o.list_representation(2, 2) == [[], [0], [1], [0, 1]]
    X, _ = concrete_normalised_10_rows_data
    spec = o.make_spec(2, 2)
    subsets = o.list_representation(2, 2)
    np.testing.assert_allclose(o.oak_K_diag(spec, X), np.diag(o.oak_K(spec, X)), rtol=1e-12)
    np.testing.assert_allclose(o.oak_K(spec, X), np.sum([o.component_K(spec, S, X) for S in subsets], axis=0), rtol=1e-10)


@pytest.mark.parametrize("num_inducing", [0, 2])
@pytest.mark.parametrize("data", [[[0.0], [1.0], [2.0]], [[0.0, 1.0], [1.0, 1.0], [2.0, 2.0]]])
def test_objective_is_finite(data, num_inducing):
    """tests/test_oak_kernel.py:14-29."""
    X = np.array(data); y = X[:, :1]
    spec = o.make_spec(X.shape[1], 2)
    val = (o.sgpr_elbo(spec, X, y, X[:num_inducing], 0.01) if num_inducing else o.gpr_log_marginal_likelihood(spec, X, y, 0.01))
    assert np.isfinite(val)


def test_prediction_components_sum_to_mean():
    """tests/test_utils.py:42-75 -- sum_S K_S(X, Z) alpha == predict_f mean (constant variance ~ 0)."""
    rng = np.random.default_rng(44)
    N, M = 400, 30
    X = rng.normal(0, 1, (N, 3))
    y = (X[:, 0] ** 2 + X[:, 1] + X[:, 1] * X[:, 2] + rng.normal(0, 0.01, N)).reshape(-1, 1)
    Z = X[:M]
    spec = o.make_spec(3, 2, order_variances=[1e-16, 1.0, 1.0])
    alpha = o.sgpr_alpha(spec, X, y, Z, 0.01)
    comps = o.prediction_components(spec, Z, alpha, X)
    mean, _ = o.sgpr_predict_f(spec, X, y, Z, 0.01, X)
    np.testing.assert_allclose(np.sum(comps, axis=0), mean[:, 0], rtol=1e-7, atol=1e-9)

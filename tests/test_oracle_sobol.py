"""Sobol path of the oracle against the reference's own checks (tests/test_sobol.py, tests/test_sobol_oak_kernel.py)."""
import numpy as np
import pytest
from scipy import integrate, stats

from oracle import oak_oracle as o


def _kk1(x, y, l=1.0):
    return np.exp(-((x - y) ** 2) / (2 * l ** 2))


def _kk2(x, y, l=1.0, d=1.0, mu=0.0):
    return l * np.sqrt(l ** 2 + 2 * d ** 2) / (l ** 2 + d ** 2) * np.exp(-((x - mu) ** 2 + (y - mu) ** 2) / (2 * (l ** 2 + d ** 2)))


@pytest.mark.parametrize("l,delta,mu", [(1.0, 1.0, 0.0), (0.6, 1.4, 0.3)])
def test_f1_f2_f4_against_quadrature(l, delta, mu):
    """tests/test_sobol.py:34-140 (100 000-sample MC, TOL 1e-3) -> adaptive quadrature, 1e-10."""
    z0, z1 = -0.75, 1.31
    pdf = lambda s: stats.norm.pdf(s, mu, delta)
    q = lambda f: integrate.quad(lambda s: f(s) * pdf(s), -np.inf, np.inf, epsabs=1e-14, epsrel=1e-12)[0]
    np.testing.assert_allclose(o.f1(z0, z1, 1, l, delta, mu), q(lambda s: _kk1(z0, s, l) * _kk1(z1, s, l)), rtol=1e-9)
    np.testing.assert_allclose(o.f2(z0, z1, 1, l, delta, mu), q(lambda s: _kk1(z0, s, l) * _kk2(z1, s, l, delta, mu)), rtol=1e-9)
    np.testing.assert_allclose(o.f4(z0, z1, 1, l, delta, mu), q(lambda s: _kk2(z0, s, l, delta, mu) * _kk2(z1, s, l, delta, mu)), rtol=1e-9)


def test_compute_L_is_the_kernel_integral():
    """compute_L (utils.py:221-240) == int k_d(x,s) k_d(s,y) N(s;0,1) ds for the Gaussian-constrained kernel."""
    dim = dict(type="rbf", lengthscale=0.9, variance=1.0, measure=("gaussian", 0.0, 1.0))
    X = np.array([[-0.4], [0.2], [1.1]])
    L = o.compute_L(X, 0.9, 1.0, 0, 1.0, 0.0)
    for i in range(3):
        for j in range(3):
            val = integrate.quad(lambda s: o.base_K(X[i:i + 1], np.array([[s]]), dim)[0, 0] * o.base_K(np.array([[s]]), X[j:j + 1], dim)[0, 0]
                                 * stats.norm.pdf(s), -np.inf, np.inf, epsabs=1e-14, epsrel=1e-12)[0]
            np.testing.assert_allclose(L[i, j], val, rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("p", (0.0, 0.77, 1.0))
def test_compute_L_binary_kernel(p):
    """tests/test_sobol.py:186-208 -- exact identity with the binary kernel, TOL 1e-16."""
    rng = np.random.default_rng(3)
    X = rng.binomial(1, p, 300).reshape(-1, 1).astype(float)
    L = o.compute_L_binary_kernel(X, p, 1, 0)
    dim = dict(type="binary", p0=p, variance=1.0)
    x0, x1 = np.zeros((1, 1)), np.ones((1, 1))
    L1 = o.base_K(X, x0, dim) @ o.base_K(x0, X, dim) * p + o.base_K(X, x1, dim) @ o.base_K(x1, X, dim) * (1 - p)
    assert np.max(np.abs(L - L1)) < 1e-15


def test_compute_L_categorical_is_weighted_gram():
    rng = np.random.default_rng(5)
    p = np.array([0.2, 0.5, 0.3]).reshape(-1, 1)
    W, kappa = rng.uniform(size=(3, 2)), np.array([1.0, 0.5, 2.0])
    X = rng.integers(0, 3, 40).reshape(-1, 1).astype(float)
    L = o.compute_L_categorical_kernel(X, W, kappa, p, 1.3, 0)
    dim = dict(type="categorical", p=p, W=W, kappa=kappa, variance=1.3)
    cats = np.arange(3.0).reshape(-1, 1)
    Kc = o.base_K(cats, X, dim)
    np.testing.assert_allclose(L, Kc.T @ (Kc * p), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("is_sgpr", (False, True))
def test_sobol_analytic_2_4_1(is_sgpr):
    """tests/test_sobol_oak_kernel.py:31-126 -- y = x0^2 + 2 x1 + x0 x1 has Sobol indices [2, 4, 1] (1 decimal),
    hyper-parameters fixed at the values the reference test assigns (:65-72)."""
    rng = np.random.default_rng(0)
    N = 500
    X = rng.normal(0, 1, (N, 2))
    Y = (X[:, 0] ** 2 + X[:, 1] * 2 + X[:, 0] * X[:, 1]).reshape(-1, 1)
    spec = o.make_spec(2, 2, lengthscales=[2.91, 9.20], order_variances=[0.76, 96.935, 128.27])
    if is_sgpr:
        from sklearn.cluster import KMeans
        Z = KMeans(n_clusters=300, random_state=0, n_init=2).fit(X).cluster_centers_
        alpha = o.sgpr_alpha(spec, X, Y, Z, 0.01)
        subsets, sobol = o.compute_sobol_oak(spec, Z, alpha)
    else:
        alpha = o.gpr_alpha(spec, X, Y, 0.01)
        subsets, sobol = o.compute_sobol_oak(spec, X, alpha)
    assert subsets == [[0], [1], [0, 1]]
    np.testing.assert_array_almost_equal(sobol, np.array([2.0, 4.0, 1.0]), decimal=1)


@pytest.mark.parametrize("both_binary", (False, True))
def test_sobol_with_binary_closed_form(both_binary):
    """tests/test_sobol_oak_kernel.py:246-365 (full-GP branch)."""
    rng = np.random.default_rng(42)
    delta, N, p1, p2 = 1.0, 200, 0.5, 0.9
    X1 = rng.binomial(1, p1, N).reshape(N, 1)
    X2 = rng.binomial(1, p2, N).reshape(N, 1) if both_binary else rng.normal(0, 1, N).reshape(N, 1)
    X = np.concatenate((X1, X2), 1).astype(float)
    Y = (X[:, 0] + X[:, 1] + X[:, 0] * X[:, 1] + rng.normal(0, 0.1, N)).reshape(-1, 1)
    Y = Y - Y.mean()
    p0 = [1 - p1, 1 - p2] if both_binary else [1 - p1, None]
    spec = o.make_spec(2, 2, p0=p0, lengthscales=None if both_binary else [1.0, 9.20])
    alpha = o.gpr_alpha(spec, X, Y, 0.01)
    subsets, sobol = o.compute_sobol_oak(spec, X, alpha)
    assert subsets == [[0], [1], [0, 1]] and np.all(np.array(sobol) >= 0)
    if both_binary:
        s1, s2 = (1 + p2) ** 2 * p1 * (1 - p1), (1 + p1) ** 2 * p2 * (1 - p2)
        tot = p1 - p1 ** 2 + p2 - p2 ** 2 + 5 * p1 * p2 - p1 ** 2 * p2 ** 2 - 2 * p1 ** 2 * p2 - 2 * p1 * p2 ** 2
        expect = [s1, s2, tot - s1 - s2]
    else:
        s1, s2 = p1 * (1 - p1), delta * (1 + p1) ** 2
        expect = [s1, s2, delta + p1 * (1 - p1) + 3 * p1 * delta - s1 - s2]
    np.testing.assert_array_almost_equal(sobol, np.array(expect), decimal=1)


def test_sobol_empirical_equals_sample_variance():
    """tests/test_sobol_oak_kernel.py:129-155 -- with the empirical measure the Sobol integral is the sample variance."""
    rng = np.random.default_rng(1)
    x = rng.normal(0, 1, (10, 1))
    y = x ** 2 + np.cos(x) + rng.normal(0, 0.1, (10, 1))
    dim = dict(type="rbf", lengthscale=1.0, variance=1.0, measure=("empirical", x, np.ones(x.shape) / 10))
    spec = dict(dims=[dim], order_variances=[0.0, 1.0], max_interaction_depth=1, share_var_across_orders=True)
    mean, _ = o.gpr_predict_f(spec, x, y, 1.0, x)
    alpha = o.gpr_alpha(spec, x, y, 1.0)
    L = o.compute_L_empirical_measure(x, np.ones(x.shape) / 10, dim, x)
    np.testing.assert_array_almost_equal(np.var(mean), float((alpha.T @ L @ alpha)[0, 0]), decimal=5)


# ---- reference-EXECUTED vectors (tests/golden/make_reference_golden.py: the reference's own f1..f4 / compute_L /
# compute_L_binary_kernel, run in the build container; inputs + outputs only) -------------------------------------------
def _reference_fixture():
    from pathlib import Path
    d = np.load(Path(__file__).resolve().parent / "golden" / "reference_sobol_L.npz")
    assert str(d["label"]) == "reference-executed: f1-f4 / compute_L / compute_L_binary_kernel only"
    return d


def test_oracle_closed_forms_match_reference_executed_vectors():
    d = _reference_fixture()
    x, y = d["f_x"], d["f_y"]
    for k, fn in enumerate((o.f1, o.f2, o.f3, o.f4), start=1):
        for p, ref in zip(d["f_params"], d[f"f{k}"]):
            np.testing.assert_allclose(fn(x, y, *p), ref, rtol=1e-15, atol=0)
    for p, ref in zip(d["L_params"], d["L"]):
        got = o.compute_L(d["L_X"], p[0], p[1], int(p[2]), p[3], p[4])
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-15 * np.abs(ref).max())
    for p, ref in zip(d["Lb_params"], d["Lb"]):
        got = o.compute_L_binary_kernel(d["Lb_X"], p[0], p[1], int(p[2]))
        np.testing.assert_allclose(got, ref, rtol=1e-15, atol=1e-17)

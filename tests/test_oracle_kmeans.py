"""The k-means oracle against scikit-learn itself (the reference's dependency for inducing-point initialisation,
/root/reference/oak/model_utils.py:31-41): same seeds -> same centres, labels, inertia and iteration count."""
import numpy as np
import pytest

from oracle import kmeans_oracle as ko

sklearn_cluster = pytest.importorskip("sklearn.cluster")


def _data(N, D, K, seed):
    rng = np.random.default_rng(seed)
    centres = rng.normal(size=(K, D)) * 3.0
    X = centres[rng.integers(0, K, N)] + rng.normal(size=(N, D))
    seeds = X[rng.choice(N, K, replace=False)].copy()
    return X, seeds


@pytest.mark.parametrize("N,D,K,seed", [(500, 2, 5, 0), (2000, 8, 20, 1), (3000, 16, 50, 2), (800, 3, 40, 3)])
def test_lloyd_matches_sklearn(N, D, K, seed):
    X, seeds = _data(N, D, K, seed)
    km = sklearn_cluster.KMeans(n_clusters=K, init=seeds, n_init=1, algorithm="lloyd", max_iter=300, tol=1e-4).fit(X)
    C, labels, inertia, n_iter = ko.lloyd(X, seeds, 300, ko.sklearn_tolerance(X, 1e-4))
    assert n_iter == km.n_iter_
    np.testing.assert_array_equal(labels, km.labels_)
    np.testing.assert_allclose(C, km.cluster_centers_, rtol=0, atol=1e-10)     # sklearn centres X first: rounding only
    assert abs(inertia - km.inertia_) <= 1e-9 * km.inertia_


def test_lloyd_max_iter_and_strict_convergence():
    X, seeds = _data(1000, 4, 10, 5)
    C1, l1, i1, n1 = ko.lloyd(X, seeds, 1, 0.0)
    assert n1 == 1
    km = sklearn_cluster.KMeans(n_clusters=10, init=seeds, n_init=1, algorithm="lloyd", max_iter=1, tol=0.0).fit(X)
    np.testing.assert_allclose(C1, km.cluster_centers_, atol=1e-10)
    np.testing.assert_array_equal(l1, km.labels_)
    # tol = 0: runs to strict convergence; a further iteration from the result is a fixed point
    C, l, inertia, n = ko.lloyd(X, seeds, 300, 0.0)
    C2, l2, inertia2, n2 = ko.lloyd(X, C, 300, 0.0)
    np.testing.assert_array_equal(l, l2)
    np.testing.assert_allclose(C, C2, atol=1e-13)


def test_empty_cluster_relocated():
    rng = np.random.default_rng(9)
    X = rng.normal(size=(300, 2))
    seeds = np.vstack([X[:4], [[50.0, 50.0]]])          # the last seed attracts no point
    C, labels, inertia, n = ko.lloyd(X, seeds, 50, 0.0)
    assert np.bincount(labels, minlength=5).min() > 0   # every cluster ends up populated
    assert np.isfinite(C).all()

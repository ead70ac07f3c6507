"""HIP Lloyd k-means (oak_kmeans) against the oracle / scikit-learn on the same seeds.  Bit-exact labels and
iteration counts; centres to 1e-10 (summation order), inertia to 1e-10 relative."""
import numpy as np
import pytest

from oracle import kmeans_oracle as ko

pytestmark = pytest.mark.gpu


def _data(N, D, K, seed, spread=3.0):
    rng = np.random.default_rng(seed)
    centres = rng.normal(size=(K, D)) * spread
    X = centres[rng.integers(0, K, N)] + rng.normal(size=(N, D))
    seeds = X[rng.choice(N, K, replace=False)].copy()
    return X, seeds


@pytest.mark.parametrize("N,D,K,seed", [
    (500, 2, 5, 0),          # dmax 8
    (2000, 8, 20, 1),
    (3001, 16, 50, 2),       # dmax 16, ragged N
    (4097, 20, 33, 3),       # dmax 32
    (1500, 40, 17, 4),       # dmax 64
    (6000, 8, 700, 5),       # K > 512: two LDS chunks of centres
    (5000, 16, 300, 6),      # K > 256 at dmax 16
    (64, 3, 64, 7),          # K == N
    (1, 1, 1, 8),
])
def test_kmeans_matches_oracle(hip, N, D, K, seed):
    X, seeds = _data(N, D, K, seed)
    tol = ko.sklearn_tolerance(X, 1e-4)
    C0, l0, i0, n0 = ko.lloyd(X, seeds, 300, tol)
    C, labels, inertia, n_iter = hip.kmeans(X, seeds, 300, tol)
    assert n_iter == n0
    np.testing.assert_array_equal(labels, l0)
    np.testing.assert_allclose(C, C0, rtol=0, atol=1e-10)
    assert abs(inertia - i0) <= 1e-10 * max(i0, 1e-300)


def test_kmeans_matches_sklearn(hip):
    sklearn_cluster = pytest.importorskip("sklearn.cluster")
    X, seeds = _data(20000, 16, 128, 11)
    km = sklearn_cluster.KMeans(n_clusters=128, init=seeds, n_init=1, algorithm="lloyd", max_iter=300, tol=1e-4).fit(X)
    C, labels, inertia, n_iter = hip.kmeans(X, seeds, 300, ko.sklearn_tolerance(X, 1e-4))
    assert n_iter == km.n_iter_
    np.testing.assert_array_equal(labels, km.labels_)
    np.testing.assert_allclose(C, km.cluster_centers_, rtol=0, atol=1e-9)
    assert abs(inertia - km.inertia_) <= 1e-9 * km.inertia_


def test_kmeans_max_iter_one_and_fixed_point(hip):
    X, seeds = _data(3000, 5, 12, 21)
    C1, l1, i1, n1 = hip.kmeans(X, seeds, 1, 0.0)
    Co, lo, io, no = ko.lloyd(X, seeds, 1, 0.0)
    assert n1 == 1 == no
    np.testing.assert_array_equal(l1, lo)
    np.testing.assert_allclose(C1, Co, atol=1e-12)
    C, l, inertia, n = hip.kmeans(X, seeds, 300, 0.0)            # tol 0 -> strict convergence
    C2, l2, inertia2, n2 = hip.kmeans(X, C, 300, 0.0)
    assert n2 == 1                                             # zero centre shift on the first M-step
    np.testing.assert_array_equal(l, l2)
    np.testing.assert_array_equal(C, C2)                         # deterministic: bitwise fixed point
    assert inertia2 == inertia


def test_kmeans_empty_cluster_relocation(hip):
    rng = np.random.default_rng(9)
    X = rng.normal(size=(300, 2))
    seeds = np.vstack([X[:4], [[50.0, 50.0]]])
    Co, lo, io, no = ko.lloyd(X, seeds, 50, 0.0)
    C, l, inertia, n = hip.kmeans(X, seeds, 50, 0.0)
    assert n == no
    np.testing.assert_array_equal(l, lo)
    np.testing.assert_allclose(C, Co, atol=1e-12)
    assert np.bincount(l, minlength=5).min() > 0


def test_kmeans_deterministic_and_strided_input(hip):
    X, seeds = _data(10000, 7, 40, 31)
    Xw = np.hstack([X, np.full((X.shape[0], 2), 123.0)])[:, :7]   # non-contiguous view -> copied by the binding
    a = hip.kmeans(Xw, seeds, 25, 0.0)
    b = hip.kmeans(X, seeds, 25, 0.0)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(np.asarray(u), np.asarray(v))


def test_kmeans_bad_arguments(hip):
    X = np.zeros((10, 3))
    with pytest.raises(ValueError):
        hip.kmeans(X, np.zeros((2, 4)))
    with pytest.raises(ValueError):
        hip.kmeans(X, np.zeros((11, 3)))          # K > N (OAK_E_ARG surfaces as ValueError, like sklearn)
    with pytest.raises(ValueError):
        hip.kmeans(np.zeros((10, 65)), np.zeros((2, 65)))   # D > 64


def test_kmeans_headline_scale_properties(hip):
    """N = 2^20, D = 16, K = 1024 (the headline inducing-point initialisation): Lloyd properties that need no oracle run.
    Inertia is non-increasing in the iteration budget; returned centres are the means of their labelled points; labels are
    nearest-centre assignments (checked on a sample)."""
    rng = np.random.default_rng(20240601)
    N, D, K = 1 << 20, 16, 1024
    X = rng.normal(size=(N, D))
    seeds = X[rng.choice(N, K, replace=False)].copy()
    prev = np.inf
    for iters in (1, 3, 6):
        C, labels, inertia, n = hip.kmeans(X, seeds, iters, 0.0)
        assert n == iters and inertia <= prev * (1 + 1e-12)
        prev = inertia
    # labels are argmin w.r.t. the returned centres (sample), inertia is their summed distance
    idx = rng.choice(N, 2048, replace=False)
    d2 = ((X[idx, None, :] - C[None, :, :]) ** 2).sum(axis=2)
    np.testing.assert_array_equal(labels[idx], np.argmin(d2, axis=1))
    # one more M-step from these labels reproduces hip's next centres
    sums = np.zeros_like(C)
    np.add.at(sums, labels, X)
    cnt = np.bincount(labels, minlength=K)
    C7, _, _, _ = hip.kmeans(X, C, 1, 0.0)
    np.testing.assert_allclose(C7[cnt > 0], (sums / np.maximum(cnt, 1)[:, None])[cnt > 0], atol=1e-11)


@pytest.mark.parametrize("N,D,K", [(5000, 6, 20), (3000, 16, 64), (2000, 1, 10)])
def test_kmeans_centres_equals_sklearn_fit(hip, N, D, K):
    """The host entry the model code calls (oak.utils.kmeans_centres) reproduces the reference's
    KMeans(n_clusters=K, random_state=0).fit(X).cluster_centers_ (/root/reference/oak/model_utils.py:38-40)."""
    sklearn_cluster = pytest.importorskip("sklearn.cluster")
    from oak.utils import kmeans_centres
    rng = np.random.default_rng(0)
    X = rng.normal(size=(N, D)) * 2 + rng.integers(0, 4, (N, 1))
    ref = sklearn_cluster.KMeans(n_clusters=K, random_state=0).fit(X).cluster_centers_
    np.testing.assert_allclose(kmeans_centres(X, K, random_state=0), ref, rtol=0, atol=1e-9)


def test_get_kmeans_centers_and_subsampled_seeding(hip, monkeypatch):
    from oak import utils, model_utils
    rng = np.random.default_rng(3)
    X = rng.normal(size=(20000, 4))
    Z = model_utils.get_kmeans_centers(X, 50)
    assert Z.shape == (50, 4) and np.isfinite(Z).all()
    # seeds drawn from a subsample: still a Lloyd fixed point of the FULL data within the tolerance
    monkeypatch.setattr(utils, "KMEANS_SEED_SAMPLE", 2000)
    Z2 = utils.kmeans_centres(X, 50, random_state=0)
    C, labels, inertia, n = hip.kmeans(X, Z2, 1, 0.0)
    assert np.abs(C - Z2).max() < 0.05
    d_full = ((X[:, None, :] - Z[None]) ** 2).sum(-1).min(1).sum()
    assert inertia < 1.05 * d_full          # as good a clustering as the all-rows seeding, within 5 %


@pytest.mark.parametrize("N,D,K,seed", [(3000, 2, 8, 0), (5000, 6, 40, 1), (20000, 16, 128, 2), (4097, 20, 30, 3),
                                         (9000, 40, 12, 4), (50, 3, 50, 5), (7, 1, 1, 6)])
def test_kmeans_plusplus_matches_sklearn(hip, N, D, K, seed):
    """Device k-means++ with the caller's RandomState picks scikit-learn's indices (the seeding KMeans.fit runs inside
    the reference's get_kmeans_centers, /root/reference/oak/model_utils.py:38-40)."""
    sklearn_cluster = pytest.importorskip("sklearn.cluster")
    rng = np.random.default_rng(100 + seed)
    X = rng.normal(size=(N, D)) * rng.uniform(0.5, 3.0, size=D)
    ref_c, ref_idx = sklearn_cluster.kmeans_plusplus(X, K, random_state=seed)
    c, idx = hip.kmeans_plusplus(X, K, random_state=seed)
    np.testing.assert_array_equal(idx, ref_idx)
    np.testing.assert_array_equal(c, X[ref_idx])


def test_kmeans_plusplus_properties_at_scale(hip):
    """N = 2^20, K = 1024: distinct, valid indices; potential well below random seeding's (needs no oracle run)."""
    rng = np.random.default_rng(7)
    N, D, K = 1 << 20, 16, 1024
    X = rng.normal(size=(N, D))
    c, idx = hip.kmeans_plusplus(X, K, random_state=0)
    assert len(np.unique(idx)) == K and idx.min() >= 0 and idx.max() < N
    np.testing.assert_array_equal(c, X[idx])
    sub = X[rng.choice(N, 20000, replace=False)]
    pot = lambda C: ((sub[:, None, :] - C[None, :256, :]) ** 2).sum(-1).min(1).sum()
    assert pot(c) < pot(X[rng.choice(N, K, replace=False)]) * 1.02


def test_inducing_point_initialisers_against_the_reference_executed_fixture():
    """tests/golden/reference_host_logic.npz: outputs of the reference's OWN ``get_kmeans_centers`` (oak/model_utils.py:31-41) and
    ``initialize_kmeans_with_binary / _with_categorical`` (oak/utils.py:533-574), executed from /root/reference in the build container
    with the installed scikit-learn.  The mirrors run the continuous part on the device (k-means++ seeding in scikit-learn's order +
    Lloyd): same centres in the same order to 1e-9, discrete columns exactly."""
    from pathlib import Path
    import sklearn
    from oak.model_utils import get_kmeans_centers
    from oak.utils import initialize_kmeans_with_binary, initialize_kmeans_with_categorical
    fx = np.load(Path(__file__).resolve().parent / "golden" / "reference_host_logic.npz")
    if str(fx["sklearn_version"]) != sklearn.__version__:
        pytest.skip(f"fixture made with scikit-learn {fx['sklearn_version']}, this is {sklearn.__version__}")
    X, K = fx["X"], int(fx["K"])
    np.testing.assert_allclose(get_kmeans_centers(X[:, [0, 2, 5]], K), fx["kmeans_centers"], rtol=0, atol=1e-9)
    Zb = initialize_kmeans_with_binary(X[:, [0, 1, 2, 4, 5]], binary_index=[1, 3], continuous_index=[0, 2, 4], n_clusters=2)
    np.testing.assert_array_equal(Zb[:, [1, 3]], fx["init_binary"][:, [1, 3]])
    np.testing.assert_allclose(Zb[:, [0, 2, 4]], fx["init_binary"][:, [0, 2, 4]], rtol=0, atol=1e-9)
    Zc = initialize_kmeans_with_categorical(X[:, [0, 3, 2, 5]], binary_index=[], categorical_index=[1], continuous_index=[0, 2, 3], n_clusters=4)
    np.testing.assert_array_equal(Zc[:, 1], fx["init_categorical"][:, 1])
    np.testing.assert_allclose(Zc[:, [0, 2, 3]], fx["init_categorical"][:, [0, 2, 3]], rtol=0, atol=1e-9)

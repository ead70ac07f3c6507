"""Pins the pieces of the path that the reference's tests leave unpinned: the ELBO scalar, the
predictive variance and the GPR log-marginal, against 50-digit mpmath restatements of their DEFINITIONS
(independent of operation order)."""
import mpmath as mp
import numpy as np
import pytest

from oracle import oak_oracle as o

mp.mp.dps = 50


def _mp(a):
    return mp.matrix(np.asarray(a, dtype=np.float64).tolist())


def _logdet_and_solve(S, b):
    L = mp.cholesky(S)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(S.rows))
    return logdet, mp.cholesky_solve(S, b)


@pytest.mark.parametrize("N,M,D,R", [(12, 5, 2, 2), (10, 4, 3, 3), (9, 3, 4, 2)])
def test_elbo_is_the_titsias_bound(N, M, D, R):
    """ELBO == log N(y | 0, Qff + s2 I) - tr(Kff - Qff) / (2 s2), Qff = Kfu Kuu^-1 Kuf (Titsias 2009), jitter included in Kuu."""
    rng = np.random.default_rng(N * 7 + M)
    X = rng.standard_normal((N, D)); y = rng.standard_normal((N, 1)); Z = X[:M] + 0.05 * rng.standard_normal((M, D))
    spec = o.make_spec(D, R, lengthscales=list(rng.uniform(0.6, 1.8, D)), order_variances=list(rng.uniform(0.4, 1.5, R + 1)))
    s2 = 0.07
    Kuf, Kuu, Kd = _mp(o.oak_K(spec, Z, X)), _mp(o.oak_K(spec, Z) + o.JITTER * np.eye(M)), o.oak_K_diag(spec, X)
    Qff = Kuf.T * mp.inverse(Kuu) * Kuf
    S = Qff + mp.mpf(s2) * mp.eye(N)
    logdet, Sinv_y = _logdet_and_solve(S, _mp(y))
    quad = (_mp(y).T * Sinv_y)[0, 0]
    logN = -mp.mpf(0.5) * (N * mp.log(2 * mp.pi) + logdet + quad)
    trace = mp.mpf(float(np.sum(Kd))) - sum(Qff[i, i] for i in range(N))
    bound = float(logN - trace / (2 * mp.mpf(s2)))
    np.testing.assert_allclose(o.sgpr_elbo(spec, X, y, Z, s2), bound, rtol=1e-10)


def test_sgpr_posterior_is_the_titsias_posterior():
    """predict_f == mean Kxu Sigma Kuf y / s2, var Kxx - Kxu Kuu^-1 Kux + Kxu Sigma Kux, Sigma = (Kuu + Kuf Kfu / s2)^-1."""
    rng = np.random.default_rng(5)
    N, M, D = 14, 5, 2
    X = rng.standard_normal((N, D)); y = rng.standard_normal((N, 1)); Z = X[:M] + 0.1
    Xs = rng.standard_normal((6, D))
    spec = o.make_spec(D, 2, lengthscales=[0.9, 1.4], order_variances=[0.5, 1.2, 0.8])
    s2 = 0.1
    Kuf, Kuu = _mp(o.oak_K(spec, Z, X)), _mp(o.oak_K(spec, Z) + o.JITTER * np.eye(M))
    Kus, Kss = _mp(o.oak_K(spec, Z, Xs)), o.oak_K_diag(spec, Xs)
    Sigma = mp.inverse(Kuu + Kuf * Kuf.T / mp.mpf(s2))
    mean = Kus.T * Sigma * Kuf * _mp(y) / mp.mpf(s2)
    Kuu_inv = mp.inverse(Kuu)
    m_or, v_or = o.sgpr_predict_f(spec, X, y, Z, s2, Xs)
    for i in range(6):
        ks = Kus[:, i]
        var = mp.mpf(float(Kss[i])) - (ks.T * Kuu_inv * ks)[0, 0] + (ks.T * Sigma * ks)[0, 0]
        np.testing.assert_allclose(m_or[i, 0], float(mean[i, 0]), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(v_or[i, 0], float(var), rtol=1e-8, atol=1e-12)
    # alpha gives the same mean (oak/utils.py:197-198,528)
    alpha = o.sgpr_alpha(spec, X, y, Z, s2)
    np.testing.assert_allclose(o.oak_K(spec, Xs, Z) @ alpha, m_or, rtol=1e-9, atol=1e-12)


def test_gpr_is_the_dense_gaussian():
    rng = np.random.default_rng(9)
    N, D = 11, 3
    X = rng.standard_normal((N, D)); y = rng.standard_normal((N, 1)); Xs = rng.standard_normal((4, D))
    spec = o.make_spec(D, 2, lengthscales=[0.8, 1.1, 1.7])
    s2 = 0.05
    K = _mp(o.oak_K(spec, X)) + mp.mpf(s2) * mp.eye(N)
    logdet, Kinv_y = _logdet_and_solve(K, _mp(y))
    logp = float(-mp.mpf(0.5) * ((_mp(y).T * Kinv_y)[0, 0] + logdet + N * mp.log(2 * mp.pi)))
    np.testing.assert_allclose(o.gpr_log_marginal_likelihood(spec, X, y, s2), logp, rtol=1e-11)
    Ksx = _mp(o.oak_K(spec, Xs, X))
    m_or, v_or = o.gpr_predict_f(spec, X, y, s2, Xs)
    Kinv = mp.inverse(K)
    for i in range(4):
        np.testing.assert_allclose(m_or[i, 0], float((Ksx[i, :] * Kinv_y)[0, 0]), rtol=1e-9)
        np.testing.assert_allclose(v_or[i, 0], float(o.oak_K_diag(spec, Xs)[i] - (Ksx[i, :] * Kinv * Ksx[i, :].T)[0, 0]), rtol=1e-8)


def test_sgpr_with_all_points_inducing_recovers_gpr():
    """Z = X: Qff -> Kff as jitter -> 0, so the bound tends to the exact log-marginal."""
    rng = np.random.default_rng(2)
    N, D = 25, 2
    X = rng.standard_normal((N, D)); y = np.sin(X[:, :1]) + 0.1 * rng.standard_normal((N, 1))
    spec = o.make_spec(D, 2, lengthscales=[0.5, 0.6])
    gpr = o.gpr_log_marginal_likelihood(spec, X, y, 0.1)
    sgpr = o.sgpr_elbo(spec, X, y, X.copy(), 0.1, jitter=1e-10)
    assert sgpr <= gpr + 1e-8
    np.testing.assert_allclose(sgpr, gpr, rtol=1e-6)


def test_gram_entries_against_mpmath():
    """Elementwise Gram of a Gaussian-measure order-3 kernel against 50-digit arithmetic (<= 1e-13 relative)."""
    rng = np.random.default_rng(11)
    D, R = 4, 3
    ls = rng.uniform(0.5, 2.0, D); ov = rng.uniform(0.4, 1.6, R + 1)
    X = rng.standard_normal((5, D)); Z = rng.standard_normal((4, D))
    spec = o.make_spec(D, R, lengthscales=list(ls), order_variances=list(ov))
    K = o.oak_K(spec, X, Z)
    import itertools
    for i in range(5):
        for j in range(4):
            ks = []
            for d in range(D):
                l = mp.mpf(float(ls[d])); x = mp.mpf(float(X[i, d])); z = mp.mpf(float(Z[j, d]))
                c = lambda t: l / mp.sqrt(l ** 2 + 1) * mp.exp(-t ** 2 / (2 * (l ** 2 + 1)))
                v = l / mp.sqrt(l ** 2 + 2)
                ks.append(mp.exp(-(x - z) ** 2 / (2 * l ** 2)) - c(x) * c(z) / v)
            tot = mp.mpf(float(ov[0]))
            for r in range(1, R + 1):
                tot += mp.mpf(float(ov[r])) * sum(mp.fprod(ks[d] for d in S) for S in itertools.combinations(range(D), r))
            np.testing.assert_allclose(K[i, j], float(tot), rtol=1e-13, atol=1e-15)

"""Definitional pins of oracle/svgp_oracle.py (the reference's SVGP arithmetic lives in gpflow 2.2.1, absent here: the
restatement is checked against the definitions it implements, not against reference outputs)."""
import numpy as np
import pytest
from scipy import integrate

import cases  # noqa: F401  (sys.path)
from oracle import oak_oracle as o, svgp_oracle as sv


def _problem(seed=3, N=40, D=3, M=9):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    Z = X[:M].copy()
    y = (rng.uniform(size=N) < 0.5).astype(float)
    spec = o.make_spec(D, 2, lengthscales=list(rng.uniform(0.7, 1.6, D)), order_variances=[0.7, 1.2, 0.5])
    q_mu = 0.5 * rng.standard_normal(M)
    q_sqrt = rng.uniform(0.3, 0.9, M)
    return spec, X, y, Z, q_mu, q_sqrt


@pytest.mark.parametrize("link", ["logit", "probit"])
def test_variational_expectations_match_adaptive_quadrature(link):
    rng = np.random.default_rng(0)
    mu, var = rng.standard_normal(12), rng.uniform(0.05, 2.0, 12)
    y = (rng.uniform(size=12) < 0.5).astype(float)
    got = sv.variational_expectations(mu, var, y, link)
    for i in range(12):
        f = lambda t: sv.bernoulli_log_prob(np.array([t]), np.array([y[i]]), link)[0] * np.exp(-0.5 * (t - mu[i]) ** 2 / var[i]) / np.sqrt(2 * np.pi * var[i])
        ref, _ = integrate.quad(f, mu[i] - 12 * np.sqrt(var[i]), mu[i] + 12 * np.sqrt(var[i]), epsabs=1e-13, epsrel=1e-13)
        # truncation error of the 20-node rule itself: the jittered probit's log has a sharper knee than the logistic's
        assert abs(got[i] - ref) <= (2e-6 if link == "logit" else 5e-4) * max(1.0, abs(ref))


@pytest.mark.parametrize("link", ["logit", "probit"])
def test_predict_log_density_matches_adaptive_quadrature(link):
    rng = np.random.default_rng(1)
    mu, var = rng.standard_normal(8), rng.uniform(0.05, 2.0, 8)
    y = (rng.uniform(size=8) < 0.5).astype(float)
    got = sv.predict_log_density_from_f(mu, var, y, link)
    for i in range(8):
        f = lambda t: np.exp(sv.bernoulli_log_prob(np.array([t]), np.array([y[i]]), link)[0]) * np.exp(-0.5 * (t - mu[i]) ** 2 / var[i]) / np.sqrt(2 * np.pi * var[i])
        ref, _ = integrate.quad(f, mu[i] - 12 * np.sqrt(var[i]), mu[i] + 12 * np.sqrt(var[i]), epsabs=1e-13, epsrel=1e-13)
        assert abs(got[i] - np.log(ref)) <= 1e-6


def test_prior_kl_is_the_dense_gaussian_kl():
    rng = np.random.default_rng(2)
    m, s = rng.standard_normal(7), rng.uniform(0.2, 1.5, 7)
    S = np.diag(s ** 2)
    dense = 0.5 * (np.trace(S) + m @ m - 7 - np.linalg.slogdet(S)[1])
    assert abs(sv.prior_kl(m, s) - dense) <= 1e-13 * abs(dense)


def test_whitened_conditional_is_the_dense_posterior_marginal():
    spec, X, _, Z, q_mu, q_sqrt = _problem()
    M = Z.shape[0]
    Kmm = o.oak_K(spec, Z) + sv.JITTER * np.eye(M)
    Lm = np.linalg.cholesky(Kmm)
    mu_u, S_u = Lm @ q_mu, Lm @ np.diag(q_sqrt ** 2) @ Lm.T            # q(u) in the non-whitened parametrisation
    Kmn = o.oak_K(spec, Z, X)
    Kinv = np.linalg.inv(Kmm)
    mean = Kmn.T @ Kinv @ mu_u
    var = o.oak_K_diag(spec, X) - np.einsum("mn,mk,kn->n", Kmn, Kinv, Kmn) + np.einsum("mn,mk,kn->n", Kmn, Kinv @ S_u @ Kinv, Kmn)
    fm, fv = sv.conditional(spec, X, Z, q_mu, q_sqrt)
    np.testing.assert_allclose(fm, mean, rtol=0, atol=1e-9)
    np.testing.assert_allclose(fv, var, rtol=0, atol=1e-9)


def test_posterior_pieces():
    spec, X, _, Z, q_mu, q_sqrt = _problem()
    alpha, L = sv.svgp_posterior(spec, Z, q_mu, q_sqrt)
    fm, _ = sv.conditional(spec, X, Z, q_mu, q_sqrt)
    np.testing.assert_allclose(o.oak_K(spec, X, Z) @ alpha, fm, rtol=0, atol=1e-9)     # mean = K(X, Z) alpha
    M = Z.shape[0]
    Lm = np.linalg.cholesky(o.oak_K(spec, Z) + sv.JITTER * np.eye(M))
    np.testing.assert_allclose(L, Lm / np.sqrt(1 - q_sqrt ** 2)[None, :], rtol=1e-6, atol=1e-9)   # closed form the HIP path uses


def test_elbo_is_finite_and_increases_along_its_gradient():
    """The bound is finite, negative, and improves when q_mu moves along its gradient (finite differences)."""
    spec, X, y, Z, q_mu, q_sqrt = _problem()
    e0 = sv.svgp_elbo(spec, X, y, Z, q_mu, q_sqrt)
    assert np.isfinite(e0) and e0 < 0
    g = np.zeros_like(q_mu)
    for j in range(q_mu.size):
        d = np.zeros_like(q_mu); d[j] = 1e-6
        g[j] = (sv.svgp_elbo(spec, X, y, Z, q_mu + d, q_sqrt) - sv.svgp_elbo(spec, X, y, Z, q_mu - d, q_sqrt)) / 2e-6
    assert sv.svgp_elbo(spec, X, y, Z, q_mu + 1e-3 * g, q_sqrt) > e0

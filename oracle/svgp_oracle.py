"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the SVGP model the reference's classification example puts on the OAK
kernel.  Nothing under ``orthogonal-additive-gaussian-processes_amd/`` may import this module.

Reference call sites: examples/uci/uci_classification_train.py:43-45 (``inv_logit``), :108-116
(``gpflow.models.SVGP(kernel, Bernoulli(invlink=inv_logit), Z, whiten=True, q_diag=True)``), :119-124 (full-batch BFGS),
:128-137 (``predict_f`` / ``predict_log_density``); oak/utils.py:174-179 (``posterior.alpha``, ``posterior.Qinv``).

The arithmetic itself lives in the third-party dependency gpflow==2.2.1 (setup.py:13), which is not installable here
(SURVEY.md section 8c): PARITY UNPINNED against the reference for this row.  The functions below restate GPflow's published
definitions op by op --
  * ``SVGP.elbo``: sum of ``likelihood.variational_expectations(f_mean, f_var, Y)`` minus ``prior_kl``;
  * ``conditionals.base_conditional`` with ``white=True`` and a diagonal ``q_sqrt``;
  * ``kullback_leiblers.gauss_kl`` with ``K=None`` and a diagonal ``q_sqrt``;
  * ``quadrature.NDiagGHQuadrature(1, 20)``: nodes ``sqrt(2) x_i``, weights ``w_i / sqrt(pi)`` from numpy's ``hermgauss``;
  * ``likelihoods.Bernoulli``: ``log(where(y == 1, p, 1 - p))`` with ``p = invlink(f)``;
  * ``posteriors``: whitened ``alpha = Lm^-T q_mu`` and ``Qinv = Lm^-T (I - q_sqrt q_sqrt^T) Lm^-1``
-- and tests/test_oracle_svgp.py pins them definitionally: the variational expectations against adaptive quadrature
(scipy.integrate.quad), the KL against the dense Gaussian KL formula, and the conditional against the dense non-whitened
posterior q(u) = N(Lm q_mu, Lm S Lm^T).  Gradients of the HIP path are checked against finite differences of this file.
"""
import numpy as np
from scipy.special import erf

from . import oak_oracle as o

JITTER = 1e-6
LINK_EPS = 1e-3
NUM_GH = 20


def inv_logit(x, eps=LINK_EPS):
    """examples/uci/uci_classification_train.py:43-45: tf.math.sigmoid(x) * (1 - 2 jitter) + jitter."""
    return 1.0 / (1.0 + np.exp(-x)) * (1 - 2 * eps) + eps


def inv_probit(x, eps=LINK_EPS):
    """gpflow.likelihoods.utils.inv_probit: 0.5 (1 + erf(x / sqrt 2)) (1 - 2 jitter) + jitter."""
    return 0.5 * (1.0 + erf(x / np.sqrt(2.0))) * (1 - 2 * eps) + eps


LINKS = {"logit": inv_logit, "probit": inv_probit}


def bernoulli_log_prob(F, Y, link="logit", eps=LINK_EPS):
    """gpflow.logdensities.bernoulli(Y, invlink(F))."""
    p = LINKS[link](F, eps)
    return np.log(np.where(np.equal(Y, 1), p, 1 - p))


def gh_rule(n=NUM_GH):
    x, w = np.polynomial.hermite.hermgauss(n)
    return x * np.sqrt(2.0), w / np.sqrt(np.pi)


def variational_expectations(Fmu, Fvar, Y, link="logit", eps=LINK_EPS, n_gh=NUM_GH):
    z, w = gh_rule(n_gh)
    X = Fmu[:, None] + np.sqrt(Fvar)[:, None] * z[None, :]
    return (bernoulli_log_prob(X, Y[:, None], link, eps) * w[None, :]).sum(1)


def predict_log_density_from_f(Fmu, Fvar, Y, link="logit", eps=LINK_EPS, n_gh=NUM_GH):
    """ScalarLikelihood._predict_log_density: logsumexp_i(log p(y | f_i) + log w_i)."""
    z, w = gh_rule(n_gh)
    X = Fmu[:, None] + np.sqrt(Fvar)[:, None] * z[None, :]
    t = bernoulli_log_prob(X, Y[:, None], link, eps) + np.log(w)[None, :]
    mx = t.max(1)
    return mx + np.log(np.exp(t - mx[:, None]).sum(1))


def conditional(spec, Xnew, Z, q_mu, q_sqrt, jitter=JITTER):
    """base_conditional(Kmn, Kmm, Knn, f=q_mu, q_sqrt=diag, white=True), full_cov=False."""
    M = Z.shape[0]
    Kmm = o.oak_K(spec, Z) + jitter * np.eye(M)
    Kmn = o.oak_K(spec, Z, Xnew)
    Knn = o.oak_K_diag(spec, Xnew)
    Lm = np.linalg.cholesky(Kmm)
    A = np.linalg.solve(Lm, Kmn)                 # triangular_solve(Lm, Kmn, lower=True)
    fvar = Knn - (A * A).sum(0)
    fmean = A.T @ q_mu
    LTA = A * q_sqrt[:, None]
    fvar = fvar + (LTA * LTA).sum(0)
    return fmean, fvar


def prior_kl(q_mu, q_sqrt):
    """gauss_kl(q_mu, q_sqrt, K=None), diagonal q_sqrt, one latent."""
    M = q_mu.size
    mahalanobis = (q_mu ** 2).sum()
    constant = -M
    logdet_qcov = np.log(q_sqrt ** 2).sum()
    trace = (q_sqrt ** 2).sum()
    return 0.5 * (mahalanobis + constant - logdet_qcov + trace)


def svgp_elbo(spec, X, Y, Z, q_mu, q_sqrt, link="logit", eps=LINK_EPS, jitter=JITTER, n_gh=NUM_GH):
    fmean, fvar = conditional(spec, X, Z, q_mu, q_sqrt, jitter)
    ve = variational_expectations(fmean, fvar, np.asarray(Y, dtype=np.float64).reshape(-1), link, eps, n_gh)
    return ve.sum() - prior_kl(q_mu, q_sqrt)


def svgp_predict_log_density(spec, Xnew, Ynew, Z, q_mu, q_sqrt, link="logit", eps=LINK_EPS, jitter=JITTER, n_gh=NUM_GH):
    fmean, fvar = conditional(spec, Xnew, Z, q_mu, q_sqrt, jitter)
    return predict_log_density_from_f(fmean, fvar, np.asarray(Ynew, dtype=np.float64).reshape(-1), link, eps, n_gh)


def svgp_posterior(spec, Z, q_mu, q_sqrt, jitter=JITTER):
    """(alpha, L): posterior.alpha and chol(inv(posterior.Qinv[0])) as oak/utils.py:174-179 computes them."""
    M = Z.shape[0]
    Lm = np.linalg.cholesky(o.oak_K(spec, Z) + jitter * np.eye(M))
    Linv = np.linalg.solve(Lm, np.eye(M))
    alpha = Linv.T @ q_mu
    Qinv = Linv.T @ (np.eye(M) - np.diag(q_sqrt ** 2)) @ Linv
    L = np.linalg.cholesky(np.linalg.inv(Qinv))
    return alpha, L

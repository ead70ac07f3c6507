"""GPU parity of the SVGP (whitened, diagonal q, Bernoulli) path through the C ABI against oracle/svgp_oracle.py.
Tolerances (fp64): ELBO <= 1e-10 relative, predictive mean / variance / log density <= 1e-9, gradients <= 1e-6 relative
to the gradient norm against central differences of the oracle (difference step 1e-5, truncation ~1e-9)."""
import numpy as np
import pytest

import cases
from oak import _capi
from oracle import oak_oracle as o, svgp_oracle as sv

pytestmark = pytest.mark.gpu


def problem(seed, N, D, M, R, kinds=("gaussian",)):
    rng = np.random.default_rng(seed)
    spec = cases.random_spec(rng, D, R, kinds=kinds)
    X = cases.random_inputs(rng, spec, N)
    Z = X[rng.choice(N, M, replace=False)].copy()
    f = np.sin(X[:, 0]) + 0.5 * X[:, 1 % D]
    y = (rng.uniform(size=N) < 1.0 / (1.0 + np.exp(-2 * f))).astype(float)
    q_mu = 0.7 * rng.standard_normal(M)
    q_sqrt = rng.uniform(0.2, 1.3, M)
    return spec, X, y, Z, q_mu, q_sqrt


@pytest.mark.parametrize("link", ["logit", "probit"])
@pytest.mark.parametrize("N,D,M,R,kinds", [(700, 8, 200, 2, ("gaussian",)), (333, 5, 65, 4, ("gaussian", "binary", "categorical")),
                                           (2049, 3, 129, 3, ("gaussian", "uniform", "mog")), (50, 2, 50, 1, ("gaussian",))])
def test_elbo_and_predictions_match_oracle(hip, link, N, D, M, R, kinds):
    spec, X, y, Z, q_mu, q_sqrt = problem(N + M, N, D, M, R, kinds)
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y.reshape(-1, 1))
    hip.sgpr_set_inducing(Z)
    e = hip.svgp_elbo(d, q_mu, q_sqrt, link=link)
    er = sv.svgp_elbo(spec, X, y, Z, q_mu, q_sqrt, link=link)
    assert abs(e - er) <= 1e-10 * abs(er), (e, er)
    rng = np.random.default_rng(1)
    Xs = cases.random_inputs(rng, spec, 301)
    ys = (rng.uniform(size=301) < 0.5).astype(float)
    m, v, ld = hip.svgp_predict(d, q_mu, q_sqrt, Xs, ys, link=link)
    mr, vr = sv.conditional(spec, Xs, Z, q_mu, q_sqrt)
    ldr = sv.svgp_predict_log_density(spec, Xs, ys, Z, q_mu, q_sqrt, link=link)
    np.testing.assert_allclose(m, mr, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(v, vr, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ld, ldr, rtol=1e-9, atol=1e-9)
    m2, v2 = hip.svgp_predict(d, q_mu, q_sqrt, Xs)
    assert np.array_equal(m, m2) and np.array_equal(v, v2)


def test_posterior_alpha_and_L(hip):
    spec, X, y, Z, q_mu, q_sqrt = problem(5, 400, 4, 60, 2)
    q_sqrt = np.minimum(q_sqrt, 0.95)
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y.reshape(-1, 1))
    hip.sgpr_set_inducing(Z)
    alpha, L = hip.svgp_posterior(d, q_mu, q_sqrt)
    ar, Lr = sv.svgp_posterior(spec, Z, q_mu, q_sqrt)
    np.testing.assert_allclose(alpha, ar, rtol=1e-7, atol=1e-7 * np.abs(ar).max())      # cond(Kuu) ~ 1e6 amplifies rounding in both
    np.testing.assert_allclose(L, Lr, rtol=1e-6, atol=1e-8)                              # the oracle inverts Qinv explicitly
    with pytest.raises(_capi.NotPositiveDefiniteError):
        hip.svgp_posterior(d, q_mu, np.full_like(q_sqrt, 1.0))
    assert hip.svgp_posterior(d, q_mu, np.full_like(q_sqrt, 1.0), get_L=False).shape == (60,)


def _unpack(spec, theta, M):
    D = len(spec["dims"])
    return theta[:M], theta[M:2 * M], theta[2 * M:2 * M + D], theta[2 * M + D:]


@pytest.mark.parametrize("link", ["logit", "probit"])
@pytest.mark.parametrize("N,D,M,R,kinds", [(257, 4, 33, 2, ("gaussian",)), (190, 5, 40, 3, ("gaussian", "binary", "uniform"))])
def test_gradient_matches_central_differences_of_the_oracle(hip, link, N, D, M, R, kinds):
    import copy
    spec, X, y, Z, q_mu, q_sqrt = problem(N, N, D, M, R, kinds)
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y.reshape(-1, 1))
    hip.sgpr_set_inducing(Z)
    e, g, gm, gs = hip.svgp_elbo(d, q_mu, q_sqrt, link=link, grad=True)
    assert abs(e - sv.svgp_elbo(spec, X, y, Z, q_mu, q_sqrt, link=link)) <= 1e-10 * abs(e)
    assert e == hip.svgp_elbo(d, q_mu, q_sqrt, link=link)

    def f(qm, qs, ls=None, ov=None):
        sp = copy.deepcopy(spec)
        if ls is not None:
            for k, dim in enumerate(sp["dims"]):
                if dim["type"] == "rbf":
                    dim["lengthscale"] = float(ls[k])
        if ov is not None:
            sp["order_variances"] = list(ov)
        return sv.svgp_elbo(sp, X, y, Z, qm, qs, link=link)

    h = 1e-5
    fd_m = np.array([(f(q_mu + h * np.eye(M)[j], q_sqrt) - f(q_mu - h * np.eye(M)[j], q_sqrt)) / (2 * h) for j in range(M)])
    fd_s = np.array([(f(q_mu, q_sqrt + h * np.eye(M)[j]) - f(q_mu, q_sqrt - h * np.eye(M)[j])) / (2 * h) for j in range(M)])
    assert np.abs(gm - fd_m).max() <= 1e-6 * np.abs(fd_m).max()
    assert np.abs(gs - fd_s).max() <= 1e-6 * np.abs(fd_s).max()
    ls0 = np.array([dim.get("lengthscale", 1.0) for dim in spec["dims"]])
    ov0 = np.array(spec["order_variances"])
    fd_l = np.array([(f(q_mu, q_sqrt, ls=ls0 + h * np.eye(D)[k]) - f(q_mu, q_sqrt, ls=ls0 - h * np.eye(D)[k])) / (2 * h) for k in range(D)])
    fd_o = np.array([(f(q_mu, q_sqrt, ov=ov0 + h * np.eye(R + 1)[k]) - f(q_mu, q_sqrt, ov=ov0 - h * np.eye(R + 1)[k])) / (2 * h) for k in range(R + 1)])
    rbf = np.array([dim["type"] == "rbf" for dim in spec["dims"]])
    assert np.abs(g[:D][rbf] - fd_l[rbf]).max() <= 1e-6 * np.abs(fd_l).max()
    assert np.abs(g[2 * D:2 * D + R + 1] - fd_o).max() <= 1e-6 * np.abs(fd_o).max()
    assert g[2 * D + R + 1] == 0.0                                   # noise slot: no Gaussian likelihood here


def test_argument_checks(hip):
    spec, X, y, Z, q_mu, q_sqrt = problem(9, 100, 3, 20, 2)
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y.reshape(-1, 1))
    hip.sgpr_set_inducing(Z)
    with pytest.raises(ValueError):
        hip.svgp_elbo(d, q_mu, -q_sqrt)
    with pytest.raises(ValueError):
        hip.svgp_elbo(d, q_mu, q_sqrt, n_gh=65)
    with pytest.raises(_capi.NotPositiveDefiniteError):
        hip.svgp_elbo(d, q_mu, q_sqrt, jitter=-10.0)

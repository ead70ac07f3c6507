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


@pytest.mark.parametrize("N,M", [(9000, 300), (8192, 384), (8300, 301)])     # ragged, whole blocks, odd (two-step form)
def test_large_batch_gradient_directional(hip, N, M):
    """N >= 8192 rows take the blocked (by-inverse, MFMA) forms of both triangular solves and the signed weighted SYRK;
    checked by a central difference of the oracle along one random direction in (q_mu, q_sqrt, lengthscales, variances)."""
    import copy
    D, R = 4, 2
    spec, X, y, Z, q_mu, q_sqrt = problem(N, N, D, M, R)
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y.reshape(-1, 1))
    hip.sgpr_set_inducing(Z)
    e, g, gm, gs = hip.svgp_elbo(d, q_mu, q_sqrt, grad=True)
    assert abs(e - sv.svgp_elbo(spec, X, y, Z, q_mu, q_sqrt)) <= 1e-10 * abs(e)
    rng = np.random.default_rng(0)
    dm, ds = rng.standard_normal(M) / np.sqrt(M), 0.1 * rng.standard_normal(M) / np.sqrt(M)
    dl, dv = 0.1 * rng.standard_normal(D), 0.1 * rng.standard_normal(R + 1)
    ls0 = np.array([dim["lengthscale"] for dim in spec["dims"]]); ov0 = np.array(spec["order_variances"])

    def f(t):
        sp = copy.deepcopy(spec)
        for k, dim in enumerate(sp["dims"]):
            dim["lengthscale"] = float(ls0[k] + t * dl[k])
        sp["order_variances"] = list(ov0 + t * dv)
        return sv.svgp_elbo(sp, X, y, Z, q_mu + t * dm, q_sqrt + t * ds)

    h = 2e-4           # five-point stencil: the oracle's own rounding (~1e-10 |elbo| through cond(Kuu)) is divided by 12 h
    fd = (8 * (f(h) - f(-h)) - (f(2 * h) - f(-2 * h))) / (12 * h)
    an = gm @ dm + gs @ ds + g[:D] @ dl + g[2 * D:2 * D + R + 1] @ dv
    assert abs(an - fd) <= 1e-6 * abs(fd), (an, fd)


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


# ---- model API: the flow of the reference's classification example -----------------------------------------------------
def _classification_data(N, D, seed=4):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, D))
    f = 2.0 * np.sin(X[:, 0]) + 1.5 * X[:, 1] * X[:, 2]
    y = (rng.uniform(size=N) < 1.0 / (1.0 + np.exp(-3 * f))).astype(float)[:, None]
    return X, y, f


def test_classification_example_flow(hip):
    """examples/uci/uci_classification_train.py:99-183 on synthetic data: oak_model.fit(optimise=False) builds the kernel,
    an SVGP with the same kernel is trained by BFGS on the analytic gradient, then predict_f / predict_log_density /
    Sobol / per-term predictions."""
    from oak import gpflow_lite as gpflow
    from oak.gpflow_lite import inv_logit, set_trainable
    from oak.model_utils import oak_model
    from oak.utils import get_model_sufficient_statistics, get_prediction_component
    X, y, _ = _classification_data(900, 4)
    Xtr, ytr, Xte, yte = X[:700], y[:700], X[700:], y[700:]
    oak = oak_model(max_interaction_depth=2, num_inducing=50, use_normalising_flow=False)
    oak.fit(Xtr, ytr, optimise=False)
    data = (np.asarray(oak.m.data[0]), ytr)
    Z = data[0][:50].copy()
    m = gpflow.models.SVGP(kernel=oak.m.kernel, likelihood=gpflow.likelihoods.Bernoulli(invlink=inv_logit), inducing_variable=Z,
                           whiten=True, q_diag=True)
    oak.m = m
    set_trainable(m.inducing_variable, False)
    assert all(v is not m.inducing_variable.Z for v in m.trainable_variables)
    # initial point: q_mu = 0, q_sqrt = 1  =>  KL = 0, f ~ prior
    assert m.prior_kl() == 0.0
    e0 = m.elbo(data)
    spec = m._spec()
    assert abs(e0 - sv.svgp_elbo(spec, data[0], ytr, Z, np.zeros(50), np.ones(50))) <= 1e-10 * abs(e0)
    res = gpflow.optimizers.Scipy().minimize(m.training_loss_closure(data), m.trainable_variables, method="BFGS",
                                             options={"maxiter": 60})
    assert np.isfinite(res.fun) and m.elbo(data) > e0 + 50.0
    # parity of the trained model against the oracle, and the loss the optimiser saw
    qm, qs = m.q_mu.numpy().reshape(-1), m.q_sqrt.numpy().reshape(-1)
    spec = m._spec()
    er = sv.svgp_elbo(spec, data[0], ytr, Z, qm, qs)
    assert abs(m.elbo(data) - er) <= 1e-10 * abs(er)
    assert abs(m.training_loss(data) + er + m.log_prior_density()) <= 1e-9 * abs(er)
    XT = oak._transform_x(Xte)
    mu, var = m.predict_f(XT)
    assert mu.shape == (200, 1) and var.shape == (200, 1) and np.all(np.asarray(var) > 0)
    prob = np.asarray(inv_logit(mu))
    err = np.mean((prob > 0.5).astype(float) != yte)
    assert err < 0.25                                           # labels are noisy draws; chance is 0.5
    nll = -np.asarray(m.predict_log_density((XT, yte))).mean()
    assert np.isfinite(nll) and nll < np.log(2.0)
    np.testing.assert_allclose(np.asarray(m.predict_log_density((XT, yte))), sv.svgp_predict_log_density(spec, XT, yte, Z, qm, qs),
                               rtol=1e-9, atol=1e-9)
    # Sobol and the per-term decomposition of the latent mean (uci_classification_train.py:149-178)
    oak.m.data = data
    oak.get_sobol()
    top = {tuple(int(i) for i in oak.tuple_of_indices[j]) for j in np.argsort(oak.normalised_sobols)[::-1][:2]}
    assert top == {(0,), (1, 2)}
    alpha = get_model_sufficient_statistics(m, get_L=False)
    parts = get_prediction_component(m, alpha, XT)
    const = np.asarray(alpha).sum() * m.kernel.variances[0].numpy()
    np.testing.assert_allclose(const + np.sum([np.asarray(p) for p in parts], axis=0), np.asarray(mu)[:, 0], rtol=1e-7, atol=1e-7)
    # posterior object surface read by oak/utils.py:174-179
    post = m.posterior()
    ar, _ = sv.svgp_posterior(spec, Z, qm, np.minimum(qs, 0.99))
    np.testing.assert_allclose(np.asarray(post.alpha)[:, 0], ar, rtol=1e-6, atol=1e-6 * np.abs(ar).max())
    if np.all(qs < 1.0):
        a2, L = get_model_sufficient_statistics(m)
        assert np.asarray(L).shape == (50, 50) and np.allclose(np.triu(np.asarray(L), 1), 0.0)


def test_svgp_rejects_configurations_outside_the_reference_use(hip):
    from oak import gpflow_lite as gpflow
    from oak.gpflow_lite import inv_logit
    k = gpflow.kernels.RBF()
    Z = np.zeros((3, 1))
    lik = gpflow.likelihoods.Bernoulli(invlink=inv_logit)
    with pytest.raises(NotImplementedError):
        gpflow.models.SVGP(k, lik, Z, whiten=False, q_diag=True)
    with pytest.raises(NotImplementedError):
        gpflow.models.SVGP(k, lik, Z, whiten=True, q_diag=False)
    with pytest.raises(NotImplementedError):
        gpflow.models.SVGP(k, gpflow.likelihoods.Gaussian(), Z, whiten=True, q_diag=True)
    with pytest.raises(NotImplementedError):
        gpflow.likelihoods.Bernoulli(invlink=lambda x: x)


def test_svgp_save_and_load_all_parameters(hip, tmp_path):
    """save_model stores every parameter of an SVGP model (oak/model_utils.py:53-56); load_model(load_all_parameters=True)
    restores them positionally."""
    from oak import gpflow_lite as gpflow
    from oak.gpflow_lite import inv_probit
    from oak.model_utils import load_model, save_model
    from oak.oak_kernel import OAKKernel
    rng = np.random.default_rng(0)
    X = rng.normal(size=(120, 3))
    y = (rng.uniform(size=(120, 1)) < 0.5).astype(float)

    def build():
        k = OAKKernel([gpflow.kernels.RBF] * 3, num_dims=3, max_interaction_depth=2, constrain_orthogonal=True)
        return gpflow.models.SVGP(k, gpflow.likelihoods.Bernoulli(invlink=inv_probit), X[:10].copy(), whiten=True, q_diag=True)

    a = build()
    a.q_mu.assign(rng.normal(size=(10, 1)))
    a.q_sqrt.assign(rng.uniform(0.3, 0.8, size=(10, 1)))
    a.kernel.variances[1].assign(0.37)
    save_model(a, tmp_path / "m" / "svgp.npz")
    b = build()
    load_model(b, tmp_path / "m" / "svgp.npz", load_all_parameters=True)
    for pa, pb in zip(a.parameters, b.parameters):
        np.testing.assert_array_equal(np.asarray(pa.numpy()), np.asarray(pb.numpy()))
    assert a.elbo((X, y)) == b.elbo((X, y))


@pytest.mark.parametrize("world", [2, 8])
def test_loopback_ranks_equal_stacked_rows(world):
    """Row-sharded SVGP arithmetic on one GPU: under the loopback communicator (`world` ranks holding the same rows, every
    all-reduce multiplies by `world`) ELBO and every gradient block equal a single-rank run on the rows stacked `world`
    times -- all-reduce placement of sum(ve), the sign flag, u, the weighted SYRKs, the gradient record, and the 1/nranks
    scaling of the replicated Kuu term."""
    spec, X, y, Z, q_mu, q_sqrt = problem(77, 1200, 4, 48, 2, ("gaussian", "binary"))
    d = _capi.KernelDesc(spec)
    ref = _capi.HipContext(0)
    ref.sgpr_set_data(np.tile(X, (world, 1)), np.tile(y.reshape(-1, 1), (world, 1))); ref.sgpr_set_inducing(Z)
    e_ref, g_ref, gm_ref, gs_ref = ref.svgp_elbo(d, q_mu, q_sqrt, grad=True)
    ctx = _capi.HipContext(0)
    ctx.sgpr_set_data(X, y.reshape(-1, 1)); ctx.sgpr_set_inducing(Z)
    ctx.comm_init_loopback(world)
    e0 = ctx.svgp_elbo(d, q_mu, q_sqrt)
    e, g, gm, gs = ctx.svgp_elbo(d, q_mu, q_sqrt, grad=True)
    assert abs(e0 - e_ref) <= 1e-11 * abs(e_ref) and abs(e - e_ref) <= 1e-11 * abs(e_ref)
    for a, b in ((g, g_ref), (gm, gm_ref), (gs, gs_ref)):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9 * np.abs(b).max())
    ctx.comm_destroy()
    assert abs(ctx.svgp_elbo(d, q_mu, q_sqrt) - e_ref) > 1e-3 * abs(e_ref)
    ctx.close(); ref.close()


@pytest.mark.parametrize("N,D,M,R", [(166, 60, 166, 2), (280, 34, 200, 4)])
def test_uci_classification_shapes(hip, N, D, M, R):
    """The shapes of the reference's classification datasets with the most inputs (sonar: 60 inputs at depth 2, every
    training row an inducing point; ionosphere: 34 inputs at depth 4): more than 32 dimensions take the generic
    backward kernel.  ELBO 1e-10; gradient by a five-point directional difference of the oracle."""
    import copy
    spec, X, y, Z, q_mu, q_sqrt = problem(N + D, N, D, M, R)
    if M == N:
        Z = X.copy()
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y.reshape(-1, 1))
    hip.sgpr_set_inducing(Z)
    e, g, gm, gs = hip.svgp_elbo(d, q_mu, q_sqrt, grad=True)
    er = sv.svgp_elbo(spec, X, y, Z, q_mu, q_sqrt)
    assert abs(e - er) <= 1e-10 * abs(er)
    rng = np.random.default_rng(1)
    dm, ds = rng.standard_normal(M) / np.sqrt(M), 0.1 * rng.standard_normal(M) / np.sqrt(M)
    dl, dv = 0.1 * rng.standard_normal(D) / np.sqrt(D), 0.1 * rng.standard_normal(R + 1)
    ls0 = np.array([dim["lengthscale"] for dim in spec["dims"]]); ov0 = np.array(spec["order_variances"])

    def f(t):
        sp = copy.deepcopy(spec)
        for k, dim in enumerate(sp["dims"]):
            dim["lengthscale"] = float(ls0[k] + t * dl[k])
        sp["order_variances"] = list(ov0 + t * dv)
        return sv.svgp_elbo(sp, X, y, Z, q_mu + t * dm, q_sqrt + t * ds)

    h = 2e-4
    fd = (8 * (f(h) - f(-h)) - (f(2 * h) - f(-2 * h))) / (12 * h)
    an = gm @ dm + gs @ ds + g[:D] @ dl + g[2 * D:2 * D + R + 1] @ dv
    assert abs(an - fd) <= 2e-6 * abs(fd), (an, fd)


def test_gradient_with_separate_base_variances(hip):
    """share_var_across_orders=False: per-dimension base variances are trainable and only the constant term keeps an order
    variance; their gradients come from the same backward contraction (base-variance slots of the record)."""
    import copy
    rng = np.random.default_rng(21)
    N, D, M, R = 300, 4, 40, 3
    spec = cases.random_spec(rng, D, R, kinds=("gaussian", "binary", "gaussian", "categorical"), share=False)
    X = cases.random_inputs(rng, spec, N)
    Z = X[:M].copy()
    y = (rng.uniform(size=N) < 0.5).astype(float)
    q_mu, q_sqrt = 0.5 * rng.standard_normal(M), rng.uniform(0.3, 1.2, M)
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y.reshape(-1, 1)); hip.sgpr_set_inducing(Z)
    e, g, gm, gs = hip.svgp_elbo(d, q_mu, q_sqrt, grad=True)
    assert abs(e - sv.svgp_elbo(spec, X, y, Z, q_mu, q_sqrt)) <= 1e-10 * abs(e)
    h = 1e-4
    for k in range(D):
        def f(t):
            sp = copy.deepcopy(spec); sp["dims"][k]["variance"] += t
            return sv.svgp_elbo(sp, X, y, Z, q_mu, q_sqrt)
        fd = (8 * (f(h) - f(-h)) - (f(2 * h) - f(-2 * h))) / (12 * h)
        assert abs(g[D + k] - fd) <= 1e-6 * max(abs(fd), np.abs(g[D:2 * D]).max()), (k, g[D + k], fd)
    def f0(t):
        sp = copy.deepcopy(spec); sp["order_variances"][0] += t
        return sv.svgp_elbo(sp, X, y, Z, q_mu, q_sqrt)
    fd0 = (8 * (f0(h) - f0(-h)) - (f0(2 * h) - f0(-2 * h))) / (12 * h)
    assert abs(g[2 * D] - fd0) <= 1e-6 * abs(fd0)


def test_gradient_of_trainable_base_variances_under_shared_order_variances(hip):
    """share_var_across_orders=True with MOG / uniform-measure dims whose base variance stays trainable (as the reference's
    OAKKernel leaves it for empirical- and MOG-measure dims): the pair contribution to d/d variance must be present."""
    import copy
    rng = np.random.default_rng(33)
    N, D, M, R = 280, 4, 36, 2
    spec = cases.random_spec(rng, D, R, kinds=("gaussian", "mog", "uniform", "binary"), share=True)
    spec["dims"][1]["variance"] = 0.75
    spec["dims"][2]["variance"] = 1.4
    spec["base_var_grad"] = True
    X = cases.random_inputs(rng, spec, N)
    Z = X[:M].copy()
    y = (rng.uniform(size=N) < 0.5).astype(float)
    q_mu, q_sqrt = 0.5 * rng.standard_normal(M), rng.uniform(0.3, 1.2, M)
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y.reshape(-1, 1)); hip.sgpr_set_inducing(Z)
    e, g, gm, gs = hip.svgp_elbo(d, q_mu, q_sqrt, grad=True)
    assert abs(e - sv.svgp_elbo(spec, X, y, Z, q_mu, q_sqrt)) <= 1e-10 * abs(e)
    h = 1e-4
    for k in (1, 2):
        def f(t):
            sp = copy.deepcopy(spec); sp["dims"][k]["variance"] += t
            return sv.svgp_elbo(sp, X, y, Z, q_mu, q_sqrt)
        fd = (8 * (f(h) - f(-h)) - (f(2 * h) - f(-2 * h))) / (12 * h)
        assert abs(g[D + k] - fd) <= 1e-6 * max(abs(fd), np.abs(g[D:2 * D]).max()), (k, g[D + k], fd)

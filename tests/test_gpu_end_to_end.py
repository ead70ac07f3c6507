"""End-to-end model API on the HIP path: fit (device k-means inducing points) -> BFGS on the analytic gradient ->
predict -> Sobol recovers the generating structure.  Mirrors the flow of /root/reference/oak/model_utils.py:249-524
(oak_model.fit / optimise / predict / get_sobol) at a size the reference would need minutes for."""
import numpy as np
import pytest

from oracle import oak_oracle as o

pytestmark = pytest.mark.gpu


def _problem(N, D, seed=1):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, D))
    f = lambda A: np.sin(A[:, 0]) + 0.5 * A[:, 1] ** 2 + 0.8 * A[:, 2] * A[:, 3]
    y = (f(X) + 0.1 * rng.normal(size=N))[:, None]
    Xt = rng.normal(size=(4000, D))
    return X, y, Xt, f(Xt)


def test_fit_optimise_predict_sobol(hip):
    from oak import gpflow_lite as gpflow
    from oak.model_utils import oak_model
    X, y, Xt, ft = _problem(30000, 6)
    oak = oak_model(max_interaction_depth=2, num_inducing=256, sparse=True, use_normalising_flow=False)
    oak.fit(X, y, optimise=False)
    assert oak.m.inducing_variable.Z.numpy().shape == (256, 6)
    loss0 = oak.m.training_loss()
    res = gpflow.Scipy().minimize(oak.m.training_loss_closure(), oak.m.trainable_variables, method="BFGS",
                                  on_linalg_error="inf", options={"maxiter": 40})
    assert res.fun < loss0 and np.isfinite(res.fun)
    pred = oak.predict(Xt)
    assert np.sqrt(np.mean((pred - ft) ** 2)) < 0.05            # noise sd is 0.1
    oak.get_sobol()
    top = {tuple(int(i) for i in oak.tuple_of_indices[j]) for j in np.argsort(oak.normalised_sobols)[::-1][:3]}
    assert top == {(0,), (1,), (2, 3)}
    s = {tuple(int(i) for i in t): v for t, v in zip(oak.tuple_of_indices, oak.normalised_sobols)}
    # analytic variance shares of sin(x0), x1^2/2, 0.8 x2 x3 under N(0,1): 0.432, 0.5, 0.64
    tot = 0.5 * (1 - np.exp(-2)) + 0.5 + 0.64
    np.testing.assert_allclose([s[(0,)], s[(1,)], s[(2, 3)]], np.array([0.5 * (1 - np.exp(-2)), 0.5, 0.64]) / tot, atol=0.03)


def test_minimize_linalg_error_modes(hip):
    """A failed Cholesky at a trial point aborts the optimisation by default (reference behaviour); 'inf' backtracks."""
    from oak import gpflow_lite as gpflow, _capi
    from oak.model_utils import oak_model
    X, y, _, _ = _problem(3000, 4, seed=3)
    oak = oak_model(max_interaction_depth=2, num_inducing=64, sparse=True, use_normalising_flow=False)
    oak.fit(X, y, optimise=False)

    class Exploding:
        def __init__(self, inner): self.inner, self.calls = inner, 0
        def __call__(self): return self.inner()
        def value_and_grad(self, variables):
            self.calls += 1
            if self.calls == 2:
                raise _capi.NotPositiveDefiniteError("synthetic failure", _capi.OAK_E_NOTPD)
            return self.inner.value_and_grad(variables)

    with pytest.raises(_capi.NotPositiveDefiniteError):
        gpflow.Scipy().minimize(Exploding(oak.m.training_loss_closure()), oak.m.trainable_variables, method="BFGS",
                                options={"maxiter": 3})
    oak.fit(X, y, optimise=False)
    loss0 = oak.m.training_loss()
    res = gpflow.Scipy().minimize(Exploding(oak.m.training_loss_closure()), oak.m.trainable_variables, method="BFGS",
                                  on_linalg_error="inf", options={"maxiter": 5})
    assert np.isfinite(res.fun) and res.fun < loss0
    with pytest.raises(ValueError):
        gpflow.Scipy().minimize(oak.m.training_loss_closure(), oak.m.trainable_variables, on_linalg_error="ignore")


@pytest.mark.parametrize("sparse", [False, True])
def test_full_order_model_as_in_the_regression_example(sparse):
    """examples/uci/uci_regression_train.py:86 builds oak_model(max_interaction_depth=X.shape[1]) -- every interaction order, 13
    on UCI housing.  D = 10 here (depth 10 > the r01 cap of 8; 1023 additive terms): fit without optimisation, one loss +
    gradient evaluation, prediction and the Sobol indices of all terms, the normalised indices against the oracle."""
    from oak.model_utils import oak_model
    from oak.oak_kernel import kernel_to_spec
    rng = np.random.default_rng(5)
    N, D = 300, 10
    X = rng.standard_normal((N, D))
    y = (np.sin(X[:, 0]) + X[:, 1] * X[:, 2] + 0.3 * X[:, 3] * X[:, 4] * X[:, 5] + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    oak = oak_model(max_interaction_depth=D, num_inducing=40, sparse=sparse, use_normalising_flow=False)
    oak.fit(X, y, optimise=False)
    spec = kernel_to_spec(oak.m.kernel)
    assert spec["max_interaction_depth"] == D and len(spec["order_variances"]) == D + 1
    loss, grads = oak.m._training_loss_and_grad(oak.m.trainable_variables)
    assert np.isfinite(loss) and all(np.all(np.isfinite(np.asarray(g))) for g in grads)
    pred = oak.predict(X[:50])
    assert pred.shape == (50,) and np.all(np.isfinite(pred))
    sob = oak.get_sobol()
    assert len(sob) == 2 ** D - 1 and abs(sob.sum() - 1.0) < 1e-9 and (sob >= -1e-12).all()
    # against the oracle's Sobol indices of the same posterior
    Xs = oak.X_scaled
    if sparse:
        Z = oak.m.inducing_variable.Z.numpy()
        alpha = o.sgpr_alpha(spec, Xs, oak.Y_scaled, Z, float(oak.m.likelihood.variance.numpy()))
        _, ref = o.compute_sobol_oak(spec, Z, alpha)
    else:
        alpha = o.gpr_alpha(spec, Xs, oak.Y_scaled, float(oak.m.likelihood.variance.numpy()))
        _, ref = o.compute_sobol_oak(spec, Xs, alpha)
    ref = np.asarray(ref)
    np.testing.assert_allclose(sob, ref / ref.sum(), atol=1e-8)

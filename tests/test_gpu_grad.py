"""GPU: analytic gradients of the SGPR bound / GPR log-marginal (HIP backward pass) against central finite
differences of the CPU ORACLE objective (the reference obtains them by TF autodiff, oak/model_utils.py:168-173),
and the reference's optimisation tests (tests/test_optimisation.py)."""
import copy

import numpy as np
import pytest

import cases
from oak import _capi
from oak import gpflow_lite as gpflow
from oak.input_measures import GaussianMeasure
from oak.model_utils import create_model_oak
from oak.oak_kernel import _categorical_chain
from oak.ortho_rbf_kernel import OrthogonalRBFKernel
from oracle import oak_oracle as o

pytestmark = pytest.mark.gpu


def fd(fun, h=1e-5):
    return (fun(h) - fun(-h)) / (2 * h)


def check(g, ref, rtol=2e-5, atol=1e-6):
    np.testing.assert_allclose(g, ref, rtol=rtol, atol=atol * max(1.0, abs(ref)))


@pytest.mark.parametrize("route", ["phi", "whitened"])
@pytest.mark.parametrize("R", [1, 2, 3])
def test_sgpr_gradient_continuous(hip, route, R):
    rng = np.random.default_rng(R)
    D, N, M = 4, 300, 24
    X, y, Z = o.synthetic_problem(N, D, M, seed=5)
    spec = o.make_spec(D, R, lengthscales=list(rng.uniform(0.8, 1.6, D)), order_variances=list(rng.uniform(0.5, 1.5, R + 1)))
    s2 = 0.07
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route(route)
    e, g = hip.sgpr_elbo_grad(_capi.KernelDesc(spec), s2)
    np.testing.assert_allclose(e, o.sgpr_elbo(spec, X, y, Z, s2), rtol=1e-10 if route == "whitened" else 1e-8)

    def with_ls(d, h):
        s = copy.deepcopy(spec); s["dims"][d]["lengthscale"] += h
        return o.sgpr_elbo(s, X, y, Z, s2)

    def with_ov(r, h):
        s = copy.deepcopy(spec); s["order_variances"][r] += h
        return o.sgpr_elbo(s, X, y, Z, s2)

    for d in range(D):
        check(g[d], fd(lambda h: with_ls(d, h)))
    for r in range(R + 1):
        check(g[2 * D + r], fd(lambda h: with_ov(r, h)))
    check(g[2 * D + R + 1], fd(lambda h: o.sgpr_elbo(spec, X, y, Z, s2 + h), h=1e-6))


@pytest.mark.parametrize("D,R", [(12, 2), (16, 3), (20, 2), (32, 4), (33, 2), (9, 5), (16, 8), (8, 7), (24, 6), (32, 8), (40, 3), (13, 13)])
def test_sgpr_gradient_kernel_variants(hip, D, R):
    """D <= 8 / <= 16 / <= 32 at depth <= 8 take the register-resident fast backward kernel (lane pairs above 16 sub-kernels; depth
    5..8 since r04), wider or deeper kernels the general two-pass kernel: all must agree with finite differences of the oracle
    (a subset of parameters is probed)."""
    rng = np.random.default_rng(D * 10 + R)
    N, M = 260, 20
    X, y, Z = o.synthetic_problem(N, D, M, seed=D)
    spec = o.make_spec(D, R, lengthscales=list(rng.uniform(0.9, 1.8, D)), order_variances=list(rng.uniform(0.3, 0.9, R + 1)))
    s2 = 0.1
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    e, g = hip.sgpr_elbo_grad(_capi.KernelDesc(spec), s2)
    for d in (0, D // 2, D - 1):
        def f(h, d=d):
            s = copy.deepcopy(spec); s["dims"][d]["lengthscale"] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(g[d], fd(f), rtol=5e-5)
    for r in (0, R):
        def f(h, r=r):
            s = copy.deepcopy(spec); s["order_variances"][r] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(g[2 * D + r], fd(f), rtol=5e-5)


@pytest.mark.parametrize("depth", [3, 5, 6])
def test_sgpr_gradient_all_kernel_types(hip, depth):
    """Every sub-kernel type / measure, base variances trainable (share_var_across_orders=False), categorical table; depth 3
    (fast kernel, mixed form), 5 and 6 (its depth-5..8 form with base-variance sums and per-dimension exponent offsets)."""
    spec, X, y, Z, s2 = cases.case_B()
    spec = copy.deepcopy(spec)
    spec["max_interaction_depth"] = depth
    spec["share_var_across_orders"] = False
    spec["order_variances"] = [0.6]
    for i, dim in enumerate(spec["dims"]):
        dim["variance"] = 0.7 + 0.15 * i
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    desc = _capi.KernelDesc(spec)
    e, g = hip.sgpr_elbo_grad(desc, s2)
    np.testing.assert_allclose(e, o.sgpr_elbo(spec, X, y, Z, s2), rtol=1e-10)
    D = 6

    def perturbed(path, h):
        s = copy.deepcopy(spec)
        if path[0] == "ls":
            s["dims"][path[1]]["lengthscale"] += h
        elif path[0] == "bv":
            s["dims"][path[1]]["variance"] += h
        elif path[0] == "ov":
            s["order_variances"][0] += h
        elif path[0] == "W":
            s["dims"][4]["W"] = s["dims"][4]["W"].copy(); s["dims"][4]["W"][path[1]] += h
        elif path[0] == "kappa":
            s["dims"][4]["kappa"] = s["dims"][4]["kappa"].copy(); s["dims"][4]["kappa"][path[1]] += h
        return o.sgpr_elbo(s, X, y, Z, s2)

    for d in (0, 1, 2, 5):          # RBF dims: Gaussian, uniform, MOG, empirical measures
        check(g[d], fd(lambda h: perturbed(("ls", d), h)), rtol=5e-5)
    for d in range(D):
        check(g[D + d], fd(lambda h: perturbed(("bv", d), h)), rtol=5e-5)
    check(g[2 * D], fd(lambda h: perturbed(("ov",), h)))
    check(g[2 * D + 1], fd(lambda h: o.sgpr_elbo(spec, X, y, Z, s2 + h), h=1e-6))     # layout: ls(6) | bv(6) | ov(1) | noise | table
    off, C = desc.cat_blocks[4]
    GB = g[2 * D + 2:][off:off + C * C].reshape(C, C)
    gW, gk = _categorical_chain(spec["dims"][4]["W"], spec["dims"][4]["kappa"], spec["dims"][4]["p"], GB)
    for idx in [(0, 0), (2, 1), (3, 0)]:
        check(gW[idx], fd(lambda h: perturbed(("W", idx), h)), rtol=1e-4)
    for i in range(C):
        check(gk[i], fd(lambda h: perturbed(("kappa", i), h)), rtol=1e-4)


@pytest.mark.parametrize("D,R,share", [(22, 3, False), (27, 2, True), (32, 4, False)])
def test_sgpr_gradient_wide_mixed_kernels(hip, D, R, share):
    """17..32 sub-kernels of mixed type: the backward pair kernel splits the dimensions of a pair over two adjacent lanes
    (interleaved), with chunks of four steps that are all-RBF, all-discrete or mixed, padding dimensions when D < 32,
    trainable base variances (share=False) and the categorical-table gradient."""
    rng = np.random.default_rng(D)
    N, M = 300, 24
    # types in an order that produces every kind of chunk: a run of RBF, then alternating, then a discrete tail
    kinds = ["rbf"] * 9 + ["bin", "rbf", "cat", "rbf", "bin", "rbf"] + ["bin", "cat", "bin", "cat", "bin", "bin", "cat"] + ["rbf"] * 10
    kinds = kinds[:D]
    C = 4
    pc = np.array([0.1, 0.2, 0.3, 0.4])
    X = rng.standard_normal((N, D))
    for d, kd in enumerate(kinds):
        if kd == "bin":
            X[:, d] = (rng.uniform(size=N) < 0.35).astype(float)
        elif kd == "cat":
            X[:, d] = rng.choice(C, size=N, p=pc).astype(float)
    y = (np.sin(X[:, 0]) + X[:, 9 % D] * 0.5 + 0.1 * rng.standard_normal(N)).reshape(-1, 1)
    Z = X[:M].copy()
    spec = o.make_spec(D, R, p0=[0.65 if kd == "bin" else None for kd in kinds], p=[pc if kd == "cat" else None for kd in kinds],
                       lengthscales=list(rng.uniform(0.9, 1.8, D)), share_var_across_orders=share,
                       order_variances=list(rng.uniform(0.3, 0.9, R + 1)) if share else [0.6],
                       base_variances=None if share else list(rng.uniform(0.7, 1.3, D)),
                       cat_W=[rng.uniform(size=(C, 2)) if kd == "cat" else None for kd in kinds],
                       cat_kappa=[rng.uniform(0.5, 1.5, C) if kd == "cat" else None for kd in kinds])
    s2 = 0.1
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    desc = _capi.KernelDesc(spec)
    e, g = hip.sgpr_elbo_grad(desc, s2)
    np.testing.assert_allclose(e, o.sgpr_elbo(spec, X, y, Z, s2), rtol=1e-10)
    rbf_dims = [d for d, kd in enumerate(kinds) if kd == "rbf"]
    for d in (rbf_dims[0], rbf_dims[len(rbf_dims) // 2], rbf_dims[-1]):       # lengthscales in the first, a middle and the last chunk
        def f(h, d=d):
            s = copy.deepcopy(spec); s["dims"][d]["lengthscale"] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(g[d], fd(f), rtol=5e-5)
    if not share:
        for d in (0, 9, 11, 16, D - 1):                                        # base variances of RBF, binary and categorical dims
            def f(h, d=d):
                s = copy.deepcopy(spec); s["dims"][d]["variance"] += h
                return o.sgpr_elbo(s, X, y, Z, s2)
            check(g[D + d], fd(f), rtol=5e-5)
    nov = R + 1 if share else 1
    for r in range(nov):
        def f(h, r=r):
            s = copy.deepcopy(spec); s["order_variances"][r] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(g[2 * D + r], fd(f), rtol=5e-5)
    cat_d = [d for d, kd in enumerate(kinds) if kd == "cat"][-1]               # a categorical dimension's table -> kappa gradient
    off, Cc = desc.cat_blocks[cat_d]
    GB = g[2 * D + nov + 1:][off:off + Cc * Cc].reshape(Cc, Cc)
    gW, gk = _categorical_chain(spec["dims"][cat_d]["W"], spec["dims"][cat_d]["kappa"], spec["dims"][cat_d]["p"], GB)
    for i in (0, Cc - 1):
        def f(h, i=i):
            s = copy.deepcopy(spec); s["dims"][cat_d]["kappa"] = s["dims"][cat_d]["kappa"].copy(); s["dims"][cat_d]["kappa"][i] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(gk[i], fd(f), rtol=1e-4)


def test_gpr_gradient(hip):
    X, y, _ = o.synthetic_problem(150, 3, 4, seed=9)
    spec = o.make_spec(3, 2, lengthscales=[0.9, 1.3, 1.1], order_variances=[0.7, 1.2, 0.9])
    s2 = 0.05
    hip.gpr_set_data(X, y)
    v, g = hip.gpr_log_marginal_grad(_capi.KernelDesc(spec), s2)
    np.testing.assert_allclose(v, o.gpr_log_marginal_likelihood(spec, X, y, s2), rtol=1e-11)
    for d in range(3):
        def f(h, d=d):
            s = copy.deepcopy(spec); s["dims"][d]["lengthscale"] += h
            return o.gpr_log_marginal_likelihood(s, X, y, s2)
        check(g[d], fd(f))
    for r in range(3):
        def f(h, r=r):
            s = copy.deepcopy(spec); s["order_variances"][r] += h
            return o.gpr_log_marginal_likelihood(s, X, y, s2)
        check(g[6 + r], fd(f))
    check(g[9], fd(lambda h: o.gpr_log_marginal_likelihood(spec, X, y, s2 + h), h=1e-6))


def test_model_loss_gradient_through_transforms_and_prior():
    """d training_loss / d unconstrained variables (softplus / sigmoid transforms, Gamma prior) vs finite differences."""
    rng = np.random.default_rng(1)
    X = rng.standard_normal((120, 3)); X[:, 2] = rng.integers(0, 2, 120)
    y = (np.sin(X[:, :1]) + X[:, 1:2] * X[:, 2:3]) + 0.1 * rng.standard_normal((120, 1))
    model = create_model_oak((X, y), inducing_pts=X[:15].copy(), p0=[None, None, 0.5], lengthscale_bounds=[1e-3, 1e3])
    variables = model.trainable_variables
    loss, grads = model._training_loss_and_grad(variables)
    np.testing.assert_allclose(loss, model.training_loss(), rtol=1e-12)
    for p, g in zip(variables, grads):
        u0 = p.unconstrained_variable.copy()
        h = 1e-5
        p._u = u0 + h; lp = model.training_loss()
        p._u = u0 - h; lm = model.training_loss()
        p._u = u0
        np.testing.assert_allclose(float(np.asarray(g).reshape(-1)[0]), (lp - lm) / (2 * h), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("num_inducings", [0, 2])
def test_OrthogonalRBFKernel_optimisation(num_inducings):
    """tests/test_optimisation.py:17-45."""
    X = np.array([[0.0], [1.0], [2.0]]); y = X.copy()
    Z = X[:num_inducings, :] if num_inducings > 0 else None
    k = OrthogonalRBFKernel(gpflow.kernels.RBF(lengthscales=10), GaussianMeasure(0, 1))
    model = gpflow.models.SGPR((X, y), kernel=k, inducing_variable=gpflow.InducingPoints(Z)) if Z is not None else gpflow.models.GPR((X, y), kernel=k)
    if Z is not None:
        gpflow.set_trainable(model.inducing_variable, False)
    initial = model.maximum_log_likelihood_objective()
    assert not np.isnan(initial)
    gpflow.optimizers.Scipy().minimize(model.training_loss_closure(), model.trainable_variables, method="BFGS", compile=True,
                                       options=dict(disp=False, maxiter=2))
    assert initial < model.maximum_log_likelihood_objective()


@pytest.mark.parametrize("num_inducings", [0, 2])
def test_oak_optimisation(num_inducings):
    """tests/test_optimisation.py:48-70."""
    X = np.array([[0.0], [1.0], [2.0]]); y = X.copy()
    Z = X[:num_inducings, :] if num_inducings > 0 else None
    model = create_model_oak((X, y), inducing_pts=Z, optimise=False, zfixed=True)
    initial = model.maximum_log_likelihood_objective()
    assert not np.isnan(initial)
    gpflow.optimizers.Scipy().minimize(model.training_loss_closure(), model.trainable_variables, method="BFGS", compile=True,
                                       options=dict(disp=False, maxiter=2))
    assert initial < model.maximum_log_likelihood_objective()


def test_fit_with_bfgs_improves_the_objective():
    """oak_model.fit(optimise=True): BFGS on the analytic gradient lowers the loss and predicts better than the initial model."""
    from oak.model_utils import oak_model
    rng = np.random.default_rng(3)
    X = rng.standard_normal((400, 3))
    y = (X[:, 0] ** 2 + X[:, 1] + X[:, 1] * X[:, 2] + 0.05 * rng.standard_normal(400)).reshape(-1, 1)
    oak = oak_model(num_inducing=40, max_interaction_depth=2, sparse=True, use_normalising_flow=False)
    oak.fit(X, y, optimise=False)
    before = oak.m.training_loss()
    res = gpflow.optimizers.Scipy().minimize(oak.m.training_loss_closure(), oak.m.trainable_variables, method="BFGS",
                                             options=dict(maxiter=15))
    assert oak.m.training_loss() < before - 1.0
    pred = oak.predict(X[:50])
    assert np.mean((pred - y[:50, 0]) ** 2) < 0.05 * np.var(y)


@pytest.mark.parametrize("route", ["phi", "whitened"])
@pytest.mark.parametrize("kinds,D,R,share", [(("gaussian",), 4, 2, True), (("gaussian", "uniform", "mog", "none", "gauss2"), 5, 3, False),
                                            (("gaussian", "binary", "categorical", "uniform"), 6, 2, True),
                                            (("gaussian",), 20, 2, True),
                                            (("gaussian", "uniform", "binary"), 6, 5, True), (("gaussian", "gauss2"), 10, 8, False),
                                            (("gaussian", "categorical", "mog"), 16, 6, True)])
def test_gradient_wrt_inducing_inputs(hip, route, kinds, D, R, share):
    """oak_sgpr_elbo_grad_z against central differences of the oracle ELBO, entry by entry of Z (every measure type; discrete
    columns get exactly 0), in both routes; the hyper-parameter gradient it returns alongside is unchanged."""
    rng = np.random.default_rng(D * 7 + R)
    spec = cases.random_spec(rng, D, R, kinds, share=share)
    N, M = 150, 9
    X = cases.random_inputs(rng, spec, N)
    Z = cases.random_inputs(rng, spec, M)
    y = rng.standard_normal((N, 1))
    s2 = 0.15
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route(route)
    e0, g0 = hip.sgpr_elbo_grad(d, s2)
    e, g, gz = hip.sgpr_elbo_grad_z(d, s2, M, D)
    assert e == e0 and gz.shape == (M, D)
    np.testing.assert_allclose(g, g0, rtol=1e-12, atol=1e-12 * np.abs(g0).max())   # categorical-table sums use LDS atomics: order varies
    probe = [(m, c) for m in (0, M // 2, M - 1) for c in range(D)]
    for (m, c) in probe:
        if spec["dims"][c]["type"] != "rbf":
            assert gz[m, c] == 0.0
            continue
        h = 1e-5
        Zp, Zm = Z.copy(), Z.copy()
        Zp[m, c] += h; Zm[m, c] -= h
        fd = (o.sgpr_elbo(spec, X, y, Zp, s2) - o.sgpr_elbo(spec, X, y, Zm, s2)) / (2 * h)
        assert abs(gz[m, c] - fd) <= 2e-5 * max(1.0, abs(fd)), f"Z[{m},{c}] ({spec['dims'][c].get('measure')}): {gz[m, c]} vs {fd}"


@pytest.mark.parametrize("kinds,D,R,share", [(("gaussian", "uniform"), 12, 10, True), (("gaussian", "binary", "mog"), 20, 6, True),
                                            (("gaussian", "categorical", "none"), 36, 3, False), (("gaussian",), 40, 2, True),
                                            (("gaussian", "gauss2"), 18, 18, True)])
def test_gradient_wrt_inducing_inputs_beyond_the_register_resident_shapes(hip, kinds, D, R, share):
    """Depth > 8, or > 16 dims above depth 4, or > 32 dims: the general inducing-input kernel (one column per lane, per-wave LDS
    accumulators).  Entries of Z against central differences of the oracle; discrete columns exactly 0."""
    rng = np.random.default_rng(D * 11 + R)
    spec = cases.random_spec(rng, D, R, kinds, share=share)
    N, M = 140, 70                       # two column blocks of 64, the second one ragged
    X = cases.random_inputs(rng, spec, N)
    Z = cases.random_inputs(rng, spec, M)
    y = rng.standard_normal((N, 1))
    s2 = 0.2
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    e0, g0 = hip.sgpr_elbo_grad(d, s2)
    e, g, gz = hip.sgpr_elbo_grad_z(d, s2, M, D)
    assert e == e0 and gz.shape == (M, D)
    np.testing.assert_allclose(g, g0, rtol=1e-12, atol=1e-12 * np.abs(g0).max())
    for (m, c) in [(m, c) for m in (0, 63, 64, M - 1) for c in (0, 1, 2, D // 2, D - 1)]:
        if spec["dims"][c]["type"] != "rbf":
            assert gz[m, c] == 0.0
            continue
        h = 1e-5
        Zp, Zm = Z.copy(), Z.copy()
        Zp[m, c] += h; Zm[m, c] -= h
        fd = (o.sgpr_elbo(spec, X, y, Zp, s2) - o.sgpr_elbo(spec, X, y, Zm, s2)) / (2 * h)
        assert abs(gz[m, c] - fd) <= 3e-5 * max(1.0, abs(fd)), f"Z[{m},{c}]: {gz[m, c]} vs {fd}"


@pytest.mark.parametrize("D,R,kinds,share", [(7, 6, ("gaussian", "binary", "uniform", "categorical", "mog", "none", "gauss2"), False),
                                              (20, 5, ("gaussian",), True), (12, 8, ("gaussian", "gauss2"), True),
                                              (30, 7, ("gaussian", "binary", "categorical"), True),
                                              (13, 11, ("gaussian",), True), (16, 16, ("gaussian", "binary", "mog", "categorical"), False),
                                              (14, 9, ("gaussian", "uniform"), True), (24, 12, ("gaussian",), True),
                                              (32, 16, ("gaussian", "binary", "gauss2", "categorical"), True), (20, 9, ("gaussian", "mog"), False),
                                              (40, 3, ("gaussian",), True), (64, 2, ("gaussian", "binary", "categorical", "uniform"), False),
                                              (50, 7, ("gaussian", "categorical", "gauss2"), True), (33, 1, ("gaussian", "binary"), True)])
def test_fast_and_general_backward_kernels_agree_at_depth_5_to_8(monkeypatch, D, R, kinds, share):
    """OAK_BWD_GENERIC=1 sends the depth-5..16 shapes and the 33..64-sub-kernel shapes back through the general two-pass kernel they used before r04: the whole
    gradient record (lengthscales, base variances, order variances, categorical table sums) agrees to rounding at a size with
    several row and column blocks."""
    rng = np.random.default_rng(D * 100 + R)
    spec = cases.random_spec(rng, D, R, kinds, share=share)
    N, M = 2500, 150
    X = cases.random_inputs(rng, spec, N)
    Z = cases.random_inputs(rng, spec, M)
    y = rng.standard_normal((N, 1))
    d = _capi.KernelDesc(spec)
    out = {}
    for mode in ("fast", "general"):
        if mode == "general":
            monkeypatch.setenv("OAK_BWD_GENERIC", "1")
        ctx = _capi.HipContext(0)
        ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("whitened")
        out[mode] = ctx.sgpr_elbo_grad(d, 0.1)
        ctx.close()
    assert out["fast"][0] == out["general"][0]
    gf, gg = out["fast"][1].copy(), out["general"][1].copy()
    if not d.struct.grad_base_var:         # base variances are constants then: their slots are not part of the result (the general kernel
        gf[D:2 * D] = gg[D:2 * D] = 0.0    # fills them anyway, the plain fast form leaves zeros)
    scale = np.abs(gg).max()
    np.testing.assert_allclose(gf, gg, rtol=1e-9, atol=1e-11 * scale)


def test_general_and_register_resident_inducing_input_kernels_agree(monkeypatch):
    """OAK_BWDZ_GENERAL=1 sends a shape the fast kernels cover through the general one: same gradient to rounding, at a size
    with several row blocks and every sub-kernel type."""
    rng = np.random.default_rng(77)
    spec = cases.random_spec(rng, 7, 3, ("gaussian", "binary", "uniform", "categorical", "mog", "none", "gauss2"), share=False)
    N, M = 3000, 150
    X = cases.random_inputs(rng, spec, N)
    Z = cases.random_inputs(rng, spec, M)
    y = rng.standard_normal((N, 1))
    d = _capi.KernelDesc(spec)
    out = {}
    for mode in ("fast", "general"):
        if mode == "general":
            monkeypatch.setenv("OAK_BWDZ_GENERAL", "1")
        ctx = _capi.HipContext(0)
        ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("whitened")
        out[mode] = ctx.sgpr_elbo_grad_z(d, 0.1, M, 7)
        ctx.close()
    assert out["fast"][0] == out["general"][0]
    np.testing.assert_array_equal(out["fast"][1], out["general"][1])
    scale = np.abs(out["fast"][2]).max()
    np.testing.assert_allclose(out["general"][2], out["fast"][2], rtol=0, atol=1e-11 * scale)


def test_trainable_inducing_inputs_through_the_model_api():
    """create_model_oak(zfixed=False) (oak/model_utils.py:156-157): Z joins the trainable variables, the model-level loss
    gradient matches central differences of the loss in a few Z entries, and BFGS moves the inducing points while improving
    the objective."""
    rng = np.random.default_rng(12)
    X = rng.normal(size=(400, 3))
    y = (np.sin(X[:, :1]) + 0.5 * X[:, 1:2] * X[:, 2:3] + 0.1 * rng.normal(size=(400, 1)))
    Z0 = X[:12].copy()
    model = create_model_oak((X, y), inducing_pts=Z0.copy(), optimise=False, zfixed=False)
    Zp = model.inducing_variable.Z
    variables = model.trainable_variables
    assert any(v is Zp for v in variables)
    loss, grads = model.training_loss_closure().value_and_grad(variables)
    gz = grads[[i for i, v in enumerate(variables) if v is Zp][0]]
    assert gz.shape == Z0.shape
    for (m, c) in ((0, 0), (5, 2), (11, 1)):
        h = 1e-5
        Zv = Zp.numpy().copy()
        Zv[m, c] += h; Zp.assign(Zv); lp = model.training_loss()
        Zv[m, c] -= 2 * h; Zp.assign(Zv); lm = model.training_loss()
        Zv[m, c] += h; Zp.assign(Zv)
        fd = (lp - lm) / (2 * h)
        assert abs(gz[m, c] - fd) <= 5e-5 * max(1.0, abs(fd))
    before = model.training_loss()
    gpflow.optimizers.Scipy().minimize(model.training_loss_closure(), variables, method="BFGS", options=dict(maxiter=5))
    assert model.training_loss() < before
    assert np.abs(model.inducing_variable.Z.numpy() - Z0).max() > 1e-6


def _shared_spec_with_trainable_base_variances():
    """share_var_across_orders=True with empirical- and MOG-measure dims: the reference pins the base variance to 1 only
    for Gaussian-measure / binary / categorical dims (oak_kernel.py:163-166,179,187), the other RBF dims keep a trainable
    base_kernel.variance."""
    spec, X, y, Z, s2 = cases.case_B()
    spec = copy.deepcopy(spec)
    spec["dims"][2]["variance"] = 0.8      # MOG measure
    spec["dims"][5]["variance"] = 1.3      # empirical measure
    spec["dims"][1]["variance"] = 1.15     # uniform measure
    spec["base_var_grad"] = True
    return spec, X, y, Z, s2


@pytest.mark.parametrize("route", ["phi", "whitened"])
def test_sgpr_base_variance_gradient_under_shared_order_variances(hip, route):
    spec, X, y, Z, s2 = _shared_spec_with_trainable_base_variances()
    D = len(spec["dims"])
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route(route)
    e, g = hip.sgpr_elbo_grad(_capi.KernelDesc(spec), s2)
    np.testing.assert_allclose(e, o.sgpr_elbo(spec, X, y, Z, s2), rtol=1e-9)
    for d in (1, 2, 5):
        def f(h, d=d):
            s = copy.deepcopy(spec); s["dims"][d]["variance"] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(g[D + d], fd(f), rtol=5e-5)
    # without the flag the library is told the base variances are constants: it must not be what the model layer passes
    nograd = copy.deepcopy(spec); nograd["base_var_grad"] = False
    _, g0 = hip.sgpr_elbo_grad(_capi.KernelDesc(nograd), s2)
    np.testing.assert_allclose(g0[:D], g[:D], rtol=1e-10)          # every other block is unaffected
    np.testing.assert_allclose(g0[2 * D:], g[2 * D:], rtol=1e-10, atol=1e-12 * np.abs(g).max())


@pytest.mark.parametrize("sparse", [True, False])
def test_model_gradient_with_empirical_and_gmm_measures(sparse):
    """oak_model(empirical_measure=..., gmm_measure=...) keeps base_kernel.variance trainable for those dims; the loss
    gradient handed to BFGS must match central differences of the loss in EVERY trainable variable (the pair
    contribution to d/d variance was dropped when share_var_across_orders=True)."""
    from oak.model_utils import oak_model
    from oak.oak_kernel import kernel_to_spec
    rng = np.random.default_rng(17)
    N = 260
    X = rng.standard_normal((N, 4))
    X[:, 3] = rng.integers(0, 2, N)
    y = (np.sin(X[:, 0]) + 0.6 * X[:, 1] * X[:, 2] + 0.4 * X[:, 3] + 0.1 * rng.standard_normal(N)).reshape(-1, 1)
    oak = oak_model(max_interaction_depth=2, num_inducing=30, sparse=sparse, binary_feature=[3], empirical_measure=[1],
                    gmm_measure=[0, 0, 2, 0], use_normalising_flow=True)   # GMM dims need the flow branch (model_utils.py:305-317,350-353)
    oak.fit(X, y, optimise=False)
    model = oak.m
    spec = kernel_to_spec(model.kernel)
    assert spec["base_var_grad"] is True
    variables = model.trainable_variables
    n_var = sum(isinstance(getattr(getattr(k, "base_kernel", k), "variance", None), gpflow.Parameter) for k in model.kernel.kernels)
    assert n_var == 2                                            # the empirical and the GMM dim
    for k in model.kernel.kernels:                               # move off the initial values
        base = getattr(k, "base_kernel", k)
        if isinstance(getattr(base, "variance", None), gpflow.Parameter):
            base.variance.assign(float(rng.uniform(0.6, 1.6)))
    loss, grads = model._training_loss_and_grad(variables)
    for p, g in zip(variables, grads):
        u0 = p.unconstrained_variable.copy()
        h = 1e-5
        p._u = u0 + h; lp = model.training_loss()
        p._u = u0 - h; lm = model.training_loss()
        p._u = u0
        np.testing.assert_allclose(float(np.asarray(g).reshape(-1)[0]), (lp - lm) / (2 * h), rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("D,R", [(10, 10), (13, 13)])
def test_sgpr_elbo_and_gradient_beyond_depth_eight(hip, D, R):
    """Full-order models as the reference's regression example builds them (max_interaction_depth = D, D = 13 for UCI
    housing): ELBO against the oracle, gradient against central differences of the oracle."""
    rng = np.random.default_rng(D + R)
    N, M = 220, 18
    X, y, Z = o.synthetic_problem(N, D, M, seed=D)
    spec = o.make_spec(D, R, lengthscales=list(rng.uniform(0.9, 1.8, D)), order_variances=list(rng.uniform(0.05, 0.6, R + 1)))
    s2 = 0.1
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    e, g = hip.sgpr_elbo_grad(_capi.KernelDesc(spec), s2)
    np.testing.assert_allclose(e, o.sgpr_elbo(spec, X, y, Z, s2), rtol=1e-9)
    for d in (0, D // 2, D - 1):
        def f(h, d=d):
            s = copy.deepcopy(spec); s["dims"][d]["lengthscale"] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(g[d], fd(f), rtol=1e-4)
    for r in (0, 1, R // 2, R):
        def f(h, r=r):
            s = copy.deepcopy(spec); s["order_variances"][r] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(g[2 * D + r], fd(f), rtol=1e-4)
    check(g[2 * D + R + 1], fd(lambda h: o.sgpr_elbo(spec, X, y, Z, s2 + h), h=1e-6), rtol=1e-4)
    m, v = hip.sgpr_predict(_capi.KernelDesc(spec), X[:40])
    mr, vr = o.sgpr_predict_f(spec, X, y, Z, s2, X[:40])
    assert np.abs(m - mr[:, 0]).max() <= 1e-8 and np.abs(v - vr[:, 0]).max() <= 1e-8


def test_mixed_kernel_gradient_is_bitwise_repeatable(hip):
    """Every output of the library is a fixed-order reduction, the categorical-table gradients included: they are summed with
    LDS atomics, but each wave adds into its own copy (a wave's instructions are ordered and one ds_add serialises colliding
    lanes in a fixed order), and the copies are combined in a fixed order.  40 evaluations of a kernel with binary and
    categorical dimensions on a problem large enough for many workgroups per launch must agree bit for bit."""
    rng = np.random.default_rng(77)
    D, R, N, M = 9, 3, 20000, 96
    spec = cases.random_spec(rng, D, R, ("gaussian", "categorical", "binary", "uniform", "categorical"))
    X = cases.random_inputs(rng, spec, N)
    Z = X[:M].copy()
    y = rng.standard_normal((N, 1))
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("phi")
    e0, g0 = hip.sgpr_elbo_grad(d, 0.2)
    assert np.abs(g0[2 * D + R + 2:]).max() > 0                       # the table block is populated
    for _ in range(40):
        e, g = hip.sgpr_elbo_grad(d, 0.2)
        assert e == e0 and np.array_equal(g, g0)


def test_whitened_gradient_on_an_ill_conditioned_kuu_uses_the_posteriors_alpha(hip):
    """Inducing points that nearly coincide in pairs (cond(Kuu + jitter I) ~ 1e8-1e9): on the whitened route the backward pass
    takes a = L^-T LB^-T c from the two transposed triangular solves -- the same vector oak_sgpr_alpha returns -- and the
    gradient matches central differences of the whitened bound (round-3 advisor finding: the explicit (LB^-1 L^-1)^T loses
    ~cond(Kuu) ulps on exactly the problems this route exists for)."""
    import copy
    rng = np.random.default_rng(12)
    N, D, M = 2500, 3, 64
    X = rng.standard_normal((N, D))
    Zh = rng.standard_normal((M // 2, D))
    Z = np.concatenate([Zh, Zh + 2e-4 * rng.standard_normal(Zh.shape)])
    y = (np.sin(X[:, 0]) + 0.3 * X[:, 1] * X[:, 2] + 0.1 * rng.standard_normal(N)).reshape(-1, 1)
    spec = o.make_spec(D, 2, lengthscales=[1.2, 0.9, 1.5], order_variances=[0.5, 1.0, 0.7])
    Kuu = o.oak_K(spec, Z) + 1e-6 * np.eye(M)
    assert np.linalg.cond(Kuu) > 1e7
    d = _capi.KernelDesc(spec)
    ctx = _capi.HipContext(0)
    try:
        ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("whitened")
        e, g = ctx.sgpr_elbo_grad(d, 0.05)
        assert abs(e - o.sgpr_elbo(spec, X, y, Z, 0.05)) <= 1e-9 * abs(e)
        alpha = ctx.sgpr_alpha(M)
        ref_alpha = o.sgpr_alpha(spec, X, y, Z, 0.05)[:, 0]
        np.testing.assert_allclose(alpha, ref_alpha, rtol=0, atol=1e-6 * np.abs(ref_alpha).max())

        def bound(mod):
            s = copy.deepcopy(spec); mod(s)
            return ctx.sgpr_elbo(_capi.KernelDesc(s), 0.05)
        for i in range(D):
            h = 1e-4
            fd = (bound(lambda s: s["dims"][i].__setitem__("lengthscale", spec["dims"][i]["lengthscale"] + h))
                  - bound(lambda s: s["dims"][i].__setitem__("lengthscale", spec["dims"][i]["lengthscale"] - h))) / (2 * h)
            assert abs(g[i] - fd) <= 2e-5 * max(1.0, abs(fd)), (i, g[i], fd)
        fdn = (ctx.sgpr_elbo(d, 0.05 + 1e-5) - ctx.sgpr_elbo(d, 0.05 - 1e-5)) / 2e-5
        assert abs(g[-1] - fdn) <= 2e-5 * abs(fdn), (g[-1], fdn)
    finally:
        ctx.close()


@pytest.mark.parametrize("R,D", [(1, 5), (2, 8), (2, 16), (3, 11), (4, 7)])
def test_rows_in_lanes_backward_kernel_equals_the_columns_in_lanes_kernel(hip, R, D, monkeypatch):
    """csrc/grad_rows.hip (a lane owns a ROW, the column features are scalar loads) is an opt-in alternative to the default
    backward pair kernel for the reference's default model (continuous inputs, unit base variances, depth <= 4): same record,
    different summation order -- the two gradients agree to rounding, and with a ragged row and column count."""
    X, y, Z = o.synthetic_problem(16384 + 77, D, 200 + 13, seed=31)
    spec = o.make_spec(D, R, lengthscales=list(np.linspace(0.7, 1.6, D)), order_variances=list(np.linspace(0.6, 1.3, R + 1)))
    d = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("phi")
    monkeypatch.setenv("OAK_BWD_ROWS", "0")
    e0, g0 = hip.sgpr_elbo_grad(d, 0.05)
    monkeypatch.setenv("OAK_BWD_ROWS", "1")
    e1, g1 = hip.sgpr_elbo_grad(d, 0.05)
    assert e0 == e1
    np.testing.assert_allclose(g1, g0, rtol=1e-10, atol=1e-11 * np.abs(g0).max())
    hip.sgpr_set_route("auto")

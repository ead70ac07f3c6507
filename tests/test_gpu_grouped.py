"""GPU: grouped sub-kernels on the MODEL paths.  ``OAKKernel(active_dims=[[0, 1], [2], ...], constrain_orthogonal=False)`` makes each
group ONE base kernel over the group's columns (oak/oak_kernel.py:74-82,199-210) -- for the RBF a product of one-column RBFs with a
shared lengthscale, i.e. one exponential of the summed squared differences.  The fused pair kernels take the further columns as extra
feature rows (DevDesc::xrow / nxc, Feat::xx): SGPR bound and its terms, hyper-parameter gradients, predictions, GPR and SVGP against
the CPU oracle (which evaluates the group with ``rbf_K`` on the group's columns) and against central differences of it."""
import copy

import numpy as np
import pytest

import cases
from oak import _capi
from oak import gpflow_lite as gpflow
from oak.oak_kernel import OAKKernel, kernel_to_spec
from oracle import oak_oracle as o

pytestmark = pytest.mark.gpu


def grouped_spec(rng, groups, R, share=True, binary_at=None):
    dims = []
    for d, g in enumerate(groups):
        if binary_at is not None and d == binary_at:
            dims.append(dict(type="binary", p0=0.4, variance=1.0 if share else 0.8, active_dim=g[0]))
            continue
        dim = dict(type="rbf", lengthscale=float(rng.uniform(0.9, 2.2)) * np.sqrt(len(g)), variance=1.0 if share else float(rng.uniform(0.6, 1.6)),
                   measure=None, active_dim=g[0])
        if len(g) > 1:
            dim["active_dims"] = list(g)
        dims.append(dim)
    return dict(dims=dims, order_variances=list(rng.uniform(0.4, 1.2, R + 1 if share else 1)), max_interaction_depth=R,
                share_var_across_orders=share)


def problem(rng, n, m, ncol, binary_col=None):
    X = rng.standard_normal((n, ncol))
    if binary_col is not None:
        X[:, binary_col] = rng.integers(0, 2, n)
    y = (np.sin(X[:, 0] + 0.5 * X[:, 1]) + 0.3 * X[:, 2] * X[:, -1] + 0.1 * rng.standard_normal(n)).reshape(-1, 1)
    y = (y - y.mean()) / y.std()
    return X, y, X[rng.choice(n, m, replace=False)].copy()


def fd(fun, h=1e-5):
    return (fun(h) - fun(-h)) / (2 * h)


def check(g, ref, rtol=5e-5, atol=1e-6):
    np.testing.assert_allclose(g, ref, rtol=rtol, atol=atol * max(1.0, abs(ref)))


GROUPS = [[0, 1], [2], [4, 3, 5]]


@pytest.mark.parametrize("route", ["phi", "whitened"])
@pytest.mark.parametrize("R", [1, 2, 3])
def test_sgpr_bound_terms_and_gradient(hip, route, R):
    rng = np.random.default_rng(10 + R)
    spec = grouped_spec(rng, GROUPS, R)
    X, y, Z = problem(rng, 320, 28, 6)
    s2 = 0.06
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route(route)
    desc = _capi.KernelDesc(spec)
    ref = o.sgpr_elbo(spec, X, y, Z, s2)
    np.testing.assert_allclose(hip.sgpr_elbo(desc, s2), ref, rtol=1e-10 if route == "whitened" else 1e-8)
    if route == "whitened":
        cases.assert_terms_match(hip.sgpr_last_terms(), o.sgpr_elbo_terms(spec, X, y, Z, s2), rtol=1e-9, what="grouped kernel:")
    e, g = hip.sgpr_elbo_grad(desc, s2)
    np.testing.assert_allclose(e, ref, rtol=1e-10 if route == "whitened" else 1e-8)
    D = len(GROUPS)

    def moved(path, h):
        s = copy.deepcopy(spec)
        if path[0] == "ls":
            s["dims"][path[1]]["lengthscale"] += h
        else:
            s["order_variances"][path[1]] += h
        return o.sgpr_elbo(s, X, y, Z, s2)

    for d in range(D):
        check(g[d], fd(lambda h: moved(("ls", d), h)))
    for r in range(R + 1):
        check(g[2 * D + r], fd(lambda h: moved(("ov", r), h)))
    check(g[2 * D + R + 1], fd(lambda h: o.sgpr_elbo(spec, X, y, Z, s2 + h), h=1e-6))


def test_base_variances_and_a_binary_next_to_the_groups(hip):
    """share_var_across_orders=False: every sub-kernel's own variance is trainable; a binary sub-kernel (the unconstrained branch
    builds OrthogonalBinary too, oak_kernel.py:205-207) sits between the groups."""
    rng = np.random.default_rng(3)
    groups = [[0, 1, 2], [3], [4], [5, 6]]
    spec = grouped_spec(rng, groups, 2, share=False, binary_at=1)
    X, y, Z = problem(rng, 280, 24, 7, binary_col=3)
    s2 = 0.09
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    desc = _capi.KernelDesc(spec)
    e, g = hip.sgpr_elbo_grad(desc, s2)
    np.testing.assert_allclose(e, o.sgpr_elbo(spec, X, y, Z, s2), rtol=1e-10)
    D = len(groups)

    def moved(key, d, h):
        s = copy.deepcopy(spec); s["dims"][d][key] += h
        return o.sgpr_elbo(s, X, y, Z, s2)

    for d in (0, 2, 3):
        check(g[d], fd(lambda h: moved("lengthscale", d, h)))
    for d in range(D):
        check(g[D + d], fd(lambda h: moved("variance", d, h)))


@pytest.mark.parametrize("ncol,R", [(20, 2), (36, 2), (12, 4), (10, 6)])
def test_kernel_shape_variants_at_a_size_that_tiles(hip, ncol, R):
    """More than 16 sub-kernels (the narrower column tile), depth > 4 (one row per lane), and enough rows for the tiled featurize
    kernel and several row blocks: the bound against the oracle, three gradient entries against its differences."""
    rng = np.random.default_rng(ncol)
    cols = list(rng.permutation(ncol))
    groups, i = [], 0
    while i < ncol:                                      # groups of 1, 2, 3, 1, 2, 3, ... columns in a shuffled order
        k = min(1 + len(groups) % 3, ncol - i)
        groups.append([int(c) for c in cols[i:i + k]]); i += k
    spec = grouped_spec(rng, groups, R)
    X, y, Z = problem(rng, 4500, 40, ncol)
    s2 = 0.1
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    desc = _capi.KernelDesc(spec)
    e, g = hip.sgpr_elbo_grad(desc, s2)
    np.testing.assert_allclose(e, o.sgpr_elbo(spec, X, y, Z, s2), rtol=1e-10)
    D = len(groups)
    for d in (1, D // 2, D - 1):
        def f(h, d=d):
            s = copy.deepcopy(spec); s["dims"][d]["lengthscale"] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(g[d], fd(f), rtol=1e-4)
    # the explicit Gram entry point (the same fused pair kernel, no statistics around it) against the oracle, entry by entry
    K = hip.gram(desc, X[:300], Z)
    np.testing.assert_allclose(K, o.oak_K(spec, X[:300], Z), rtol=0, atol=1e-12 * np.abs(K).max())


def test_predictions_gpr_and_what_stays_refused(hip):
    rng = np.random.default_rng(8)
    spec = grouped_spec(rng, GROUPS, 2)
    X, y, Z = problem(rng, 240, 30, 6)
    Xs = rng.standard_normal((50, 6))
    s2 = 0.05
    desc = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    hip.sgpr_elbo(desc, s2)
    mean, var = hip.sgpr_predict(desc, Xs)
    rm, rv = o.sgpr_predict_f(spec, X, y, Z, s2, Xs)
    np.testing.assert_allclose(mean, rm.ravel(), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(var, rv.ravel(), rtol=1e-7, atol=1e-9)
    # full GP
    hip.gpr_set_data(X, y)
    lml, g = hip.gpr_log_marginal_grad(desc, s2)
    np.testing.assert_allclose(lml, o.gpr_log_marginal_likelihood(spec, X, y, s2), rtol=1e-10)

    def f(h):
        s = copy.deepcopy(spec); s["dims"][2]["lengthscale"] += h
        return o.gpr_log_marginal_likelihood(s, X, y, s2)
    check(g[2], fd(f))
    gm, gv = hip.gpr_predict(desc, Xs)
    rm, rv = o.gpr_predict_f(spec, X, y, s2, Xs)
    np.testing.assert_allclose(gm, rm.ravel(), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(gv, rv.ravel(), rtol=1e-7, atol=1e-9)
    # the Sobol pass keeps one column per sub-kernel (the reference defines no Sobol index for an unconstrained kernel), and says so
    with pytest.raises(ValueError, match="several columns"):
        hip.sobol(desc, Z, np.ones(len(Z)), [[0], [0, 1]])
    # the fp32 statistics mode is not taken for a grouped kernel: same numbers as fp64
    ref = hip.sgpr_elbo(desc, s2)
    hip.sgpr_set_route("phi")
    e64 = hip.sgpr_elbo(desc, s2)
    hip.sgpr_set_precision("fp32")
    try:
        assert hip.sgpr_elbo(desc, s2) == e64
        assert hip.sgpr_stats_precision() == "fp64"
    finally:
        hip.sgpr_set_precision("fp64")
    np.testing.assert_allclose(e64, ref, rtol=1e-8)


def test_model_api_trains_a_grouped_kernel(hip):
    """Through the host mirror: gpflow.SGPR over OAKKernel(active_dims=[[0, 1], [2]], constrain_orthogonal=False); loss and gradient
    against the oracle / its differences in the unconstrained parameters, then a few optimiser steps lower the loss."""
    rng = np.random.default_rng(21)
    X, y, Z = problem(rng, 300, 20, 3)
    k = OAKKernel([gpflow.kernels.RBF, gpflow.kernels.RBF], num_dims=3, max_interaction_depth=2, active_dims=[[0, 1], [2]],
                  constrain_orthogonal=False)
    k.kernels[0].lengthscales.assign(1.4); k.kernels[1].lengthscales.assign(0.9)
    m = gpflow.SGPR((X, y), kernel=k, inducing_variable=Z)
    gpflow.set_trainable(m.inducing_variable, False)
    m.likelihood.variance.assign(0.08)
    spec = kernel_to_spec(k)
    assert spec["dims"][0]["active_dims"] == [0, 1]
    np.testing.assert_allclose(float(m.elbo()), o.sgpr_elbo(spec, X, y, Z, 0.08), rtol=1e-9)
    loss0 = float(m.training_loss())
    gpflow.optimizers.Scipy().minimize(m.training_loss_closure(), m.trainable_variables, method="BFGS", options=dict(maxiter=8))
    assert float(m.training_loss()) < loss0 - 1e-3
    mean, var = m.predict_f(X[:10])
    spec = kernel_to_spec(k)
    rm, rv = o.sgpr_predict_f(spec, X, y, Z, float(m.likelihood.variance.numpy()), X[:10])
    np.testing.assert_allclose(mean.numpy().ravel(), rm.ravel(), rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(var.numpy().ravel(), rv.ravel(), rtol=1e-6, atol=1e-8)


def test_svgp_bound_gradient_and_prediction(hip):
    """The classification model (whitened diagonal q, Bernoulli) over a grouped kernel: bound and prediction against
    oracle/svgp_oracle.py, a lengthscale and an order-variance gradient against its differences."""
    from oracle import svgp_oracle as sv
    rng = np.random.default_rng(17)
    spec = grouped_spec(rng, GROUPS, 2)
    X, _, Z = problem(rng, 350, 30, 6)
    y = (rng.uniform(size=350) < 1.0 / (1.0 + np.exp(-2 * np.sin(X[:, 0] + X[:, 1])))).astype(float)
    q_mu, q_sqrt = 0.7 * rng.standard_normal(30), rng.uniform(0.2, 1.2, 30)
    desc = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y.reshape(-1, 1)); hip.sgpr_set_inducing(Z)
    e, g, gm, gs = hip.svgp_elbo(desc, q_mu, q_sqrt, grad=True)
    er = sv.svgp_elbo(spec, X, y, Z, q_mu, q_sqrt)
    assert abs(e - er) <= 1e-10 * abs(er)

    def f(h):
        s = copy.deepcopy(spec); s["dims"][0]["lengthscale"] += h
        return sv.svgp_elbo(s, X, y, Z, q_mu, q_sqrt)

    def fo(h):
        s = copy.deepcopy(spec); s["order_variances"][2] += h
        return sv.svgp_elbo(s, X, y, Z, q_mu, q_sqrt)
    check(g[0], fd(f))
    check(g[2 * 3 + 2], fd(fo))
    Xs = rng.standard_normal((40, 6))
    m, v = hip.svgp_predict(desc, q_mu, q_sqrt, Xs)
    mr, vr = sv.conditional(spec, Xs, Z, q_mu, q_sqrt)
    np.testing.assert_allclose(m, mr, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(v, vr, rtol=1e-9, atol=1e-9)


def test_components_of_a_grouped_kernel(hip):
    """KernelComponenent.K / K_diag (oak/oak_kernel.py:300-335) multiply the sub-kernels of a subset, each on its own columns:
    a subset that holds a group (alone, first, last), through the generic kernel (explicit entry point) and the fused one
    (component predictions = K_S(x*, Xc) alpha)."""
    rng = np.random.default_rng(5)
    spec = grouped_spec(rng, GROUPS, 3)
    desc = _capi.KernelDesc(spec)
    X, X2 = rng.standard_normal((60, 6)), rng.standard_normal((37, 6))
    alpha = rng.standard_normal(37)
    subsets = [[0], [2], [1, 2], [0, 1], [0, 1, 2]]
    for S in subsets:
        ref = o.component_K(spec, S, X, X2)
        got = hip.gram_component(desc, S, True, X, X2)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        np.testing.assert_allclose(hip.gram_component_diag(desc, S, True, X), o.component_K_diag(spec, S, X), rtol=1e-13)
    pred = hip.component_predict(desc, X, X2, alpha, subsets)
    for row, S in zip(pred, subsets):
        ref = o.component_K(spec, S, X, X2) @ alpha
        np.testing.assert_allclose(row, ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()))
    # the same subsets in the reference's arithmetic (oak_set_gram_form): the generic kernel with the COMPONENT's weights
    # (it once read the whole kernel's order variances here)
    try:
        hip.set_gram_form("reference")
        for S in subsets:
            ref = o.component_K(spec, S, X, X2)
            np.testing.assert_allclose(hip.gram_component(desc, S, True, X, X2), ref, rtol=0, atol=1e-13 * np.abs(ref).max())
        specB = cases.case_B()[0]
        XB = cases.random_inputs(rng, specB, 40)
        for S in ([1], [0, 3], [2, 4, 5]):
            ref = o.component_K(specB, S, XB, XB[:17])
            np.testing.assert_allclose(hip.gram_component(_capi.KernelDesc(specB), S, True, XB, XB[:17]), ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    finally:
        hip.set_gram_form("native")


@pytest.mark.parametrize("route", ["phi", "whitened"])
def test_gradient_wrt_the_inducing_inputs_of_a_grouped_kernel(hip, route):
    """Every column of a group is an input of the model: dF/dZ[m, c] for the group's first AND further columns against central
    differences of the oracle (general inducing-input kernel: slot D + q for further column q)."""
    rng = np.random.default_rng(31)
    groups = [[0, 1], [2], [4, 3, 5], [6]]
    spec = grouped_spec(rng, groups, 3, binary_at=3)
    X, y, Z = problem(rng, 200, 70, 7, binary_col=6)
    s2 = 0.12
    desc = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route(route)
    e0, g0 = hip.sgpr_elbo_grad(desc, s2)
    e, g, gz = hip.sgpr_elbo_grad_z(desc, s2, len(Z), 7)
    assert e == e0 and gz.shape == (70, 7)
    np.testing.assert_allclose(g, g0, rtol=1e-12, atol=1e-12 * np.abs(g0).max())
    assert np.all(gz[:, 6] == 0.0)                     # the binary column
    for m in (0, 33, 64, 69):
        for c in range(6):
            h = 1e-5
            Zp, Zm = Z.copy(), Z.copy()
            Zp[m, c] += h; Zm[m, c] -= h
            ref = (o.sgpr_elbo(spec, X, y, Zp, s2) - o.sgpr_elbo(spec, X, y, Zm, s2)) / (2 * h)
            assert abs(gz[m, c] - ref) <= 3e-5 * max(1.0, abs(ref)), f"Z[{m},{c}]: {gz[m, c]} vs {ref}"


@pytest.mark.parametrize("seed", range(10))
def test_random_grouped_problems(hip, seed):
    """Random groupings of 3..14 columns (groups of 1..4, shuffled), depth, route, sizes and variances: the bound against the
    oracle, two hyper-parameter entries and two inducing-input entries against its differences.  OAK_FUZZ_SEED0 shifts the streams."""
    import os
    rng = np.random.default_rng(int(os.environ.get("OAK_FUZZ_SEED0", "0")) + 21000 + seed)
    ncol = int(rng.integers(3, 15))
    cols = [int(c) for c in rng.permutation(ncol)]
    groups, i = [], 0
    while i < ncol:
        k = min(int(rng.integers(1, 5)), ncol - i)
        groups.append(cols[i:i + k]); i += k
    D = len(groups)
    R = int(rng.integers(1, min(D, 5) + 1))
    share = bool(rng.integers(0, 2))
    spec = grouped_spec(rng, groups, R, share=share)
    N, M = int(rng.integers(100, 900)), int(rng.choice([20, 33, 64, 70]))
    X, y, Z = problem(rng, N, M, ncol)
    s2 = float(rng.uniform(0.05, 0.4))
    route = ("whitened", "phi")[seed % 2]
    desc = _capi.KernelDesc(spec)
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route(route)
    e, g, gz = hip.sgpr_elbo_grad_z(desc, s2, M, ncol)
    np.testing.assert_allclose(e, o.sgpr_elbo(spec, X, y, Z, s2), rtol=1e-10 if route == "whitened" else 1e-8)
    for d in rng.choice(D, size=min(2, D), replace=False):
        def f(h, d=int(d)):
            s = copy.deepcopy(spec); s["dims"][d]["lengthscale"] += h
            return o.sgpr_elbo(s, X, y, Z, s2)
        check(g[int(d)], fd(f), rtol=1e-4)
    for _ in range(2):
        m, c = int(rng.integers(0, M)), int(rng.integers(0, ncol))
        def fz(h):
            Zs = Z.copy(); Zs[m, c] += h
            return o.sgpr_elbo(spec, X, y, Zs, s2)
        ref = fd(fz)
        assert abs(gz[m, c] - ref) <= 5e-5 * max(1.0, abs(ref)), f"seed {seed}: Z[{m},{c}] {gz[m, c]} vs {ref} (groups {groups}, depth {R}, {route})"

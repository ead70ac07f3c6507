"""Randomised Gram parity: many random kernels (dimension count, depth, sub-kernel types and measures, shared / separate
variances, lengthscales and base variances over several decades, ragged sizes) against a BRUTE-FORCE elementary-symmetric
combination of the oracle's per-dimension matrices -- independent of both the HIP recurrence and the reference's
Newton-Girard power sums (whose cancellation error the oracle inherits when the k_d differ by many orders of magnitude)."""
import itertools
import os

import numpy as np
import pytest

import cases
from oak import _capi
from oracle import oak_oracle as o

pytestmark = pytest.mark.gpu
SEED0 = int(os.environ.get("OAK_FUZZ_SEED0", "0"))      # shift every stream: OAK_FUZZ_SEED0=100000 pytest ... for a fresh sweep
KINDS = ("gaussian", "uniform", "mog", "none", "gauss2", "binary", "categorical")


def brute_force_K(spec, X, X2, diag=False):
    D, R = len(spec["dims"]), spec["max_interaction_depth"]
    ks = []
    for d, dim in enumerate(spec["dims"]):
        c = o.active_col(spec, d)
        ks.append(o.base_K_diag(X[:, [c]], dim) if diag else o.base_K(X[:, [c]], X2[:, [c]], dim))
    ov = spec["order_variances"]
    share = spec.get("share_var_across_orders", True)
    out = np.full_like(ks[0], ov[0], dtype=np.float64)
    for r in range(1, R + 1):
        e = np.zeros_like(ks[0], dtype=np.float64)
        for S in itertools.combinations(range(D), r):
            t = np.ones_like(ks[0], dtype=np.float64)
            for d in S:
                t = t * ks[d]
            e += t
        out = out + (ov[r] if share else 1.0) * e
    return out


@pytest.mark.parametrize("seed", range(40))
def test_random_kernels_against_brute_force(hip, seed):
    rng = np.random.default_rng(SEED0 + 1000 + seed)
    D = int(rng.integers(1, 9))
    R = int(rng.integers(0, min(D, 4) + 1))
    share = bool(rng.integers(0, 2))
    kinds = tuple(rng.choice(KINDS, size=D))
    spec = cases.random_spec(rng, D, R, kinds, share=share)
    for dim in spec["dims"]:
        if dim["type"] == "rbf":
            dim["lengthscale"] = float(10 ** rng.uniform(-1.0, 1.3))
        if not share:
            dim["variance"] = float(10 ** rng.uniform(-3, 3))
    n1, n2 = int(rng.integers(1, 200)), int(rng.integers(1, 150))
    X, X2 = cases.random_inputs(rng, spec, n1), cases.random_inputs(rng, spec, n2)
    d = _capi.KernelDesc(spec)
    for got, ref in ((hip.gram(d, X, X2), brute_force_K(spec, X, X2)), (hip.gram(d, X), brute_force_K(spec, X, X)),
                     (hip.gram_diag(d, X), brute_force_K(spec, X, None, diag=True))):
        scale = max(np.abs(ref).max(), 1e-300)
        err = np.abs(got - ref).max() / scale
        assert err <= 5e-12, f"seed {seed}: D={D} R={R} share={share} kinds={kinds} err={err:.2e}"


@pytest.mark.parametrize("seed", range(16))
def test_random_kernels_gradient_against_forward_differences(hip, seed):
    """Analytic ELBO gradient (both backward kernels, every sub-kernel type, shared and separate variances) against central
    differences of the HIP forward pass itself, parameter by parameter.  Relative steps of 1e-4: the forward value carries
    ~1e-13 |F| of rounding noise, which a 1e-6 step would turn into 1e-5-level noise in the difference quotient."""
    import copy
    rng = np.random.default_rng(SEED0 + 5000 + seed)
    D = int(rng.integers(2, 8))
    R = int(rng.integers(1, min(D, 5) + 1))          # depth 5 exercises the generic two-pass backward kernel
    share = bool(rng.integers(0, 2))
    kinds = tuple(rng.choice(KINDS, size=D))
    spec = cases.random_spec(rng, D, R, kinds, share=share)
    N, M = 240, 18
    X = cases.random_inputs(rng, spec, N)
    Z = cases.random_inputs(rng, spec, M)
    y = rng.standard_normal((N, 1))
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    s2 = 0.2
    e, g = hip.sgpr_elbo_grad(_capi.KernelDesc(spec), s2)

    def fwd(sp, noise=s2):
        return hip.sgpr_elbo(_capi.KernelDesc(sp), noise)

    def check(analytic, make, h, what=""):
        fd = (fwd(*make(+h)) - fwd(*make(-h))) / (2 * h)
        assert abs(analytic - fd) <= 2e-5 * max(1.0, abs(fd)), f"seed {seed} D={D} R={R} share={share} kinds={kinds} {what}: {analytic} vs {fd}"

    for d, dim in enumerate(spec["dims"]):
        if dim["type"] == "rbf":
            def mk(h, d=d):
                sp = copy.deepcopy(spec); sp["dims"][d]["lengthscale"] += h
                return (sp,)
            check(g[d], mk, 1e-4 * dim["lengthscale"], f"lengthscale[{d}]")
        if not share:
            def mkv(h, d=d):
                sp = copy.deepcopy(spec); sp["dims"][d]["variance"] += h
                return (sp,)
            check(g[D + d], mkv, 1e-4 * dim["variance"], f"variance[{d}]")
    n_ov = len(spec["order_variances"])
    for r in range(n_ov):
        def mko(h, r=r):
            sp = copy.deepcopy(spec); sp["order_variances"][r] += h
            return (sp,)
        check(g[2 * D + r], mko, 1e-4, f"order_variance[{r}]")
    check(g[2 * D + n_ov], lambda h: (spec, s2 + h), 1e-5, "noise")


@pytest.mark.parametrize("seed", range(10))
def test_random_kernels_sobol_and_components(hip, seed):
    """Sobol indices and per-component predictions of random kernels over the measure types that have a closed form
    (Gaussian N(0,1), binary, categorical), shared and separate variances, against the oracle (1e-9)."""
    rng = np.random.default_rng(SEED0 + 9000 + seed)
    D = int(rng.integers(2, 7))
    R = int(rng.integers(1, min(D, 3) + 1))
    share = bool(rng.integers(0, 2))
    kinds = tuple(rng.choice(("gaussian", "binary", "categorical"), size=D))
    spec = cases.random_spec(rng, D, R, kinds, share=share)
    M = int(rng.integers(8, 40))
    Z = cases.random_inputs(rng, spec, M)
    alpha = rng.standard_normal((M, 1))
    subsets, ref = o.compute_sobol_oak(spec, Z, alpha, share_var_across_orders=share)
    got = hip.sobol(_capi.KernelDesc(spec), Z, alpha[:, 0], subsets, use_order_var=share)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * max(np.abs(ref).max(), 1e-300))
    Xs = cases.random_inputs(rng, spec, 30)
    comp = hip.component_predict(_capi.KernelDesc(spec), Xs, Z, alpha[:, 0], subsets, use_order_var=share)
    refc = np.array(o.prediction_components(spec, Z, alpha, Xs, share_var_across_orders=share))
    np.testing.assert_allclose(comp, refc, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("seed", range(12))
def test_random_subset_lists_through_both_sobol_evaluations(hip, seed):
    """Random term lists -- any order, any mix of sizes 1..6, occasionally a size-7 subset or a repeated dim (which must fall back
    to the per-term kernel) -- through the Gram of products and the per-term kernel: each against the oracle (1e-9 of the largest
    term), the column budget and the direct order-1 terms included whenever the plan takes them."""
    rng = np.random.default_rng(SEED0 + 12000 + seed)
    D = int(rng.integers(3, 11))
    depth = int(rng.integers(1, 5))
    share = bool(rng.integers(0, 2))
    kinds = tuple(rng.choice(("gaussian", "binary", "categorical", "gauss2"), size=D))
    spec = cases.random_spec(rng, D, depth, kinds, share=share)
    n = int(rng.integers(5, 200))
    Z = cases.random_inputs(rng, spec, n)
    alpha = rng.standard_normal(n) * rng.uniform(0.05, 3.0, n)
    if seed % 4 == 0:
        alpha = np.abs(alpha)                                   # one sign set empty
    max_len = min(D, depth if share else 6)                     # with shared variances a term's order needs an order variance
    subsets = []
    for _ in range(int(rng.integers(1, 400))):
        k = int(rng.integers(1, max_len + 1))
        subsets.append([int(v) for v in rng.permutation(D)[:k]])
    d = _capi.KernelDesc(spec)
    ref = np.array(o.compute_sobol_oak(spec, Z, alpha.reshape(-1, 1), share_var_across_orders=share, subsets=subsets, L_cache={})[1])
    scale = max(np.abs(ref).max(), 1e-300)
    try:
        for path in ("gram", "terms", "auto"):
            hip.sobol_set_path(path)
            got = hip.sobol(d, Z, alpha, subsets, use_order_var=share)
            np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-10 * scale, err_msg=path)
        if not share and D >= 7:
            big = [int(v) for v in rng.permutation(D)[:7]]
            hip.sobol_set_path("auto")
            got = hip.sobol(d, Z, alpha, subsets + [big], use_order_var=False)
            assert hip.sobol_last_info()["path"] == "terms"
            ref_big = o.compute_sobol_oak(spec, Z, alpha.reshape(-1, 1), share_var_across_orders=False, subsets=[big], L_cache={})[1][0]
            np.testing.assert_allclose(got[-1], ref_big, rtol=1e-8, atol=1e-10 * scale)
    finally:
        hip.sobol_set_path("auto")


@pytest.mark.parametrize("seed", range(8))
def test_random_multi_output_problems(seed):
    """Random P-column targets, kernels, routes, panel chunkings and inducing counts (multiples of 32 and not: the right-hand sides
    ride through chol(B) or take the explicit solves): bound and alpha against the oracle's N x P formulas, gradient against the
    sum of the single-output gradients."""
    rng = np.random.default_rng(SEED0 + 15000 + seed)
    D = int(rng.integers(2, 7))
    R = int(rng.integers(1, min(D, 3) + 1))
    kinds = tuple(rng.choice(("gaussian", "binary", "categorical", "uniform"), size=D))
    spec = cases.random_spec(rng, D, R, kinds, share=bool(rng.integers(0, 2)))
    N = int(rng.integers(300, 2500))
    M = int(rng.choice([24, 32, 50, 64, 96, 128, 130]))
    P = int(rng.integers(2, 12))
    X, Z = cases.random_inputs(rng, spec, N), cases.random_inputs(rng, spec, M)
    Y = rng.standard_normal((N, P)) + np.sin(X[:, :1])
    s2 = float(rng.uniform(0.05, 0.5))
    route = ("phi", "whitened", "auto")[seed % 3]
    d = _capi.KernelDesc(spec)
    ctx = _capi.HipContext(0)
    try:
        ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route(route)
        ctx.sgpr_set_panel_rows(0 if seed % 2 else int(rng.integers(64, N)))
        g_sum, e_sum = 0.0, 0.0
        for p in range(P):
            ctx.sgpr_set_data(X, Y[:, p])
            e, g = ctx.sgpr_elbo_grad(d, s2)
            e_sum, g_sum = e_sum + e, g_sum + g
        ctx.sgpr_set_data(X, Y[:, 0]); ctx.sgpr_set_extra_targets(Y[:, 1:])
        e, g = ctx.sgpr_elbo_grad(d, s2)
        er = o.sgpr_elbo(spec, X, Y, Z, s2)
        # against the oracle: the phi route's error grows with cond(Kuu) (random inducing inputs over discrete columns repeat
        # rows; the single-output tests allow it 1e-8 too); against the P single-output evaluations of the same route: 1e-10
        assert abs(e - er) <= (1e-8 if route == "phi" else 1e-9) * abs(er) and abs(e - e_sum) <= 1e-10 * abs(e_sum), (e, er, e_sum)
        np.testing.assert_allclose(g, g_sum, rtol=1e-7, atol=1e-8 * np.abs(g_sum).max())
        # posterior of single outputs: the predictive mean (alpha itself is cond(Kuu)-sensitive with random inducing inputs: its
        # agreement with the oracle is no better for ONE output)
        Xs = cases.random_inputs(rng, spec, 40)
        mr, vr = o.sgpr_predict_f(spec, X, Y, Z, s2, Xs)
        for p in (P - 1, 0):
            ctx.sgpr_select_output(p)
            mean, var = ctx.sgpr_predict(d, Xs)
            np.testing.assert_allclose(mean, mr[:, p], rtol=1e-6, atol=1e-7 * np.abs(mr).max())
            np.testing.assert_allclose(var, vr[:, p], rtol=1e-6, atol=1e-8)
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_svgp_problems(hip, seed):
    """SVGP ELBO, predictions and the q_mu / q_sqrt gradients of random kernels and variational parameters against the
    oracle: ELBO 1e-10, mean / var / log density 1e-9, gradients 1e-6 of their largest entry vs central differences along
    eight random directions."""
    from oracle import svgp_oracle as sv
    rng = np.random.default_rng(SEED0 + 13000 + seed)
    D = int(rng.integers(1, 7))
    R = int(rng.integers(1, min(D, 4) + 1))
    kinds = tuple(rng.choice(("gaussian", "uniform", "mog", "binary", "categorical"), size=D))
    spec = cases.random_spec(rng, D, R, kinds, share=True)
    N, M = int(rng.integers(20, 600)), int(rng.integers(4, 90))
    X, Z = cases.random_inputs(rng, spec, N), cases.random_inputs(rng, spec, M)
    y = (rng.uniform(size=N) < 0.5).astype(float)
    q_mu, q_sqrt = rng.standard_normal(M) * 10 ** rng.uniform(-1, 0.3), 10 ** rng.uniform(-1.5, 0.3, M)
    link = ("logit", "probit")[seed % 2]
    d = _capi.KernelDesc(spec)
    ctx = _capi.HipContext(0)
    try:
        ctx.sgpr_set_data(X, y.reshape(-1, 1)); ctx.sgpr_set_inducing(Z)
        e, g, gm, gs = ctx.svgp_elbo(d, q_mu, q_sqrt, link=link, grad=True)
        er = sv.svgp_elbo(spec, X, y, Z, q_mu, q_sqrt, link=link)
        assert abs(e - er) <= 1e-10 * abs(er), f"seed {seed}: {e} vs {er}"
        Xs = cases.random_inputs(rng, spec, 40); ys = (rng.uniform(size=40) < 0.5).astype(float)
        m, v, ld = ctx.svgp_predict(d, q_mu, q_sqrt, Xs, ys, link=link)
        mr, vr = sv.conditional(spec, Xs, Z, q_mu, q_sqrt)
        np.testing.assert_allclose(m, mr, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(v, vr, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(ld, sv.predict_log_density_from_f(mr, vr, ys, link), rtol=1e-9, atol=1e-9)
        scale = max(np.abs(gm).max(), np.abs(gs).max())
        for _ in range(8):
            dm, ds = rng.standard_normal(M), rng.standard_normal(M) * q_sqrt
            nrm = np.sqrt(dm @ dm + ds @ ds); dm, ds = dm / nrm, ds / nrm
            h = 1e-4 * min(1.0, q_sqrt.min() / np.abs(ds).max() * 0.1)
            f = lambda t: sv.svgp_elbo(spec, X, y, Z, q_mu + t * dm, q_sqrt + t * ds, link=link)
            fd = (8 * (f(h) - f(-h)) - (f(2 * h) - f(-2 * h))) / (12 * h)
            an = gm @ dm + gs @ ds
            assert abs(an - fd) <= 1e-6 * scale * np.sqrt(2 * M), f"seed {seed}: {an} vs {fd} (scale {scale})"
    finally:
        ctx.close()

"""GPU: BASELINE.json's full-size configurations -- the headline N = 1 048 576, D = 16, M = 1024, order 2 AND config 3 as
stated (order 3: a different instantiation of the fused Gram / backward kernels) -- checked through size-independent
properties, since the CPU oracle cannot finish that size in seconds:
  * row-shard additivity of the sufficient statistics (the multi-GPU contract) and exact symmetry of Phi;
  * bitwise determinism of repeated evaluations;
  * a 1/16 row sample against the multi-core oracle at the full M, both routes and -- on the phi route -- both accumulations of Phi
    (the default exact int8 / CRT one and the fp64 MFMA SYRK): the total (<= 1e-10) AND every
    kernel-dependent term of the bound on its own (sum log diag LB, c^T c, tr AAT, kappa, log det Kuu; <= 1e-10 relative each:
    at this size the total is dominated by the data-only terms, so a bound on the total alone would tolerate ~1e-2 absolute
    error in the kernel-dependent ones);
  * a directional finite difference of the HIP forward against the HIP analytic gradient;
  * the explicit Gram panel against the oracle on sampled rows (<= 1e-12 of max|K|: the kernel changes sign, so the bound is
    scaled by the largest entry rather than entry by entry).
"""
import numpy as np
import pytest

import cases
from oak import _capi
from oracle import c_oracle, oak_oracle as o

pytestmark = pytest.mark.gpu
N, D, M = 1 << 20, 16, 1024
ORDERS = [2, 3]


@pytest.fixture(scope="module")
def data():
    return o.synthetic_problem(N, D, M)


@pytest.fixture(params=ORDERS, ids=lambda r: f"order{r}")
def problem(request, data):
    X, y, Z = data
    return X, y, Z, o.make_spec(D, request.param), request.param


def test_fullsize_statistics_additivity_symmetry_determinism(problem):
    X, y, Z, spec, R = problem
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    ctx.sgpr_set_data(X, y)
    ctx.sgpr_local_stats(d)
    full = ctx.sgpr_get_stats()
    assert ctx.sgpr_stats_precision() == "int8crt"              # the default at this size: Phi accumulated exactly on the int8 pipe
    e1 = ctx.sgpr_elbo(d, 0.01)
    t1 = ctx.sgpr_last_terms()
    assert ctx.sgpr_elbo(d, 0.01) == e1 and ctx.sgpr_last_terms() == t1      # bitwise repeatable, term by term
    Phi = full[:M * M].reshape(M, M)
    np.testing.assert_array_equal(Phi, Phi.T)
    assert full[M * M + M + 2] == N and full[-2] == 0 and full[-1] == 1
    acc = np.zeros_like(full)
    cuts = [0, 300_001, 700_000, N]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ctx.sgpr_set_data(X[lo:hi], y[lo:hi])
        ctx.sgpr_local_stats(d)
        acc += ctx.sgpr_get_stats()
    np.testing.assert_allclose(acc[:-2], full[:-2], rtol=1e-12, atol=1e-12 * np.abs(full).max())
    assert acc[-1] == 3 and acc[-2] == 0
    ctx.sgpr_set_stats(acc, False)
    e2, terms = ctx.sgpr_tail(d, 0.01)
    assert abs(e2 - e1) <= 1e-11 * abs(e1) and terms[5] == N
    cases.assert_terms_match(ctx.sgpr_last_terms(), t1, rtol=1e-11, what=f"order {R}, shard sum vs one pass:")
    ctx.close()


def test_fullsize_sample_against_multicore_oracle(problem):
    X, y, Z, spec, R = problem
    ns = N // 16
    ref, parts = c_oracle.sgpr_elbo_chunked(spec, X[:ns], y[:ns], Z, 0.01, chunk=16384, return_parts=True)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X[:ns], y[:ns]); ctx.sgpr_set_inducing(Z)
    # phi route with both accumulations of Phi (the default at this size: exact int8 / CRT, csrc/crt.hip; and the fp64 MFMA SYRK),
    # whitened route (fp64 kernels): all three at the same 1e-10 on the total and on every term
    for route, precision, expect in (("phi", "auto", "int8crt"), ("phi", "fp64", "fp64"), ("whitened", "auto", "fp64")):
        ctx.sgpr_set_route(route); ctx.sgpr_set_precision(precision)
        e = ctx.sgpr_elbo(d, 0.01)
        assert ctx.sgpr_stats_precision() == expect, (route, precision, ctx.sgpr_stats_precision())
        assert abs(e - ref) <= 1e-10 * abs(ref), (route, precision, e, ref)
        cases.assert_terms_match(ctx.sgpr_last_terms(), parts["terms"], rtol=1e-10, what=f"order {R}, route {route}, precision {precision}:")
    ctx.sgpr_set_precision("auto")
    # explicit Gram on sampled rows
    rows = np.random.default_rng(0).choice(N, 2048, replace=False)
    K = ctx.gram(d, X[rows], Z)
    Kr = c_oracle.gram(spec, X[rows], Z)
    assert np.abs(K - Kr).max() <= 1e-12 * np.abs(Kr).max()
    ctx.close()


def test_fullsize_gradient_directional_check(problem):
    """<grad, v> from the analytic backward pass vs a central difference of the HIP forward along a random direction."""
    import copy
    X, y, Z, spec, R = problem
    ctx = _capi.HipContext(0)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    e, g = ctx.sgpr_elbo_grad(_capi.KernelDesc(spec), 0.01)
    # at this size the default forms the adjoint panel Kfu H on the int8 pipe (csrc/crt_gemm.hip): held against the fp64 GEMM on the same statistics
    info = ctx.bench_crt_info()
    assert ctx.sgpr_stats_precision() == "int8crt" and info["gemm_planes"] >= 13 and info["gemm_bits"] >= 44, info
    import os
    try:
        os.environ["OAK_CRT_GEMM"] = "0"
        e0, g0 = ctx.sgpr_elbo_grad(_capi.KernelDesc(spec), 0.01)
    finally:
        os.environ.pop("OAK_CRT_GEMM", None)
    assert ctx.bench_crt_info()["gemm_planes"] == 0 and e0 == e
    np.testing.assert_allclose(g, g0, rtol=0, atol=1e-12 * np.abs(g0).max())
    rng = np.random.default_rng(1)
    v_ls, v_ov, v_n = rng.uniform(-1, 1, D), rng.uniform(-1, 1, R + 1), 0.01 * rng.uniform(-1, 1)

    def forward(h):
        s = copy.deepcopy(spec)
        for i in range(D):
            s["dims"][i]["lengthscale"] += h * v_ls[i]
        s["order_variances"] = [s["order_variances"][r] + h * v_ov[r] for r in range(R + 1)]
        return ctx.sgpr_elbo(_capi.KernelDesc(s), 0.01 + h * v_n)

    h = 1e-5
    fd = (forward(h) - forward(-h)) / (2 * h)
    analytic = g[:D] @ v_ls + g[2 * D:2 * D + R + 1] @ v_ov + g[2 * D + R + 1] * v_n
    np.testing.assert_allclose(analytic, fd, rtol=1e-5)
    ctx.close()


def test_fullsize_prediction_properties(data):
    """predict_f of 2^18 rows at the full model: the mean equals K(X*, Z) alpha with alpha from oak_sgpr_alpha (the identity
    the reference's tests/test_utils.py:42-75 pins), the variance lies in (0, K_diag(X*)], the whole batch (blocked TRSM)
    agrees with 4096-row batches (substitution leaf), and whitened and phi posteriors agree."""
    X, y, Z = data
    spec = o.make_spec(D, 2)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    ctx.sgpr_elbo(d, 0.01)
    rng = np.random.default_rng(17)
    Xs = rng.standard_normal((1 << 18, D))
    mean, var = ctx.sgpr_predict(d, Xs)
    alpha = ctx.sgpr_alpha(M)
    idx = rng.choice(Xs.shape[0], 4096, replace=False)
    Ks = ctx.gram(d, Xs[idx], Z)
    np.testing.assert_allclose(mean[idx], Ks @ alpha, rtol=1e-8, atol=1e-9 * np.abs(mean).max())
    kdiag = ctx.gram_diag(d, Xs[idx])
    assert var.min() > 0 and (var[idx] <= kdiag * (1 + 1e-12)).all()
    m2, v2 = ctx.sgpr_predict(d, Xs[idx])                      # small batch: substitution path
    np.testing.assert_allclose(m2, mean[idx], rtol=1e-9, atol=1e-10 * np.abs(mean).max())
    np.testing.assert_allclose(v2, var[idx], rtol=1e-9, atol=1e-10 * np.abs(var).max())
    ctx.sgpr_set_route("whitened")
    ctx.sgpr_elbo(d, 0.01)
    m3, v3 = ctx.sgpr_predict(d, Xs[idx])
    np.testing.assert_allclose(m3, mean[idx], rtol=1e-8, atol=1e-9 * np.abs(mean).max())
    np.testing.assert_allclose(v3, var[idx], rtol=1e-8, atol=1e-9 * np.abs(var).max())
    ctx.close()


def test_c5_size_mixed_kernel_properties():
    """BASELINE.json config 5 at full size (N = 262 144, D = 32 mixed: 20 RBF + 8 binary + 4 categorical, M = 2048, depth 4):
    row-shard additivity, a 16 384-row sample against the multicore oracle at the full M (1e-10, literal route), a
    directional difference of the forward against the analytic gradient (D <= 32 fast backward kernel, one column per lane),
    and the Sobol indices of all 41 448 terms: the Gram-of-products evaluation against the per-term kernel (all terms) and
    against the oracle (320 sampled terms of every order and first-factor type) at this size."""
    import bench
    N5, D5, M5, R5 = 262144, 32, 2048, 4
    X, y, Z = bench.synthetic(N5, D5, M5, mixed=True)
    spec = bench.make_spec(D5, R5, mixed=True)
    d = _capi.KernelDesc(spec)
    ctx = _capi.HipContext(0)
    ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    ctx.sgpr_set_data(X, y)
    ctx.sgpr_local_stats(d)
    full = ctx.sgpr_get_stats()
    assert ctx.sgpr_stats_precision() == "int8crt"              # mixed kernel at depth 4: the fused residue epilogue's CPT = 2 form
    ctx.sgpr_set_precision("fp64"); ctx.sgpr_local_stats(d); full64 = ctx.sgpr_get_stats(); ctx.sgpr_set_precision("auto")
    P, P64 = full[:M5 * M5].reshape(M5, M5), full64[:M5 * M5].reshape(M5, M5)
    dg = np.sqrt(np.outer(np.diag(P64), np.diag(P64)))
    assert (np.abs(P - P64) / dg).max() <= 1e-13               # the two accumulations of Phi against each other (measured 8e-15)
    acc = np.zeros_like(full)
    for lo, hi in ((0, 100_003), (100_003, N5)):
        ctx.sgpr_set_data(X[lo:hi], y[lo:hi]); ctx.sgpr_local_stats(d); acc += ctx.sgpr_get_stats()
    np.testing.assert_allclose(acc[:-2], full[:-2], rtol=1e-12, atol=1e-12 * np.abs(full).max())
    ns = 16384
    ctx.sgpr_set_data(X[:ns], y[:ns]); ctx.sgpr_set_route("whitened")
    e = ctx.sgpr_elbo(d, 0.01)
    er, parts = c_oracle.sgpr_elbo_chunked(spec, X[:ns], y[:ns], Z, 0.01, 1e-6, chunk=4096, return_parts=True)
    assert abs(e - er) <= 1e-10 * abs(er)
    cases.assert_terms_match(ctx.sgpr_last_terms(), parts["terms"], rtol=1e-10, what="C5 sample, whitened route:")
    # directional derivative at full size (phi route)
    import copy
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_route("phi")
    e0, g = ctx.sgpr_elbo_grad(d, 0.01)
    assert ctx.bench_crt_info()["gemm_planes"] >= 13                     # int8 adjoint GEMM at M = 2048 (two 1024-term folds per plane)
    rng = np.random.default_rng(1)
    v_ls = rng.uniform(-1, 1, D5) * np.array([dm["type"] == "rbf" for dm in spec["dims"]])
    v_ov = rng.uniform(-1, 1, R5 + 1)

    def forward(h):
        s = copy.deepcopy(spec)
        for i in range(D5):
            if s["dims"][i]["type"] == "rbf":
                s["dims"][i]["lengthscale"] += h * v_ls[i]
        s["order_variances"] = [s["order_variances"][r] + h * v_ov[r] for r in range(R5 + 1)]
        return ctx.sgpr_elbo(_capi.KernelDesc(s), 0.01)

    h = 1e-5
    fd = (forward(h) - forward(-h)) / (2 * h)
    an = float(g[:D5] @ v_ls + g[2 * D5:2 * D5 + R5 + 1] @ v_ov)
    assert abs(an - fd) <= 1e-5 * max(1.0, abs(fd)), (an, fd)
    # Sobol over every term of depth <= 4
    ctx.sgpr_elbo(d, 0.01)
    alpha = ctx.sgpr_alpha(M5)
    subsets = o.list_representation(D5, R5)[1:]
    assert len(subsets) == 41448
    sob = ctx.sobol(d, Z, alpha, subsets)
    info = ctx.sobol_last_info()
    assert info["path"] == "gram" and info["columns"] == 512     # 527 canonical columns trimmed to four 128-tiles (sobol_make_plan_budgeted) and info["pair_rows"] == M5 * (M5 + 1) // 2
    assert np.isfinite(sob).all() and (sob >= -1e-12 * sob.max()).all()
    # each order-4 term is read from the Gram matrix under its three pairings ab|cd, ac|bd, ad|bc: they agree
    assert 0.0 < info["pairing_disagreement"] < 1e-11
    # ... against the independent per-term kernel (a fused product-reduction over the stacked L_d), every one of the 41 448 terms
    ctx.sobol_set_path("terms")
    sob_terms = ctx.sobol(d, Z, alpha, subsets)
    ctx.sobol_set_path("auto")
    np.testing.assert_allclose(sob, sob_terms, rtol=1e-8, atol=1e-12 * sob.max())
    np.testing.assert_allclose(sob / sob.sum(), sob_terms / sob_terms.sum(), atol=1e-10)
    # ... and against the ORACLE at this size on 320 sampled terms: every order, and among the order-3 / order-4 samples every
    # sub-kernel type (RBF / binary / categorical) as first factor (the factor that carries the order variance, utils.py:376-380;
    # binary first factors enter with v instead of v^2, :266)
    rng = np.random.default_rng(11)
    types = [dm["type"] for dm in spec["dims"]]
    by_order = {k: [i for i, S in enumerate(subsets) if len(S) == k] for k in (1, 2, 3, 4)}
    pick = list(by_order[1])
    pick += list(rng.choice(by_order[2], 64, replace=False))
    for k in (3, 4):
        for t in ("rbf", "binary", "categorical"):
            cand = [i for i in by_order[k] if types[subsets[i][0]] == t]
            assert cand, (k, t)
            pick += list(rng.choice(cand, min(45, len(cand)), replace=False))
    pick = sorted(set(int(i) for i in pick))
    assert len(pick) >= 256
    chosen = [subsets[i] for i in pick]
    _, ref = o.compute_sobol_oak(spec, Z, alpha.reshape(-1, 1), subsets=chosen, L_cache={})
    ref = np.array(ref)
    total = sob.sum()
    np.testing.assert_allclose(sob[pick] / total, ref / total, atol=1e-9)                 # the normalised indices
    np.testing.assert_allclose(sob[pick], ref, rtol=1e-7, atol=1e-11 * np.abs(ref).max())  # and the raw values
    ctx.close()

"""GPU: Phi = Kuf Kuf^T accumulated EXACTLY on the int8 matrix pipe (oak_sgpr_set_precision("int8crt"), csrc/crt.hip; the default
"auto" mode picks it on large phi-route problems).  What the mode changes is HOW the sum over N rows is formed (scaled 48-bit
integers, residue planes, int8 MFMA, Chinese remainder reconstruction) -- the fp64 Gram entries, psi, kappa and the whole tail are the
fp64 kernels' -- so it is held to the fp64 contract, not to a looser one:
  * Phi against an extended-precision (80-bit accumulation) reference: the int8 route's error -- ONE rounding per panel entry, to 2^-B
    of the column's bound, B = 48 .. 50 -- stays below 4e-14 of sqrt(Phi_aa Phi_bb) (measured 2e-16 .. 9e-16 on these problems, the
    fp64 MFMA accumulation 3e-16 .. 4e-16; at the headline size both are ~1e-15), i.e. four orders inside the 1e-10 contract of the
    bound's terms;
  * the ELBO and every kernel-dependent term against the oracle at <= 1e-10 (the fp64 tolerance), all sub-kernel types;
  * bit-exact identities that follow from integer arithmetic: fused Gram epilogue == stand-alone conversion pass, LDS-DMA SYRK ==
    register-staged SYRK, one panel chunk == many chunks (residues carried between chunks), repeated evaluation == itself;
  * row-shard additivity, ragged shapes (N not a multiple of 16 / 128, M not a multiple of 256), extra output columns, gradient calls,
    the whitened route and too-small problems (the mode steps aside and says so), a description with a negative order variance (the
    a-priori bound must not rely on positive semi-definiteness).
"""
import os

import numpy as np
import pytest

import cases
from oak import _capi
from oracle import c_oracle, oak_oracle as o

pytestmark = pytest.mark.gpu


def _problem(N, D, M, R, kinds, seed=0, ls=(0.8, 1.6)):
    rng = np.random.default_rng(1000 * seed + N + D)
    spec = cases.random_spec(rng, D, R, kinds)
    for dim in spec["dims"]:
        if dim["type"] == "rbf":
            dim["lengthscale"] = float(rng.uniform(*ls))
    X = cases.random_inputs(rng, spec, N)
    Z = X[rng.choice(N, M, replace=False)].copy()
    y = (np.sin(X[:, 0]) + 0.3 * X[:, 1 % D] + 0.1 * rng.standard_normal(N)).reshape(-1, 1)
    y = (y - y.mean()) / y.std()
    return spec, X, y, Z


def _stats(ctx, d, mode, s2=0.05):
    ctx.sgpr_set_precision(mode)
    e = ctx.sgpr_elbo(d, s2)
    return e, ctx.sgpr_last_terms(), ctx.sgpr_get_stats(), ctx.sgpr_stats_precision()


@pytest.mark.parametrize("N,D,M,R,kinds", [(65536, 8, 512, 2, ("gaussian",)),
                                           (40000, 20, 384, 4, ("gaussian", "binary", "categorical", "uniform")),
                                           (50001, 6, 300, 3, ("gaussian", "mog"))])
def test_phi_against_an_extended_precision_accumulation(N, D, M, R, kinds):
    """Sampled entries of Phi against sums formed in 80-bit arithmetic from the device's own fp64 Gram panel."""
    spec, X, y, Z = _problem(N, D, M, R, kinds)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    e64, t64, s64, u64 = _stats(ctx, d, "fp64")
    ec, tc, sc, uc = _stats(ctx, d, "int8crt")
    assert u64 == "fp64" and uc == "int8crt"
    P64, Pc = s64[:M * M].reshape(M, M), sc[:M * M].reshape(M, M)
    np.testing.assert_array_equal(Pc, Pc.T)
    np.testing.assert_array_equal(sc[M * M + M:], s64[M * M + M:])              # kappa, yy, counts: the fp64 kernels' own
    np.testing.assert_allclose(sc[M * M:M * M + M], s64[M * M:M * M + M], rtol=1e-13, atol=1e-13 * np.abs(s64[M * M:M * M + M]).max())   # psi: same entries, another summation tree
    cols = np.unique(np.random.default_rng(5).choice(M, 24, replace=False))
    K = np.empty((N, len(cols)), dtype=np.longdouble)
    for lo in range(0, N, 8192):
        K[lo:lo + 8192] = ctx.gram(d, X[lo:lo + 8192], Z[cols])
    ref = (K.T @ K).astype(np.longdouble)                                        # 64-bit mantissa accumulation
    dg = np.sqrt(np.outer(np.diag(ref), np.diag(ref))).astype(np.float64)
    sub = np.ix_(cols, cols)
    err_c = np.abs((Pc[sub].astype(np.longdouble) - ref).astype(np.float64)) / dg
    err_64 = np.abs((P64[sub].astype(np.longdouble) - ref).astype(np.float64)) / dg
    # one rounding of 2^-48 of the column bound per entry: relative to sqrt(Phi_aa Phi_bb) that is 2^-47 (bound / rms) / sqrt(6 N), with
    # the Cauchy-Schwarz bound a few hundred times the column's rms for these kernels
    assert err_c.max() <= 4e-14, err_c.max()
    assert err_64.max() <= 4e-15, err_64.max()                   # (the fp64 MFMA accumulation, for the record)
    print(f"Phi error vs 80-bit accumulation: int8crt {err_c.max():.2e}, fp64 {err_64.max():.2e}")
    # (two Phi that differ by 1e-15 relative reach the bound through W = L^-1 Phi L^-T, i.e. amplified by cond(Kuu): the routes are
    # compared with each other at 1e-9 here and with the ORACLE at the contract's 1e-10 below)
    assert abs(ec - e64) <= 1e-9 * abs(e64)
    ctx.close()


@pytest.mark.parametrize("case", ["c2", "mixed", "ragged"])
def test_bound_and_terms_against_the_oracle(case):
    if case == "c2":                                             # BASELINE config 2 at full size (cond(Kuu) ~ 1e5: the phi route's hard case)
        X, y, Z = o.synthetic_problem(65536, 8, 512)
        spec = o.make_spec(8, 2)
    elif case == "mixed":
        spec, X, y, Z = _problem(30000, 18, 256, 4, ("gaussian", "binary", "categorical", "mog", "uniform"), seed=3)
    else:
        spec, X, y, Z = _problem(20011, 7, 130, 3, ("gaussian", "gauss2"), seed=4)
    ref, parts = c_oracle.sgpr_elbo_chunked(spec, X, y, Z, 0.02, chunk=8192, return_parts=True)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    e, t, s, used = _stats(ctx, d, "int8crt", 0.02)
    assert used == "int8crt"
    assert abs(e - ref) <= 1e-10 * abs(ref), (e, ref)
    cases.assert_terms_match(t, parts["terms"], rtol=1e-10, what=f"int8crt, {case}:")
    assert ctx.sgpr_elbo(d, 0.02) == e and ctx.sgpr_last_terms() == t          # bitwise repeatable
    ctx.close()


def test_integer_identities_between_the_code_paths():
    """Exact integer accumulation: every route to the residue planes and through the int8 SYRK gives the SAME Phi, bit for bit."""
    spec, X, y, Z = _problem(70000, 9, 520, 2, ("gaussian",), seed=7)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    _, _, base, used = _stats(ctx, d, "int8crt")
    assert used == "int8crt"
    M = Z.shape[0]
    try:
        os.environ["OAK_CRT_UNFUSED"] = "1"                      # stand-alone conversion of the fp64 panel instead of the Gram epilogue
        _, _, s1, _ = _stats(ctx, d, "int8crt")
        del os.environ["OAK_CRT_UNFUSED"]
        os.environ["OAK_CRT_SYRK"] = "4"                         # register-staged int8 SYRK instead of the LDS-DMA pipeline
        _, _, s2, _ = _stats(ctx, d, "int8crt")
        del os.environ["OAK_CRT_SYRK"]
    finally:
        os.environ.pop("OAK_CRT_UNFUSED", None); os.environ.pop("OAK_CRT_SYRK", None)
    np.testing.assert_array_equal(s1[:M * M], base[:M * M])
    np.testing.assert_array_equal(s2[:M * M], base[:M * M])
    ctx.sgpr_set_panel_rows(16384)                               # five panel chunks: residues carried from chunk to chunk
    _, _, s3, used3 = _stats(ctx, d, "int8crt")
    ctx.sgpr_set_panel_rows(0)
    assert used3 == "int8crt"
    np.testing.assert_array_equal(s3[:M * M], base[:M * M])
    ctx.close()


def test_row_shards_extra_outputs_gradient_and_fallbacks():
    spec, X, y, Z = _problem(48000, 10, 384, 3, ("gaussian", "binary"), seed=9)
    N, M = X.shape[0], Z.shape[0]
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi"); ctx.sgpr_set_precision("int8crt")
    ctx.sgpr_set_data(X, y)
    e_full = ctx.sgpr_elbo(d, 0.05); full = ctx.sgpr_get_stats(); t_full = ctx.sgpr_last_terms()
    assert ctx.sgpr_stats_precision() == "int8crt"
    acc = np.zeros_like(full)
    for lo, hi in ((0, 17001), (17001, 31000), (31000, N)):
        ctx.sgpr_set_data(X[lo:hi], y[lo:hi]); ctx.sgpr_local_stats(d); acc += ctx.sgpr_get_stats()
        assert ctx.sgpr_stats_precision() == "int8crt"
    np.testing.assert_allclose(acc[:-2], full[:-2], rtol=1e-12, atol=1e-12 * np.abs(full).max())
    # the fp64 kernels on the same data
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_precision("fp64")
    e64, g64 = ctx.sgpr_elbo_grad(d, 0.05)
    ctx.sgpr_set_precision("int8crt")
    eg, g = ctx.sgpr_elbo_grad(d, 0.05)                          # fused pass that also writes the fp64 panel the backward reads
    assert ctx.sgpr_stats_precision() == "int8crt"
    assert abs(eg - e64) <= 1e-9 * abs(e64) and eg == e_full
    np.testing.assert_allclose(g, g64, rtol=1e-9, atol=1e-9 * np.abs(g64).max())
    # extra output columns read the fp64 panel for their Kuf y
    ctx.sgpr_set_extra_targets(np.column_stack([np.cos(X[:, 1]), X[:, 2] ** 2]))
    e3 = ctx.sgpr_elbo(d, 0.05)
    assert ctx.sgpr_stats_precision() == "int8crt"
    ctx.sgpr_set_precision("fp64")
    e3_64 = ctx.sgpr_elbo(d, 0.05)
    assert abs(e3 - e3_64) <= 1e-9 * abs(e3_64) and e3_64 != e64
    ctx.sgpr_set_extra_targets(None)
    # whitened route: no a-priori bound on L^-1 Kuf -- the fp64 kernels run and the context says so
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_precision("int8crt"); ctx.sgpr_set_route("whitened")
    ew = ctx.sgpr_elbo(d, 0.05)
    assert ctx.sgpr_stats_precision() == "fp64" and abs(ew - e64) <= 1e-10 * abs(e64)
    # too few rows: steps aside as well
    ctx.sgpr_set_route("phi"); ctx.sgpr_set_data(X[:2000], y[:2000])
    ctx.sgpr_elbo(d, 0.05)
    assert ctx.sgpr_stats_precision() == "fp64"
    # default mode: 384 inducing points stay on the fp64 kernels (the automatic rule starts at M = 512)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_precision("auto")
    assert ctx.sgpr_elbo(d, 0.05) == e64 and ctx.sgpr_stats_precision() == "fp64"
    ctx.close()


def test_negative_order_variance_uses_the_entrywise_bound():
    """A description whose kernel is not positive semi-definite: Cauchy-Schwarz does not bound its entries; the scales fall back to
    sum_r |w_r| e_r(max k_d) and Phi still matches the fp64 accumulation."""
    spec, X, y, Z = _problem(30000, 6, 256, 2, ("gaussian",), seed=11)
    spec["order_variances"] = [0.3, -0.8, 1.1]
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    M = Z.shape[0]
    ctx.sgpr_set_precision("fp64"); ctx.sgpr_local_stats(d); s64 = ctx.sgpr_get_stats()
    ctx.sgpr_set_precision("int8crt"); ctx.sgpr_local_stats(d); sc = ctx.sgpr_get_stats()
    assert ctx.sgpr_stats_precision() == "int8crt"
    P64, Pc = s64[:M * M].reshape(M, M), sc[:M * M].reshape(M, M)
    dg = np.sqrt(np.outer(np.diag(P64), np.diag(P64)))
    assert (np.abs(Pc - P64) / dg).max() <= 1e-13
    ctx.close()


def test_deep_kernel_takes_the_stand_alone_conversion():
    """Depth 6 is outside the fused epilogue's instantiations (<= 4): the fp64 panel is converted by its own pass."""
    spec, X, y, Z = _problem(24000, 7, 256, 6, ("gaussian",), seed=13)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    e64, _, _, _ = _stats(ctx, d, "fp64")
    ec, _, _, used = _stats(ctx, d, "int8crt")
    assert used == "int8crt" and abs(ec - e64) <= 1e-9 * abs(e64)
    ctx.close()


def _ill_conditioned(N, M, D, ls, seed=0):
    """Near-duplicate inducing points and a long lengthscale: cond(Kuu + 1e-6 I) ~ 1e6 .. 1e8."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    y = (np.sin(X[:, 0]) + 0.5 * X[:, 1] * X[:, 2] + 0.1 * rng.standard_normal(N)).reshape(-1, 1)
    y = (y - y.mean()) / y.std()
    Z = X[:M].copy()
    Z[M // 2:] = Z[:M - M // 2] + 0.02 * rng.standard_normal((M - M // 2, D))
    return o.make_spec(D, 2, lengthscales=[ls] * D), X, y, Z


@pytest.mark.parametrize("ls", [1.5, 3.0])
def test_ill_conditioned_kuu_stays_on_the_phi_route_with_double_double_whitening(ls):
    """The automatic route used to send such evaluations through the N-sized triangular solve (GPflow's A = L^-1 Kuf), because the fp64
    products L^-1 Phi L^-T lose cond(Kuu) eps.  With the exact (double-double) Phi of the int8 route and double-double products
    (csrc/ddgemm.hip) the phi route meets the same 1e-10 -- on the total and on every term -- and is the one the auto route takes."""
    spec, X, y, Z = _ill_conditioned(65536, 768, 8, ls)
    ref, parts = c_oracle.sgpr_elbo_chunked(spec, X, y, Z, 0.01, chunk=8192, return_parts=True)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)                       # route auto, precision auto: the defaults
    e = ctx.sgpr_elbo(d, 0.01)
    t = ctx.sgpr_last_terms()
    info = ctx.bench_crt_info()
    assert not ctx.sgpr_stats_whitened() and ctx.sgpr_stats_precision() == "int8crt" and info["tail_dd"] == 1
    assert t["cond_estimate"] > 1e2
    # every term at 1e-10 of its own size; the total is a difference of terms a few hundred times its size here (0.5 c^T c = 3e6 against
    # |ELBO| = 1e4), so it is held to 1e-10 of the largest of them (and to 2e-10 |ELBO| against the whitened route below)
    cases.assert_terms_match(t, parts["terms"], rtol=1e-10, what=f"auto route (int8 Phi, double-double tail), ls {ls}:")
    scale = max(abs(ref), 0.5 * abs(parts["terms"]["cTc"]), 0.5 * abs(parts["terms"]["tr_AAT"]))
    assert abs(e - ref) <= 1e-10 * scale, (e, ref)
    assert ctx.sgpr_elbo(d, 0.01) == e                                          # bitwise repeatable
    # the same statistics through fp64 products: what the double-double tail is for
    try:
        os.environ["OAK_TAIL_DD"] = "0"
        e64 = ctx.sgpr_elbo(d, 0.01)
    finally:
        os.environ.pop("OAK_TAIL_DD", None)
    assert ctx.bench_crt_info()["tail_dd"] == 0
    assert abs(e64 - ref) > 20 * abs(e - ref) and abs(e64 - ref) > 1e-9 * abs(ref)
    # GPflow's own order, for comparison: the two routes agree
    ctx.sgpr_set_route("whitened")
    ew = ctx.sgpr_elbo(d, 0.01)
    assert ctx.sgpr_stats_whitened() and abs(ew - e) <= 2e-10 * abs(ref)
    # gradient of the auto route (phi statistics, fp64 backward) against the whitened route's
    _, gw = ctx.sgpr_elbo_grad(d, 0.01)
    ctx.sgpr_set_route("auto")
    eg, g = ctx.sgpr_elbo_grad(d, 0.01)
    assert not ctx.sgpr_stats_whitened() and abs(eg - e) <= 1e-12 * abs(e)
    np.testing.assert_allclose(g, gw, rtol=1e-5, atol=1e-6 * np.abs(gw).max())
    # predictions from the phi-route posterior against the whitened one
    m1, v1 = ctx.sgpr_predict(d, X[:2048])
    ctx.sgpr_set_route("whitened"); ctx.sgpr_elbo(d, 0.01)
    m2, v2 = ctx.sgpr_predict(d, X[:2048])
    np.testing.assert_allclose(m1, m2, rtol=1e-7, atol=1e-8 * np.abs(m2).max())
    np.testing.assert_allclose(v1, v2, rtol=1e-6, atol=1e-8 * np.abs(v2).max())
    ctx.close()


def test_double_double_whitening_with_several_output_columns():
    """Further output columns (oak_sgpr_set_extra_targets) ride through the double-double tail as further L^-1 psi_p rows: the
    P-column bound equals the sum of the P single-column bounds of the oracle."""
    spec, X, y, Z = _ill_conditioned(65536, 768, 8, 3.0, seed=3)
    Y = np.column_stack([y[:, 0], np.cos(X[:, 1]) + 0.3 * X[:, 4], np.tanh(X[:, 2] * X[:, 3])])
    refs = [c_oracle.sgpr_elbo_chunked(spec, X, Y[:, p], Z, 0.01, chunk=8192, return_parts=True) for p in range(3)]
    ref = sum(r[0] for r in refs)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, Y[:, 0]); ctx.sgpr_set_extra_targets(Y[:, 1:]); ctx.sgpr_set_inducing(Z)
    e = ctx.sgpr_elbo(d, 0.01)
    assert not ctx.sgpr_stats_whitened() and ctx.sgpr_stats_precision() == "int8crt" and ctx.bench_crt_info()["tail_dd"] == 1
    scale = max(abs(ref), sum(0.5 * abs(r[1]["terms"]["cTc"]) for r in refs), 1.5 * abs(refs[0][1]["terms"]["tr_AAT"]))
    assert abs(e - ref) <= 1e-10 * scale, (e, ref)
    ctx.sgpr_set_route("whitened")
    ew = ctx.sgpr_elbo(d, 0.01)
    assert ctx.sgpr_stats_whitened() and abs(ew - e) <= 2e-10 * abs(ref)
    # the single outputs' posteriors against the whitened route's
    Xs = X[:1024]
    pw = []
    for p in range(3):
        ctx.sgpr_select_output(p); pw.append(ctx.sgpr_predict(d, Xs))
    ctx.sgpr_set_route("auto"); ctx.sgpr_elbo(d, 0.01)
    assert ctx.bench_crt_info()["tail_dd"] == 1
    for p in range(3):
        ctx.sgpr_select_output(p)
        m1, v1 = ctx.sgpr_predict(d, Xs)
        np.testing.assert_allclose(m1, pw[p][0], rtol=1e-7, atol=1e-8 * np.abs(pw[p][0]).max())
        np.testing.assert_allclose(v1, pw[p][1], rtol=1e-6, atol=1e-8 * np.abs(pw[p][1]).max())
    ctx.close()


def test_well_conditioned_kuu_keeps_the_fp64_products():
    spec, X, y, Z = _problem(65536, 8, 768, 2, ("gaussian",), seed=21, ls=(0.6, 0.9))
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
    ctx.sgpr_elbo(d, 0.05)
    t = ctx.sgpr_last_terms()
    if t["cond_estimate"] <= 1e2:
        assert ctx.bench_crt_info()["tail_dd"] == 0 and ctx.sgpr_stats_precision() == "int8crt" and not ctx.sgpr_stats_whitened()
    ctx.close()


def _with_env(name, value, fn):
    old = os.environ.get(name)
    try:
        if value is None: os.environ.pop(name, None)
        else: os.environ[name] = value
        return fn()
    finally:
        if old is None: os.environ.pop(name, None)
        else: os.environ[name] = old


@pytest.mark.parametrize("N,D,M,kinds,ls", [(50001, 7, 640, ("gaussian", "uniform", "gaussian", "categorical"), (0.3, 0.5)),
                                              (65536, 8, 1024, ("gaussian",), (0.35, 0.6)),
                                              (196608, 8, 1024, ("gaussian",), (0.35, 0.6))])      # a row shard's size: the partitioned forward pass
def test_int8_adjoint_gemm_of_the_gradient(N, D, M, kinds, ls):
    """csrc/crt_gemm.hip: a gradient call whose forward pass took the int8 route forms the adjoint panel Kfu H from the residue planes
    (LDS transpose reads, one int8 GEMM per modulus, fp64 fraction reconstruction) when chol(Kuu) looks well-conditioned.  Its operands
    are 49-51-bit fixed-point numbers per column: the gradient is held to 1e-10 of its largest entry against the fp64 GEMM and against
    the fp64 kernels (measured 1e-12 .. 6e-12).  Ragged N and M (plane columns padded to 768), a discrete sub-kernel, further output
    columns, the gradient w.r.t. the inducing inputs.  (Short lengthscales: the estimate of these problems is 27 and 31.)  In a partitioned
    forward pass (small N, row shards) the estimate arrives behind the first Gram launch: the fp64 panel is written and the backward decides."""
    spec, X, y, Z = _problem(N, D, M, 2, kinds, seed=31, ls=ls)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi"); ctx.sgpr_set_precision("int8crt")
    e0, g0 = _with_env("OAK_CRT_GEMM", "0", lambda: ctx.sgpr_elbo_grad(d, 0.05))
    assert ctx.bench_crt_info()["gemm_planes"] == 0 and ctx.sgpr_stats_precision() == "int8crt"
    e1, g1 = ctx.sgpr_elbo_grad(d, 0.05)
    info = ctx.bench_crt_info()
    assert ctx.sgpr_last_terms()["cond_estimate"] <= 1e2 and info["gemm_planes"] >= 13 and info["gemm_bits"] >= 44, info
    assert e1 == e0                                               # the forward pass is the same arithmetic with or without the fp64 panel
    tol = 1e-10 * np.abs(g0).max()
    np.testing.assert_allclose(g1, g0, rtol=0, atol=tol)
    ctx.sgpr_set_precision("fp64")
    _, g64 = ctx.sgpr_elbo_grad(d, 0.05)
    assert ctx.bench_crt_info()["gemm_planes"] == 0
    np.testing.assert_allclose(g1, g64, rtol=0, atol=tol)
    ctx.sgpr_set_precision("int8crt")
    assert ctx.sgpr_elbo_grad(d, 0.05)[1].tobytes() == g1.tobytes()            # repeatable bit for bit
    # gradient w.r.t. the inducing inputs reads the same adjoint panel
    _, _, gz0 = _with_env("OAK_CRT_GEMM", "0", lambda: ctx.sgpr_elbo_grad_z(d, 0.05, M, D))
    _, _, gz1 = ctx.sgpr_elbo_grad_z(d, 0.05, M, D)
    assert ctx.bench_crt_info()["gemm_planes"] >= 13
    np.testing.assert_allclose(gz1, gz0, rtol=0, atol=1e-9 * np.abs(gz0).max())
    # further output columns: y_p a_p^T joins the int8 product in a pass of its own
    ctx.sgpr_set_extra_targets(np.column_stack([np.cos(X[:, 1]), X[:, 2] ** 2]))
    ex0, gx0 = _with_env("OAK_CRT_GEMM", "0", lambda: ctx.sgpr_elbo_grad(d, 0.05))
    ex1, gx1 = ctx.sgpr_elbo_grad(d, 0.05)
    assert ctx.bench_crt_info()["gemm_planes"] >= 13 and ex1 == ex0 and ex1 != e1
    np.testing.assert_allclose(gx1, gx0, rtol=0, atol=1e-10 * np.abs(gx0).max())
    ctx.sgpr_set_extra_targets(None)
    # several panel chunks: the planes of the last chunk only -- the fp64 product runs
    ctx.sgpr_set_panel_rows(20000)
    ec, gc = ctx.sgpr_elbo_grad(d, 0.05)
    ctx.sgpr_set_panel_rows(0)
    assert ctx.bench_crt_info()["gemm_planes"] == 0 and ctx.sgpr_stats_precision() == "int8crt"
    np.testing.assert_allclose(gc, g0, rtol=1e-9, atol=tol)
    ctx.close()


def test_int8_adjoint_gemm_steps_aside_on_ill_conditioned_kuu():
    """G = Kfu H cancels like cond(Kuu): above the estimate 1e2 the fp64 GEMM runs (measured at estimate 5e5: 1e-5 of the largest gradient
    entry from the whitened route's with the int8 product, 5e-9 with the fp64 one)."""
    spec, X, y, Z = _ill_conditioned(65536, 768, 8, 1.5)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
    _, g = ctx.sgpr_elbo_grad(d, 0.01)
    info = ctx.bench_crt_info()
    assert ctx.sgpr_stats_precision() == "int8crt" and info["tail_dd"] == 1 and info["gemm_planes"] == 0
    ctx.sgpr_set_route("whitened")
    _, gw = ctx.sgpr_elbo_grad(d, 0.01)
    np.testing.assert_allclose(g, gw, rtol=1e-5, atol=1e-7 * np.abs(gw).max())
    ctx.sgpr_set_route("auto")
    _, gf = _with_env("OAK_CRT_GEMM", "1", lambda: ctx.sgpr_elbo_grad(d, 0.01))      # forced: runs, and shows why it is not the default here
    assert ctx.bench_crt_info()["gemm_planes"] >= 13
    assert np.abs(gf - gw).max() > 10 * np.abs(g - gw).max()
    ctx.close()


def _two_rank_job(spec, X, y, Z, split, fn, route="auto", precision="auto"):
    """Two host-exchange ranks on one GPU (two threads, the callback is a two-party sum through a barrier): fn(ctx, desc) runs on both.
    (OAK_COMM_DD=1: the exact exchange of Phi is on request only under the host-exchange communicator -- it doubles the bytes a slow control
    plane carries; under RCCL and the loopback communicator it is the default.)"""
    import threading
    if "OAK_NO_COMM_DD" not in os.environ: os.environ["OAK_COMM_DD"] = "1"
    d = _capi.KernelDesc(spec)
    ranks = []
    for lo, hi in ((0, split), (split, len(X))):
        c = _capi.HipContext(0)
        c.sgpr_set_data(X[lo:hi], y[lo:hi]); c.sgpr_set_inducing(Z); c.sgpr_set_route(route); c.sgpr_set_precision(precision)
        c.sgpr_set_global_rows(len(X))
        ranks.append(c)
    bar = threading.Barrier(2)
    slots = [None, None]

    def make_cb(r):
        def cb(a):
            slots[r] = a
            bar.wait()
            out = slots[0] + slots[1]
            bar.wait()
            return out
        return cb
    for r, c in enumerate(ranks):
        c.comm_init_host(2, r, make_cb(r))
    res = {}

    def run(r, c):
        res[r] = fn(c, d)
    th = [threading.Thread(target=run, args=(r, c)) for r, c in enumerate(ranks)]
    [t.start() for t in th]; [t.join(300) for t in th]
    os.environ.pop("OAK_COMM_DD", None)
    assert not any(t.is_alive() for t in th)
    return ranks, res


def test_two_ranks_sum_phi_exactly_and_keep_the_phi_route_on_ill_conditioned_kuu():
    """Under a communicator the shards' Phi used to be summed in fp64, so an ill-conditioned Kuu sent every rank through the N-sized
    triangular solve.  Now each rank splits its exact (double-double) Phi into two fixed-point limbs on a grid all ranks share, the two fp64
    all-reduces are exact, and the double-double tail serves the sum: the auto route stays on the phi route and meets the oracle to 1e-10 on
    every term -- bit-identical on both ranks."""
    spec, X, y, Z = _ill_conditioned(90000, 768, 8, 1.5, seed=5)
    ref, parts = c_oracle.sgpr_elbo_chunked(spec, X, y, Z, 0.01, chunk=8192, return_parts=True)

    def job(c, d):
        e, g = c.sgpr_elbo_grad(d, 0.01)
        return e, g, c.sgpr_last_terms(), c.sgpr_stats_whitened(), c.sgpr_stats_precision(), c.bench_crt_info()
    ranks, res = _two_rank_job(spec, X, y, Z, 41000, job)
    for r in (0, 1):
        e, g, t, whitened, prec, info = res[r]
        assert not whitened and prec == "int8crt" and info["tail_dd"] == 1 and t["cond_estimate"] > 1e2, (whitened, prec, info)
        cases.assert_terms_match(t, parts["terms"], rtol=1e-10, what=f"rank {r}, two shards, exact exchange of Phi:")
        scale = max(abs(ref), 0.5 * abs(parts["terms"]["cTc"]), 0.5 * abs(parts["terms"]["tr_AAT"]))
        assert abs(e - ref) <= 1e-10 * scale, (e, ref)
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    for c in ranks: c.close()
    # the same job with the exchange switched off: the old rule whitens (rank 0's estimate), and agrees
    try:
        os.environ["OAK_NO_COMM_DD"] = "1"
        ranks, res2 = _two_rank_job(spec, X, y, Z, 41000, job)
    finally:
        os.environ.pop("OAK_NO_COMM_DD", None)
    assert res2[0][3] and res2[1][3]
    assert abs(res2[0][0] - res[0][0]) <= 2e-10 * scale           # (the total is a difference of terms a few hundred times its size)
    np.testing.assert_allclose(res[0][1], res2[0][1], rtol=1e-5, atol=1e-6 * np.abs(res2[0][1]).max())
    for c in ranks: c.close()


def test_two_ranks_exact_exchange_equals_one_rank_on_a_well_conditioned_problem():
    """Same protocol, well-conditioned Kuu, ragged shards: the two-rank ELBO / gradient equal the one-rank evaluation of the same rows to
    rounding, and one rank that cannot take the int8 route (too few rows) still follows the exchange with its fp64 Phi."""
    spec, X, y, Z = _problem(70001, 8, 768, 2, ("gaussian",), seed=41, ls=(0.4, 0.7))
    one = _capi.HipContext(0)
    d1 = _capi.KernelDesc(spec)
    one.sgpr_set_data(X, y); one.sgpr_set_inducing(Z)
    e1, g1 = one.sgpr_elbo_grad(d1, 0.05)
    assert one.sgpr_stats_precision() == "int8crt" and not one.sgpr_stats_whitened()
    one.close()

    def job(c, d):
        e, g = c.sgpr_elbo_grad(d, 0.05)
        return e, g, c.sgpr_stats_whitened(), c.sgpr_stats_precision()
    for split in (35000, 68001):                   # second: rank 1 holds 2000 rows -- fp64 kernels there
        ranks, res = _two_rank_job(spec, X, y, Z, split, job)
        assert not res[0][2] and not res[1][2] and res[0][3] == "int8crt"
        assert res[1][3] == ("int8crt" if split == 35000 else "fp64")
        assert abs(res[0][0] - e1) <= 1e-12 * abs(e1) and res[0][0] == res[1][0]
        np.testing.assert_allclose(res[0][1], g1, rtol=1e-9, atol=1e-10 * np.abs(g1).max())
        for c in ranks: c.close()


def test_loopback_eight_ranks_exact_exchange_on_ill_conditioned_kuu():
    """The loopback communicator (world 8: every collective returns 8 x the local vector, i.e. eight ranks holding the same shard) runs the
    exact exchange by default: the limbs are multiplied by 8 without rounding, the double-double tail serves the sum, and the result is the
    oracle's on the shard repeated eight times."""
    spec, X, y, Z = _ill_conditioned(36000, 768, 8, 1.5, seed=7)
    X8, y8 = np.tile(X, (8, 1)), np.tile(y, (8, 1))
    ref, parts = c_oracle.sgpr_elbo_chunked(spec, X8, y8, Z, 0.01, chunk=16384, return_parts=True)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_global_rows(len(X8))
    ctx.comm_init_loopback(8)
    e = ctx.sgpr_elbo(d, 0.01)
    t = ctx.sgpr_last_terms()
    assert not ctx.sgpr_stats_whitened() and ctx.sgpr_stats_precision() == "int8crt" and ctx.bench_crt_info()["tail_dd"] == 1
    cases.assert_terms_match(t, parts["terms"], rtol=1e-10, what="loopback x 8, exact exchange:")
    scale = max(abs(ref), 0.5 * abs(parts["terms"]["cTc"]), 0.5 * abs(parts["terms"]["tr_AAT"]))
    assert abs(e - ref) <= 1e-10 * scale, (e, ref)
    # undeclared global row count: no rank-independent bound for the limbs' grid -- the old rule (rank 0's estimate) whitens
    ctx.sgpr_set_global_rows(0)
    ew = ctx.sgpr_elbo(d, 0.01)
    assert ctx.sgpr_stats_whitened() and abs(ew - ref) <= 2e-10 * scale
    ctx.close()

"""GPU: the opt-in fp32 STATISTICS mode (oak_sgpr_set_precision(1); BASELINE.json config 5 asks for fp32, the reference itself
is fp64-only).  Only the N-sized forward statistics change type -- fp32 Kfu panel (v_exp_f32), fp32-MFMA Phi partials -- so the
mode is pinned against the library's own fp64 path and, through it, the oracle.  Tolerances (stated, not the fp64 1e-10):
every kernel-dependent term of the bound within 2e-5 relative of fp64, the total within 1e-6; Phi entries within 1e-5 of
max|Phi|.  Gradient calls and the whitened route must be bit-identical to fp64 (they ignore the mode), and so must an
evaluation whose Kuu looks ill-conditioned (an fp32 error in Phi is divided by lambda_min(Kuu) on its way into W): the mode
then falls back to fp64 and says so (oak_sgpr_stats_precision)."""
import numpy as np
import pytest

import cases
from oak import _capi
from oracle import c_oracle, oak_oracle as o

pytestmark = pytest.mark.gpu


def _both(ctx, d, s2):
    ctx.sgpr_set_precision("fp64")
    e64 = ctx.sgpr_elbo(d, s2); t64 = ctx.sgpr_last_terms()
    assert ctx.sgpr_stats_precision() == "fp64"
    ctx.sgpr_set_precision("fp32")
    e32 = ctx.sgpr_elbo(d, s2); t32 = ctx.sgpr_last_terms()
    used = ctx.sgpr_stats_precision()
    ctx.sgpr_set_precision("fp64")
    return e64, t64, e32, t32, used


@pytest.mark.parametrize("N,D,M,R,kinds", [(65536, 16, 512, 2, ("gaussian",)), (30000, 24, 384, 4, ("gaussian", "binary", "categorical", "uniform")),
                                           (20011, 14, 130, 3, ("gaussian", "mog")), (9000, 13, 200, 13, ("gaussian",))])
def test_fp32_statistics_track_the_fp64_path(N, D, M, R, kinds):
    rng = np.random.default_rng(N + D)
    spec = cases.random_spec(rng, D, R, kinds)
    for dim in spec["dims"]:                                    # moderate lengthscales in >= 13 dims: cond(Kuu) ~ 1e2..1e3
        if dim["type"] == "rbf":
            dim["lengthscale"] = float(rng.uniform(0.8, 1.3))
    X = cases.random_inputs(rng, spec, N)
    Z = X[rng.choice(N, M, replace=False)].copy()
    y = (np.sin(X[:, 0]) + 0.3 * X[:, 1 % D] + 0.1 * rng.standard_normal(N)).reshape(-1, 1)
    y = (y - y.mean()) / y.std()
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    e64, t64, e32, t32, used = _both(ctx, d, 0.05)
    assert used == "fp32" and e32 != e64                        # the mode really ran
    assert abs(e32 - e64) <= 1e-6 * abs(e64), (e32, e64)
    cases.assert_terms_match(t32, t64, rtol=2e-5, what="fp32 statistics vs fp64:")
    assert t32["kappa"] == t64["kappa"] and t32["yy"] == t64["yy"] and t32["logdet_Kuu"] == t64["logdet_Kuu"]   # fp64 pieces untouched
    # the statistics themselves (left in place by the fused evaluation)
    ctx.sgpr_set_precision("fp32"); ctx.sgpr_elbo(d, 0.05); s32 = ctx.sgpr_get_stats()
    ctx.sgpr_set_precision("fp64"); ctx.sgpr_elbo(d, 0.05); s64 = ctx.sgpr_get_stats()
    P32, P64 = s32[:M * M].reshape(M, M), s64[:M * M].reshape(M, M)
    assert np.abs(P32 - P64).max() <= 1e-5 * np.abs(P64).max()
    np.testing.assert_array_equal(P32, P32.T)
    assert np.abs(s32[M * M:M * M + M] - s64[M * M:M * M + M]).max() <= 1e-5 * np.abs(s64[M * M:M * M + M]).max()
    # gradient calls and the whitened route ignore the mode: bit-identical to fp64
    g64 = ctx.sgpr_elbo_grad(d, 0.05)
    ctx.sgpr_set_precision("fp32")
    g32 = ctx.sgpr_elbo_grad(d, 0.05)
    assert g32[0] == g64[0]
    np.testing.assert_allclose(g32[1], g64[1], rtol=1e-12, atol=1e-12 * np.abs(g64[1]).max())   # categorical-table LDS atomics: order varies
    ctx.sgpr_set_route("whitened")
    ew32 = ctx.sgpr_elbo(d, 0.05)
    ctx.sgpr_set_precision("fp64")
    assert ctx.sgpr_elbo(d, 0.05) == ew32
    ctx.close()


def test_fp32_statistics_against_the_oracle_and_chunked_panels():
    """Against the CPU oracle directly (the fp64 path is within 1e-10 of it, so the bar is the fp32 one), with the panel
    chunked so that partial accumulation across panels is exercised, and predictions from the fp32-statistics posterior."""
    X, y, Z = o.synthetic_problem(40000, 16, 256, seed=3)
    spec = o.make_spec(16, 2)
    ref, parts = c_oracle.sgpr_elbo_chunked(spec, X, y, Z, 0.01, chunk=8192, return_parts=True)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi"); ctx.sgpr_set_precision("fp32")
    for rows in (0, 7000):
        ctx.sgpr_set_panel_rows(rows)
        e = ctx.sgpr_elbo(d, 0.01)
        assert ctx.sgpr_stats_precision() == "fp32"
        assert abs(e - ref) <= 2e-6 * abs(ref), (rows, e, ref)
        cases.assert_terms_match(ctx.sgpr_last_terms(), parts["terms"], rtol=5e-5, what=f"fp32 statistics vs oracle (panel rows {rows}):")
    m, v = ctx.sgpr_predict(d, X[:500])
    mr, vr = o.sgpr_predict_f(spec, X, y, Z, 0.01, X[:500])
    assert np.abs(m - mr[:, 0]).max() <= 1e-3 * max(1.0, np.abs(mr).max()) and np.abs(v - vr[:, 0]).max() <= 1e-3
    with pytest.raises(ValueError):
        ctx.sgpr_set_precision(7)
    ctx.close()


def test_fp32_mode_falls_back_to_fp64_on_an_ill_conditioned_kuu():
    """BASELINE config 2's shape (D = 8, M = 512: cond(Kuu) ~ 9e4): the conditioning estimate refuses fp32 and the evaluation
    is the fp64 one, bit for bit."""
    X, y, Z = o.synthetic_problem(20000, 8, 512)
    spec = o.make_spec(8, 2)
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    e64, t64, e32, t32, used = _both(ctx, d, 0.01)
    est = t32.pop("cond_estimate"); t64.pop("cond_estimate")           # only the fp32-mode evaluation asks for the estimate
    assert est > 1e2 and used == "fp64" and e32 == e64 and t32 == t64
    ctx.close()


def test_fp32_mode_under_a_communicator():
    """The mode's conditioning decision is a collective (rank 0's reading, shared): under the loopback communicator (two ranks
    holding the same rows) the evaluation must run in fp32 on a well-conditioned problem, fall back on an ill-conditioned one,
    and agree with the single-rank run on the rows stacked twice to the mode's own accuracy."""
    X, y, Z = o.synthetic_problem(30000, 16, 256, seed=8)
    spec = o.make_spec(16, 2)
    d = _capi.KernelDesc(spec)
    ref = _capi.HipContext(0)
    ref.sgpr_set_data(np.tile(X, (2, 1)), np.tile(y, (2, 1))); ref.sgpr_set_inducing(Z); ref.sgpr_set_route("phi")
    e64 = ref.sgpr_elbo(d, 0.05)
    ctx = _capi.HipContext(0)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi"); ctx.sgpr_set_precision("fp32")
    ctx.comm_init_loopback(2)
    e32 = ctx.sgpr_elbo(d, 0.05)
    assert ctx.sgpr_stats_precision() == "fp32" and abs(e32 - e64) <= 1e-6 * abs(e64)
    # ill-conditioned Kuu (config 2's shape): both "ranks" fall back together
    X2, y2, Z2 = o.synthetic_problem(20000, 8, 512)
    spec2 = o.make_spec(8, 2)
    ctx.sgpr_set_data(X2, y2); ctx.sgpr_set_inducing(Z2)
    ctx.sgpr_elbo(_capi.KernelDesc(spec2), 0.01)
    assert ctx.sgpr_stats_precision() == "fp64"
    ctx.close(); ref.close()


def test_c5_full_size_in_fp32_as_baseline_config_5_states_it():
    """BASELINE.json configs[4] literally: N = 262 144, D = 32 mixed (20 RBF + 8 binary + 4 categorical), M = 2048, depth 4,
    fp32.  The fp32 statistics mode must be honoured at that size (well-conditioned Kuu) and stay within its stated
    tolerance of the fp64 path -- ELBO and every kernel-dependent term <= 1e-5 relative, Phi <= 1e-5 of max|Phi| -- while a
    16 384-row sample ties the fp64 path itself to the oracle at the full M (1e-10) and the fp32 mode to the oracle directly
    (bound and every term <= 1e-5).  Sobol indices computed from the fp32-statistics posterior still sum to one."""
    import bench
    N5, D5, M5, R5 = 262144, 32, 2048, 4
    X, y, Z = bench.synthetic(N5, D5, M5, mixed=True)
    spec = bench.make_spec(D5, R5, mixed=True)
    d = _capi.KernelDesc(spec)
    ctx = _capi.HipContext(0)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    e64, t64, e32, t32, used = _both(ctx, d, 0.01)
    assert used == "fp32" and e32 != e64
    assert abs(e32 - e64) <= 1e-5 * abs(e64), (e32, e64)
    est = t32.pop("cond_estimate"); t64.pop("cond_estimate")
    assert 0 < est <= 1e2
    cases.assert_terms_match(t32, t64, rtol=1e-5, what="C5 full size, fp32 statistics vs fp64:")
    ctx.sgpr_set_precision("fp32"); ctx.sgpr_elbo(d, 0.01); s32 = ctx.sgpr_get_stats()
    alpha32 = ctx.sgpr_alpha(M5)
    ctx.sgpr_set_precision("fp64"); ctx.sgpr_elbo(d, 0.01); s64 = ctx.sgpr_get_stats()
    P32, P64 = s32[:M5 * M5], s64[:M5 * M5]
    assert np.abs(P32 - P64).max() <= 1e-5 * np.abs(P64).max()
    # the fp64 path against the oracle on a row sample at the full M and depth
    ns = 16384
    ctx.sgpr_set_data(X[:ns], y[:ns])
    e = ctx.sgpr_elbo(d, 0.01)
    er, parts = c_oracle.sgpr_elbo_chunked(spec, X[:ns], y[:ns], Z, 0.01, 1e-6, chunk=4096, return_parts=True)
    assert abs(e - er) <= 1e-10 * abs(er)
    # ... and the fp32 statistics mode DIRECTLY against the oracle on the same sample (not only through the fp64 path): the bound
    # and every kernel-dependent term within the mode's stated 1e-5
    ctx.sgpr_set_precision("fp32")
    e32s = ctx.sgpr_elbo(d, 0.01)
    assert ctx.sgpr_stats_precision() == "fp32"
    assert abs(e32s - er) <= 1e-5 * abs(er), (e32s, er)
    t32s = ctx.sgpr_last_terms(); t32s.pop("cond_estimate", None)
    cases.assert_terms_match(t32s, parts["terms"], rtol=1e-5, what="C5 sample, fp32 statistics vs oracle:")
    ctx.sgpr_set_precision("fp64")
    # the fp32 Kuf panel itself (all three sub-kernel types, depth 4) against the oracle on sampled rows
    rows = np.random.default_rng(3).choice(N5, 512, replace=False)
    K32, Kr = ctx.gram_f32(d, X[rows], Z), c_oracle.gram(spec, X[rows], Z)
    assert K32.dtype == np.float32 and np.abs(K32 - Kr).max() <= 1e-5 * np.abs(Kr).max()
    # Sobol path from the fp32-statistics posterior: all 41 448 terms, normalised
    subsets = [list(s) for s in o.list_representation(D5, R5)[1:]]
    sob = ctx.sobol(d, Z, alpha32, subsets)
    assert len(sob) == 41448 and np.all(np.isfinite(sob)) and sob.min() >= -1e-12
    assert abs((sob / sob.sum()).sum() - 1.0) <= 1e-12
    ctx.close()

"""GPU parity: the fused HIP Gram kernel (through the C ABI) against the CPU oracle.  Tolerance: elementwise
|K_hip - K_oracle| <= 1e-12 * max|K_oracle| (fp64)."""
import numpy as np
import pytest

import cases
from conftest import GOLDEN
from oak import _capi
from oracle import oak_oracle as o

pytestmark = pytest.mark.gpu
TOL = 1e-12


def close(a, b, tol=TOL):
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(np.asarray(a) - np.asarray(b)).max() / scale
    assert err <= tol, f"max scaled error {err:.3e} > {tol:.1e}"


@pytest.mark.parametrize("R", range(0, 9))
@pytest.mark.parametrize("kinds", [("gaussian",), ("gaussian", "uniform", "mog", "none", "gauss2"),
                                   ("gaussian", "binary", "categorical")])
def test_gram_all_orders_and_kernel_types(hip, R, kinds):
    rng = np.random.default_rng(100 + R)
    D = 9
    spec = cases.random_spec(rng, D, R, kinds)
    X, X2 = cases.random_inputs(rng, spec, 203), cases.random_inputs(rng, spec, 77)
    d = _capi.KernelDesc(spec)
    close(hip.gram(d, X, X2), o.oak_K(spec, X, X2))
    close(hip.gram(d, X), o.oak_K(spec, X))
    close(hip.gram_diag(d, X), o.oak_K_diag(spec, X))


@pytest.mark.parametrize("D", [1, 16, 17, 33, 64])
def test_gram_wide_inputs(hip, D):
    """D > 16 switches the column tile (LDS budget); D = 64 is the documented maximum."""
    rng = np.random.default_rng(D)
    spec = cases.random_spec(rng, D, min(2, D), ("gaussian", "gaussian", "binary"))
    X, X2 = cases.random_inputs(rng, spec, 150), cases.random_inputs(rng, spec, 300)
    d = _capi.KernelDesc(spec)
    close(hip.gram(d, X, X2), o.oak_K(spec, X, X2))


@pytest.mark.parametrize("n1,n2", [(1, 1), (1, 513), (513, 1), (16, 255), (17, 257), (1000, 3), (3, 1000)])
def test_gram_ragged_shapes(hip, n1, n2):
    rng = np.random.default_rng(n1 * 1000 + n2)
    spec = cases.random_spec(rng, 4, 2)
    X, X2 = rng.standard_normal((n1, 4)), rng.standard_normal((n2, 4))
    d = _capi.KernelDesc(spec)
    K = hip.gram(d, X, X2)
    assert K.shape == (n1, n2)
    close(K, o.oak_K(spec, X, X2))


def test_empty_inputs(hip):
    spec = cases.random_spec(np.random.default_rng(0), 3, 2)
    d = _capi.KernelDesc(spec)
    assert hip.gram(d, np.zeros((0, 3)), np.zeros((5, 3))).shape == (0, 5)
    assert hip.gram(d, np.zeros((5, 3)), np.zeros((0, 3))).shape == (5, 0)
    assert hip.gram_diag(d, np.zeros((0, 3))).shape == (0,)


def test_not_shared_variances(hip):
    """share_var_across_orders=False: sigma2_0 e_0 + e_1 + ... + e_R with trainable base variances (oak_kernel.py:261-265)."""
    rng = np.random.default_rng(3)
    spec = cases.random_spec(rng, 5, 3, ("gaussian", "binary"), share=False)
    X, X2 = cases.random_inputs(rng, spec, 90), cases.random_inputs(rng, spec, 40)
    d = _capi.KernelDesc(spec)
    close(hip.gram(d, X, X2), o.oak_K(spec, X, X2))
    close(hip.gram_diag(d, X), o.oak_K_diag(spec, X))


def test_active_columns_and_extra_columns(hip):
    """Sub-kernels may read any column (active_dims); unused columns are ignored."""
    rng = np.random.default_rng(4)
    spec = cases.random_spec(rng, 3, 2)
    for d_, col in zip(spec["dims"], (4, 0, 2)):
        d_["active_dim"] = col
    X, X2 = rng.standard_normal((50, 6)), rng.standard_normal((20, 6))
    close(hip.gram(_capi.KernelDesc(spec), X, X2), o.oak_K(spec, X, X2))


def test_discrete_inputs_truncate_like_tf_cast(hip):
    """tf.cast(float64 -> int32) truncates (ortho_binary_kernel.py:47): 1.9 -> category 1."""
    rng = np.random.default_rng(5)
    spec = cases.random_spec(rng, 2, 2, ("binary", "categorical"))
    X = cases.random_inputs(rng, spec, 60) + 0.45
    close(hip.gram(_capi.KernelDesc(spec), X), o.oak_K(spec, X))


def test_components_sum_to_kernel(hip):
    """KernelComponenent terms through oak_gram_component (tests/test_oak_kernel.py:32-144)."""
    rng = np.random.default_rng(6)
    spec = cases.random_spec(rng, 4, 3, ("gaussian", "binary", "uniform", "categorical"))
    X, X2 = cases.random_inputs(rng, spec, 45), cases.random_inputs(rng, spec, 31)
    d = _capi.KernelDesc(spec)
    total, total_diag = np.zeros((45, 31)), np.zeros(45)
    for S in o.list_representation(4, 3):
        KS = hip.gram_component(d, S, True, X, X2)
        close(KS, o.component_K(spec, S, X, X2))
        total += KS
        total_diag += hip.gram_component_diag(d, S, True, X)
    close(total, o.oak_K(spec, X, X2), 1e-11)
    close(total_diag, o.oak_K_diag(spec, X), 1e-11)


def test_small_and_large_lengthscales_against_mpmath(hip):
    """The reference's |x|^2+|z|^2-2xz distance loses digits for tiny lengthscales; the HIP kernel uses (x-z)^2.
    Both are checked against 50-digit arithmetic: HIP within 1e-13, and never worse than the oracle."""
    import mpmath as mp
    mp.mp.dps = 50
    rng = np.random.default_rng(7)
    for l in (0.02, 0.3, 30.0):
        spec = o.make_spec(1, 1, lengthscales=[l], order_variances=[0.0, 1.0])
        X, Z = rng.standard_normal((6, 1)) * 2, rng.standard_normal((5, 1)) * 2
        K = hip.gram(_capi.KernelDesc(spec), X, Z)
        Ko = o.oak_K(spec, X, Z)
        lm = mp.mpf(l)
        c = lambda t: lm / mp.sqrt(lm ** 2 + 1) * mp.exp(-t ** 2 / (2 * (lm ** 2 + 1)))
        v = lm / mp.sqrt(lm ** 2 + 2)
        truth = np.array([[float(mp.exp(-(mp.mpf(float(x)) - mp.mpf(float(z))) ** 2 / (2 * lm ** 2)) - c(mp.mpf(float(x))) * c(mp.mpf(float(z))) / v)
                           for z in Z[:, 0]] for x in X[:, 0]])
        e_hip, e_or = np.abs(K - truth).max(), np.abs(Ko - truth).max()
        assert e_hip <= 1e-13 and e_hip <= max(e_or, 1e-15) * 4


def test_golden_vectors(hip):
    g = np.load(GOLDEN / "oracle_vectors.npz")
    for name, rows in (("A", 64), ("B", 50)):
        spec, X, y, Z, noise = getattr(cases, f"case_{name}")()
        d = _capi.KernelDesc(spec)
        close(hip.gram(d, X[:rows], Z), g[f"{name}_K"])
        close(hip.gram_diag(d, X), g[f"{name}_Kdiag"])


def test_additive_terms_and_measure_helpers(hip):
    """compute_additive_terms (oak_kernel.py:223-249) and cov_X_s / var_s (ortho_rbf_kernel.py:47-152) entry points."""
    rng = np.random.default_rng(8)
    mats = rng.standard_normal((4, 10))
    for R in (0, 2, 4):
        out = hip.additive_terms(mats, R)
        ref = o.compute_additive_terms([m for m in mats], R)
        for a, b in zip(out, ref):
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)
    spec, *_ = cases.case_B()
    X = rng.standard_normal((30, 6))
    d = _capi.KernelDesc(spec)
    for dim in (0, 1, 2, 5):
        c, v = hip.measure_cov(d, dim, X)
        np.testing.assert_allclose(c, o.cov_X_s(X[:, dim:dim + 1], spec["dims"][dim])[:, 0], rtol=1e-12)
        np.testing.assert_allclose(v, o.var_s(spec["dims"][dim]), rtol=1e-12)


def test_bad_arguments_raise(hip):
    spec = cases.random_spec(np.random.default_rng(0), 3, 2)
    d = _capi.KernelDesc(spec)
    with pytest.raises(ValueError):
        hip.gram(d, np.zeros((4, 2)))            # kernel reads column 2
    with pytest.raises(ValueError):
        hip.gram(d, np.zeros((4, 3)), np.zeros((4, 4)))
    bad = dict(spec, dims=[dict(spec["dims"][0], lengthscale=-1.0)] + spec["dims"][1:])
    with pytest.raises(ValueError):
        hip.gram(_capi.KernelDesc(bad), np.zeros((4, 3)))


def test_exp2_accuracy(hip):
    """The hand-written table+polynomial exp2 inside the Gram kernel against 50-digit arithmetic: an unconstrained 1-D RBF
    is exactly exp(-(x-z)^2 / (2 l^2)).  Forming the exponent t = -(x s)^2 in fp64 carries ~3 eps relative error, which
    any exp amplifies to ~2|t| ulp; the bound below is 4 ulp for the exp2 itself plus that conditioning term."""
    import mpmath as mp
    mp.mp.dps = 50
    rng = np.random.default_rng(9)
    spec = dict(dims=[dict(type="rbf", lengthscale=1.0, variance=1.0, measure=None)], order_variances=[0.0, 1.0],
                max_interaction_depth=1, share_var_across_orders=True)
    x = np.concatenate([rng.uniform(-6, 6, 300), 10.0 ** rng.uniform(-6, 1.5, 200), [0.0, 1e-300, 38.0]]).reshape(-1, 1)
    z = np.zeros((1, 1))
    K = hip.gram(_capi.KernelDesc(spec), x, z)[:, 0]
    truth = np.array([float(mp.exp(-mp.mpf(float(v)) ** 2 / 2)) for v in x[:, 0]])
    ok = truth > 1e-290
    ulp = np.abs(K[ok] - truth[ok]) / (np.spacing(truth[ok]))
    bound = 4.0 + 3.0 * np.abs(np.log2(truth[ok]))
    assert np.all(ulp <= bound), (ulp / bound).max()
    near = truth[ok] > 0.01                       # |t| < 7: the exp2 error dominates
    assert ulp[near].max() <= 24.0, ulp[near].max()
    assert K[-3] == 1.0 and K[-2] == 1.0
    assert np.all(K[~ok] < 1e-289)


@pytest.mark.parametrize("variances", [(0.3, 0.02, 1e-4), (1.0, 7.5, 300.0), (2.0 ** -20, 2.0 ** 20, 0.999)])
def test_far_apart_pairs_with_non_unit_base_variance(hip, variances):
    """Pairs further apart than ~38 lengthscales drive the kernels' clamped exponent form to its floor (2^-1024); with a base
    variance below 1 that floor used to fall outside the biased table's exponent range.  Gram, its diagonal and the ELBO
    gradient stay on the oracle for tiny lengthscales and base variances on both sides of 1."""
    import copy
    rng = np.random.default_rng(5)
    D = len(variances)
    spec = o.make_spec(D, 2, lengthscales=[0.01, 0.05, 0.002])
    spec["share_var_across_orders"] = False
    spec["order_variances"] = [0.8]                          # only sigma2_0 exists when the orders do not share variances
    for d_, v in zip(spec["dims"], variances):
        d_["variance"] = v
    X = rng.normal(size=(300, D)) * 2.0                       # |x - z| / l up to ~ 4000
    Z = X[:37] + 1e-3 * rng.normal(size=(37, D))              # a few near pairs as well
    dsc = _capi.KernelDesc(spec)
    K, Kr = hip.gram(dsc, X, Z), o.oak_K(spec, X, Z)
    assert np.isfinite(K).all()
    # lengthscales of 0.002: x / l ~ 1e3, so the scaled difference of near pairs loses ~3 digits in ANY fp64 evaluation
    # (the oracle's GPflow-form square distance more than the kernel's direct form): 1e-9, not the usual 1e-12
    close(K, Kr, tol=1e-9)
    # K_diag: the oracle follows the reference's Newton-Girard power sums, e_2 = (s_1^2 - s_2) / 2, which cancel when one
    # k_d (here up to 2^20) dwarfs the others -- 9e-11 off a 40-digit evaluation, the HIP recurrence 2e-16 (tests/dev/dev_diag_check.py)
    close(hip.gram_diag(dsc, X), o.oak_K_diag(spec, X), tol=1e-9)
    y = rng.normal(size=(300, 1))
    hip.sgpr_set_data(X, y); hip.sgpr_set_inducing(Z); hip.sgpr_set_route("whitened")
    e, g = hip.sgpr_elbo_grad(dsc, 0.3)
    er = o.sgpr_elbo(spec, X, y, Z, 0.3)
    assert abs(e - er) <= 1e-8 * abs(er) and np.isfinite(g).all()
    h = 1e-6 * variances[1]
    sp, sm = copy.deepcopy(spec), copy.deepcopy(spec)
    sp["dims"][1]["variance"] += h; sm["dims"][1]["variance"] -= h
    fd = (o.sgpr_elbo(sp, X, y, Z, 0.3) - o.sgpr_elbo(sm, X, y, Z, 0.3)) / (2 * h)
    assert abs(g[D + 1] - fd) <= 1e-4 * max(abs(fd), 1e-8)


@pytest.mark.parametrize("ls", [1e-3, 3e-2, 50.0, 1e3])
def test_lengthscales_at_the_bounds_of_the_reference(hip, ls, monkeypatch):
    """oak_model bounds the lengthscales to [1e-3, 1e3] (model_utils.py:199).  At the upper bound exp(.) and cn cn cancel to
    ~1e-6; at the lower bound x / l reaches the thousands and the reference's squared distance in GPflow's
    |x|^2 + |z|^2 - 2 x z form loses |x / l|^2 * eps ~ 1e-9 to cancellation on near-coincident pairs, where the HIP kernel
    subtracts first.  So: HIP vs the oracle with the distance formed as (x - z)^2 <= 1e-12 of the largest entry at every
    lengthscale; HIP vs the reference-form oracle within that form's own cancellation bound."""
    rng = np.random.default_rng(int(ls * 1000) % 97)
    D, R = 4, 2
    spec = o.make_spec(D, R, lengthscales=[ls, ls * 1.5, 1.0, ls], order_variances=[0.8, 1.1, 0.6])
    X, X2 = rng.standard_normal((150, D)), rng.standard_normal((70, D))
    X2[:5] = X[:5] + 4e-4 * rng.standard_normal((5, D))      # near-coincident pairs: the worst case of the expanded form
    d = _capi.KernelDesc(spec)
    got, got_diag = hip.gram(d, X, X2), hip.gram_diag(d, X)
    ref_form = o.oak_K(spec, X, X2)
    bound = 16 * np.finfo(float).eps * max(1.0, float(np.abs(X / min(ls, 1.0)).max()) ** 2) * np.abs(ref_form).max()
    assert np.abs(got - ref_form).max() <= max(bound, 1e-12 * np.abs(ref_form).max())

    def rbf_direct(A, B, lengthscale, variance):
        A = np.asarray(A, dtype=np.float64); B = A if B is None else np.asarray(B, dtype=np.float64)
        r2 = (((A[:, None, :] - B[None, :, :]) / lengthscale) ** 2).sum(-1)
        return variance * np.exp(-0.5 * r2)

    monkeypatch.setattr(o, "rbf_K", rbf_direct)
    ref = o.oak_K(spec, X, X2)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    ref_diag = o.oak_K_diag(spec, X)
    assert np.abs(got_diag - ref_diag).max() <= 1e-12 * np.abs(ref_diag).max()


@pytest.mark.parametrize("D,R", [(9, 9), (13, 13), (12, 10), (20, 16), (60, 11), (7, 12)])
def test_gram_beyond_depth_eight(hip, D, R):
    """The reference's Newton-Girard loop takes any max_interaction_depth (oak_kernel.py:236-249) and its regression example
    runs depth = D, 13 on UCI housing (examples/uci/uci_regression_train.py:86).  Depths 9..16 run the next larger
    instantiation of the fused kernels with zero weights above R: K, K_diag and a component against the oracle; where the
    subset count allows, also against the exact sum over subsets (Newton-Girard itself loses digits at high order)."""
    import test_gpu_fuzz as fz
    rng = np.random.default_rng(D * 31 + R)
    spec = cases.random_spec(rng, D, R, ("gaussian", "uniform", "binary", "gaussian", "categorical"))
    X, X2 = cases.random_inputs(rng, spec, 83), cases.random_inputs(rng, spec, 140)
    d = _capi.KernelDesc(spec)
    K, Kd = hip.gram(d, X, X2), hip.gram_diag(d, X)
    if D <= 13:                                    # <= 8192 subsets: exact reference
        Kb = fz.brute_force_K(spec, X, X2)
        assert np.abs(K - Kb).max() <= 1e-12 * np.abs(Kb).max()
        Kdb = fz.brute_force_K(spec, X, None, diag=True)
        assert np.abs(Kd - Kdb).max() <= 1e-12 * np.abs(Kdb).max()
    Kr = o.oak_K(spec, X, X2)
    assert np.abs(K - Kr).max() <= 1e-9 * np.abs(Kr).max()          # the oracle's Newton-Girard carries the cancellation
    assert np.abs(Kd - o.oak_K_diag(spec, X)).max() <= 1e-9 * np.abs(Kd).max()
    sub = list(range(min(D, R)))[:9]
    close(hip.gram_component(d, sub, True, X, X2), o.component_K(spec, sub, X, X2))
    with pytest.raises(ValueError):
        _capi.KernelDesc(dict(spec, max_interaction_depth=65, order_variances=[1.0] * 66))


# ---- A/B: the reference's arithmetic reproduced on the device (oak_set_gram_form) ------------------------------------------
def _ab_cases():
    rng = np.random.default_rng(77)
    out = []
    # (a) lengthscales at the lower bound with near-coincident pairs: GPflow's expanded distance loses |x/l|^2 eps there
    spec = o.make_spec(4, 2, lengthscales=[1e-3, 1.5e-3, 1.0, 1e-3], order_variances=[0.8, 1.1, 0.6])
    X, X2 = rng.standard_normal((150, 4)), rng.standard_normal((70, 4))
    X2[:5] = X[:5] + 4e-4 * rng.standard_normal((5, 4))
    out.append(("lengthscale 1e-3, near-coincident pairs", spec, X, X2))
    # (b) base variances 2^20 apart at depth 5: the Newton-Girard alternating sum cancels (6e-10 here), the recurrence does not
    spec = o.make_spec(8, 5, lengthscales=list(np.linspace(0.7, 1.6, 8)), order_variances=list(np.linspace(0.5, 1.5, 6)))
    for d, dim in enumerate(spec["dims"]):
        dim["variance"] = float(2.0 ** (20 * (d % 2)))
    out.append(("base variances 1 and 2^20, depth 5", spec, rng.standard_normal((90, 8)), rng.standard_normal((60, 8))))
    # (c) an ordinary mixed kernel: the two forms agree to rounding
    spec = cases.random_spec(rng, 9, 3, ("gaussian", "uniform", "binary", "categorical", "mog"))
    out.append(("ordinary mixed kernel, depth 3", spec, cases.random_inputs(rng, spec, 80), cases.random_inputs(rng, spec, 50)))
    return out


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_reference_arithmetic_form_matches_the_oracle_entry_by_entry(hip, idx):
    """oak_set_gram_form("reference"): GPflow's expanded squared distance, power sums and the Newton-Girard alternating sum
    (oak/oak_kernel.py:236-249, ortho_rbf_kernel.py:157-172) evaluated on the device must equal oracle.oak_K -- which IS that
    form, operation for operation -- to 1e-13 of the largest entry EVEN where the form itself is ill-behaved, while the native
    arithmetic (direct distance, exact-sum recurrence) differs from it there by the form's own cancellation error.  This is the
    entry-by-entry demonstration that the two deliberate deviations are deviations towards accuracy."""
    name, spec, X, X2 = _ab_cases()[idx]
    d = _capi.KernelDesc(spec)
    ref = o.oak_K(spec, X, X2)
    ref_diag = o.oak_K_diag(spec, X)
    scale = np.abs(ref).max()
    try:
        hip.set_gram_form("reference")
        ab, ab_diag = hip.gram(d, X, X2), hip.gram_diag(d, X)
        ab_sym = hip.gram(d, X)
    finally:
        hip.set_gram_form("native")
    native = hip.gram(d, X, X2)
    if idx == 1:
        # here the cancellation sits AFTER the exponentials: a last-place difference between the device's exp and NumPy's is
        # amplified exactly like the form's own rounding, so two executions of the reference form agree with each other only
        # as well as either agrees with the exact value -- which is the point: both carry the form's error, the native
        # arithmetic does not
        import test_gpu_fuzz as fz
        exact0 = fz.brute_force_K(spec, X, X2)
        err_ref, err_ab = np.abs(ref - exact0).max() / scale, np.abs(ab - exact0).max() / scale
        assert err_ref > 1e-13 and err_ab <= 10 * err_ref and err_ab >= err_ref / 10, (err_ref, err_ab)
        assert np.abs(ab - ref).max() <= 10 * err_ref * scale
    else:
        assert np.abs(ab - ref).max() <= 1e-13 * scale, (name, np.abs(ab - ref).max() / scale)
    dev = np.abs(native - ref).max() / scale
    if idx == 2:
        assert dev <= 1e-12                                   # nothing to deviate about
    else:
        # the native form is NOT the reference's here: it differs by more than the A/B form does, by the reference form's own
        # error (bounded by its cancellation estimate), and it is the one that agrees with an exact evaluation
        assert dev > 1e-13
        if idx == 0:      # against the same kernel with the distance formed directly, in extended precision
            def rbf_direct(A, B, lengthscale, variance):
                A = np.asarray(A, dtype=np.longdouble); B = A if B is None else np.asarray(B, dtype=np.longdouble)
                r2 = (((A[:, None, :] - B[None, :, :]) / np.longdouble(lengthscale)) ** 2).sum(-1)
                return (variance * np.exp(-0.5 * r2)).astype(np.float64)
            saved, o.rbf_K = o.rbf_K, rbf_direct
            try:
                exact = o.oak_K(spec, X, X2)
            finally:
                o.rbf_K = saved
        else:             # against the exact sum over subsets
            import test_gpu_fuzz as fz
            exact = fz.brute_force_K(spec, X, X2)
        assert np.abs(native - exact).max() <= 1e-12 * scale
        assert np.abs(ref - exact).max() > np.abs(native - exact).max()        # the reference form is the less accurate one
    print(f"[A/B] {name}: reference-form device vs oracle {np.abs(ab - ref).max() / scale:.1e}, native vs oracle {dev:.1e} (of max|K|)")


@pytest.mark.parametrize("D,R", [(20, 20), (24, 17), (32, 32), (40, 33)])
def test_gram_beyond_an_effective_depth_of_sixteen(hip, D, R):
    """The reference's loop takes any depth (oak_kernel.py:236-249; its regression example runs depth = D, 32 on pumadyn32nm).
    Effective depths min(R, D) of 17..32 run the R = 24 / 32 instantiations of the fused kernels -- K, K_diag, the SGPR objective and
    its gradient; beyond 32 the explicit Gram entry points run a generic one-thread-per-entry kernel and the model paths refuse.
    K and K_diag against the exact-sum recurrence in extended precision."""
    rng = np.random.default_rng(D * 7 + R)
    spec = cases.random_spec(rng, D, R, ("gaussian", "binary", "gaussian", "categorical"))
    X, X2 = cases.random_inputs(rng, spec, 23), cases.random_inputs(rng, spec, 31)
    d = _capi.KernelDesc(spec)
    K, Kd = hip.gram(d, X, X2), hip.gram_diag(d, X)
    # extended-precision recurrence over the oracle's per-dimension matrices
    mats = [o.base_K(X[:, [o.active_col(spec, i)]], X2[:, [o.active_col(spec, i)]], dim).astype(np.longdouble) for i, dim in enumerate(spec["dims"])]
    e = [np.ones_like(mats[0])] + [np.zeros_like(mats[0]) for _ in range(R)]
    for k in mats:
        for r in range(R, 0, -1):
            e[r] = e[r] + k * e[r - 1]
    ref = sum(np.longdouble(w) * er for w, er in zip(spec["order_variances"], e))
    assert np.abs(K - ref).max() <= 1e-12 * np.abs(ref).max()
    diags = [o.base_K_diag(X[:, [o.active_col(spec, i)]], dim).astype(np.longdouble) for i, dim in enumerate(spec["dims"])]
    ed = [np.ones_like(diags[0])] + [np.zeros_like(diags[0]) for _ in range(R)]
    for k in diags:
        for r in range(R, 0, -1):
            ed[r] = ed[r] + k * ed[r - 1]
    refd = sum(np.longdouble(w) * er for w, er in zip(spec["order_variances"], ed))
    assert np.abs(Kd - refd).max() <= 1e-12 * np.abs(refd).max()
    ctx = _capi.HipContext(0)
    y = rng.standard_normal((len(X), 1))
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(X2[:9])
    if min(R, D) > 32:
        # the fused model paths say so loudly instead of running at a wrong depth
        with pytest.raises(ValueError, match="effective interaction depth"):
            ctx.sgpr_elbo(d, 0.1)
    else:
        # the bound from the extended-precision Gram matrices (the oracle's Newton-Girard form is no reference at this depth), and the
        # gradient against central differences of the device objective
        def kmat(A, B_):
            ms = [o.base_K(A[:, [o.active_col(spec, i)]], B_[:, [o.active_col(spec, i)]], dim).astype(np.longdouble) for i, dim in enumerate(spec["dims"])]
            ee = [np.ones_like(ms[0])] + [np.zeros_like(ms[0]) for _ in range(R)]
            for k in ms:
                for r in range(R, 0, -1):
                    ee[r] = ee[r] + k * ee[r - 1]
            return np.asarray(sum(np.longdouble(w) * er for w, er in zip(spec["order_variances"], ee)), dtype=np.float64)
        Z = X2[:9]
        s2, N, M = 0.1, len(X), 9
        Kuu, Kuf = kmat(Z, Z) + 1e-6 * np.eye(M), kmat(Z, X)
        L = np.linalg.cholesky(Kuu)
        A = np.linalg.solve(L, Kuf) / np.sqrt(s2)
        LB = np.linalg.cholesky(np.eye(M) + A @ A.T)
        c = np.linalg.solve(LB, A @ y) / np.sqrt(s2)
        bound = (-0.5 * N * np.log(2 * np.pi * s2) - np.log(np.diag(LB)).sum() - 0.5 * float((y ** 2).sum()) / s2 + 0.5 * float((c ** 2).sum())
                 - 0.5 * float(np.asarray(refd, dtype=np.float64).sum()) / s2 + 0.5 * np.trace(A @ A.T))
        for route in ("phi", "whitened"):
            ctx.sgpr_set_route(route)
            assert abs(ctx.sgpr_elbo(d, s2) - bound) <= 1e-9 * abs(bound), route
        e0, g = ctx.sgpr_elbo_grad(d, s2)
        assert np.isfinite(g).all()
        import copy
        for which, idx in (("lengthscale", 0), ("order", R)):
            gi = idx if which == "lengthscale" else 2 * D + idx
            if which == "lengthscale" and spec["dims"][0]["type"] != "rbf":
                continue
            def at(v):
                sp = copy.deepcopy(spec)
                if which == "lengthscale":
                    sp["dims"][0]["lengthscale"] = v
                else:
                    sp["order_variances"][idx] = v
                return ctx.sgpr_elbo(_capi.KernelDesc(sp), s2)
            x0 = float(spec["dims"][0]["lengthscale"]) if which == "lengthscale" else float(spec["order_variances"][idx])
            h = 1e-5 * max(1.0, abs(x0))
            fd = (at(x0 + h) - at(x0 - h)) / (2 * h)
            assert abs(g[gi] - fd) <= 1e-5 * max(1.0, abs(fd)), (which, g[gi], fd)
    ctx.close()


def test_depth_beyond_the_number_of_dimensions_runs_at_that_number(hip):
    """max_interaction_depth = 20 over 6 sub-kernels: e_r vanishes for r > 6, so every path (fused Gram, SGPR objective and its
    gradient) runs at depth 6 and equals the depth-6 model with the same first seven order variances; the gradient w.r.t.
    the order variances beyond 6 is exactly zero."""
    rng = np.random.default_rng(5)
    D, R = 6, 20
    ov = list(rng.uniform(0.5, 1.5, R + 1))
    spec = o.make_spec(D, R, lengthscales=list(rng.uniform(0.7, 1.5, D)), order_variances=ov)
    spec6 = o.make_spec(D, D, lengthscales=[dm["lengthscale"] for dm in spec["dims"]], order_variances=ov[:D + 1])
    X, y, Z = o.synthetic_problem(3000, D, 40, seed=9)
    d, d6 = _capi.KernelDesc(spec), _capi.KernelDesc(spec6)
    np.testing.assert_array_equal(hip.gram(d, X[:50], Z), hip.gram(d6, X[:50], Z))
    ctx = _capi.HipContext(0)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
    e, g = ctx.sgpr_elbo_grad(d, 0.05)
    e6, g6 = ctx.sgpr_elbo_grad(d6, 0.05)
    assert e == e6
    np.testing.assert_array_equal(g[:2 * D + D + 1], g6[:2 * D + D + 1])
    assert np.all(g[2 * D + D + 1:2 * D + R + 1] == 0.0) and g[2 * D + R + 1] == g6[2 * D + D + 1]     # [.., order vars 7..20 = 0, noise]
    ctx.close()


def test_grouped_active_dims_are_one_rbf_over_the_group(hip):
    """OAKKernel(active_dims=[[0, 1], [2], [4, 3, 5]], constrain_orthogonal=False): the reference evaluates each group as ONE
    base kernel over the group's columns (oak/oak_kernel.py:74-82,199-210) -- for the RBF a product of one-column RBFs with a
    shared lengthscale.  K and K_diag through the host mirror against the oracle (rbf_K on the group's columns), in the
    native and in the reference arithmetic."""
    from oak import gpflow_lite as gpflow
    from oak.oak_kernel import OAKKernel, kernel_to_spec
    groups = [[0, 1], [2], [4, 3, 5]]
    k = OAKKernel([gpflow.RBF] * 3, num_dims=6, max_interaction_depth=3, active_dims=groups, constrain_orthogonal=False)
    for sub, ls in zip(k.kernels, (0.8, 1.7, 1.2)):
        sub.lengthscales.assign(ls)
    for v, val in zip(k.variances, (0.7, 1.3, 0.9, 0.4)):
        v.assign(val)
    rng = np.random.default_rng(11)
    X, X2 = rng.standard_normal((70, 6)), rng.standard_normal((45, 6))
    spec = kernel_to_spec(k)
    assert [d.get("active_dims") for d in spec["dims"]] == [[0, 1], None, [4, 3, 5]]
    # the oracle on the same description, and by hand: e_r of the three group matrices
    mats = [o.rbf_K(X[:, g], X2[:, g], ls, 1.0) for g, ls in zip(groups, (0.8, 1.7, 1.2))]
    e1 = mats[0] + mats[1] + mats[2]
    e2 = mats[0] * mats[1] + mats[0] * mats[2] + mats[1] * mats[2]
    e3 = mats[0] * mats[1] * mats[2]
    by_hand = 0.7 + 1.3 * e1 + 0.9 * e2 + 0.4 * e3
    ref = o.oak_K(spec, X, X2)
    np.testing.assert_allclose(ref, by_hand, rtol=1e-12)
    got = k.K(X, X2).numpy()
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    np.testing.assert_allclose(k.K(X).numpy(), o.oak_K(spec, X), rtol=0, atol=1e-12 * np.abs(ref).max())
    np.testing.assert_allclose(k.K_diag(X).numpy(), np.full(70, 0.7 + 1.3 * 3 + 0.9 * 3 + 0.4), rtol=1e-13)
    try:
        hip.set_gram_form("reference")
        assert np.abs(hip.gram(_capi.KernelDesc(spec), X, X2) - ref).max() <= 1e-13 * np.abs(ref).max()
    finally:
        hip.set_gram_form("native")
    ctx = _capi.HipContext(0)     # the model paths take the grouped description too (tests/test_gpu_grouped.py)
    ctx.sgpr_set_data(X, X[:, :1]); ctx.sgpr_set_inducing(X2)
    np.testing.assert_allclose(ctx.sgpr_elbo(_capi.KernelDesc(spec), 0.1), o.sgpr_elbo(spec, X, X[:, :1], X2, 0.1), rtol=1e-9)
    ctx.close()
    # a constrained kernel is one-dimensional in the reference too (ortho_rbf_kernel.py:50,83)
    with pytest.raises(NotImplementedError):
        _capi.KernelDesc(dict(dims=[dict(type="rbf", lengthscale=1.0, variance=1.0, measure=("gaussian", 0.0, 1.0), active_dims=[0, 1])],
                              order_variances=[1.0, 1.0], max_interaction_depth=1, share_var_across_orders=True))


def test_binary_sub_kernel_with_a_non_unit_base_variance_in_every_gram_form(hip):
    """The forward Gram kernel evaluates a binary sub-kernel as the rank-one product cn_a * cn_b with cn = a(x) sqrt(bv) riding in
    the feature slot the continuous dims use for their constraint term (csrc/oak_internal.h, Feat); the generic kernel
    (oak_set_gram_form, one thread per entry) and the diagonal kernel read the 2 x 2 table instead.  With bv != 1 the two only
    agree if the sqrt(bv) factor and the table carry the same variance: fused == generic == oracle."""
    rng = np.random.default_rng(5)
    X = np.column_stack([rng.normal(size=300), (rng.random(300) < 0.35).astype(float), rng.normal(size=300), (rng.random(300) < 0.6).astype(float)])
    X2 = X[:70].copy()
    spec = o.make_spec(4, 3, p0=[None, 0.65, None, 0.4], lengthscales=[0.8, 1.0, 1.3, 1.0], base_variances=[1.0, 2.75, 0.6, 0.3],
                       order_variances=[0.9, 1.1, 0.7, 1.4])
    d = _capi.KernelDesc(spec)
    ref = o.oak_K(spec, X, X2)
    fused = hip.gram(d, X, X2)
    close(fused, ref)
    close(hip.gram_diag(d, X), o.oak_K_diag(spec, X))
    try:
        hip.set_gram_form("reference")
        generic = hip.gram(d, X, X2)
    finally:
        hip.set_gram_form("native")
    close(generic, ref, 1e-11)
    close(generic, fused, 1e-11)

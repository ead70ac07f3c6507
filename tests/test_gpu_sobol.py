"""GPU parity of the Sobol / per-component prediction path (oak_sobol, oak_sobol_L, oak_component_predict)."""
import numpy as np
import pytest

import cases
from oak import _capi
from oracle import oak_oracle as o

pytestmark = pytest.mark.gpu


def test_L_matrices_match_oracle(hip):
    rng = np.random.default_rng(0)
    X = rng.standard_normal((40, 3))
    d = _capi.KernelDesc(dict(dims=[dict(type="rbf", lengthscale=1.7, variance=1.0, measure=("gaussian", 0.0, 1.0), active_dim=1)],
                              order_variances=[0.0, 1.0], max_interaction_depth=1, share_var_across_orders=True))
    np.testing.assert_allclose(hip.sobol_L(d, 0, 2.3, 1.0, 0.0, X), o.compute_L(X, 1.7, 2.3, 1, 1.0, 0.0), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(hip.sobol_L(d, 0, 1.0, 1.4, 0.3, X), o.compute_L(X, 1.7, 1.0, 1, 1.4, 0.3), rtol=1e-11, atol=1e-13)
    Xb = rng.integers(0, 2, (40, 2)).astype(float)
    db = _capi.KernelDesc(dict(dims=[dict(type="binary", p0=0.77, variance=1.0, active_dim=1)], order_variances=[0.0, 1.0],
                               max_interaction_depth=1, share_var_across_orders=True))
    assert np.abs(hip.sobol_L(db, 0, 1.0, 1.0, 0.0, Xb) - o.compute_L_binary_kernel(Xb, 0.77, 1.0, 1)).max() < 1e-15
    p = np.array([0.2, 0.5, 0.3]).reshape(-1, 1)
    W, kappa = rng.uniform(size=(3, 2)), np.array([1.0, 0.5, 2.0])
    Xc = rng.integers(0, 3, (40, 1)).astype(float)
    dc = _capi.KernelDesc(dict(dims=[dict(type="categorical", p=p, W=W, kappa=kappa, variance=1.0)], order_variances=[0.0, 1.0],
                               max_interaction_depth=1, share_var_across_orders=True))
    np.testing.assert_allclose(hip.sobol_L(dc, 0, 1.3, 1.0, 0.0, Xc), o.compute_L_categorical_kernel(Xc, W, kappa, p, 1.3, 0), rtol=1e-12)


@pytest.mark.parametrize("name", ["A", "B"])
def test_sobol_matches_oracle_on_golden_cases(hip, name):
    spec, X, y, Z, noise = getattr(cases, f"case_{name}")()
    if name == "B":
        spec["dims"][2]["measure"] = ("gaussian", 0.0, 1.0)   # MOG has no Sobol closed form (utils.py:413-414)
    alpha = o.sgpr_alpha(spec, X, y, Z, noise)
    subsets, ref = o.compute_sobol_oak(spec, Z, alpha)
    got = hip.sobol(_capi.KernelDesc(spec), Z, alpha[:, 0], subsets)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
    np.testing.assert_allclose(got / got.sum(), np.array(ref) / np.sum(ref), atol=1e-9)


def test_sobol_mog_not_supported(hip):
    spec, X, y, Z, noise = cases.case_B()
    with pytest.raises(ValueError):
        hip.sobol(_capi.KernelDesc(spec), Z, np.ones(len(Z)), [[2]])


def test_component_predictions(hip):
    spec, X, y, Z, noise = cases.case_B()
    alpha = o.sgpr_alpha(spec, X, y, Z, noise)
    subsets = o.list_representation(6, 3)[1:]
    got = hip.component_predict(_capi.KernelDesc(spec), X[:70], Z, alpha[:, 0], subsets)
    ref = o.prediction_components(spec, Z, alpha, X[:70])
    np.testing.assert_allclose(got, np.array(ref), rtol=1e-9, atol=1e-11)


def test_categorical_L_clamps_codes_outside_the_table(hip):
    """A categorical code outside [0, C-1] (an unseen code at an inducing point) is clamped exactly as the Gram path clamps
    it, never read past the C x C table."""
    rng = np.random.default_rng(3)
    p = np.array([0.2, 0.5, 0.3]).reshape(-1, 1)
    W, kappa = rng.uniform(size=(3, 2)), np.array([1.0, 0.5, 2.0])
    dc = _capi.KernelDesc(dict(dims=[dict(type="categorical", p=p, W=W, kappa=kappa, variance=1.0)], order_variances=[0.0, 1.0],
                               max_interaction_depth=1, share_var_across_orders=True))
    Xc = rng.integers(0, 3, (30, 1)).astype(float)
    Xbad = Xc.copy()
    Xbad[Xc[:, 0] == 2] = 7.0          # beyond the table -> C-1
    Xbad[Xc[:, 0] == 0] = -4.0         # below it -> 0
    np.testing.assert_array_equal(hip.sobol_L(dc, 0, 1.3, 1.0, 0.0, Xbad), hip.sobol_L(dc, 0, 1.3, 1.0, 0.0, Xc))
    np.testing.assert_array_equal(hip.gram(dc, Xbad), hip.gram(dc, Xc))


# ---- Gram of products (every term one entry of a weighted fp64-MFMA Gram matrix over the index pairs) ---------------------
def _both_paths(hip, desc, Z, alpha, subsets, **kw):
    try:
        hip.sobol_set_path("terms")
        terms = hip.sobol(desc, Z, alpha, subsets, **kw)
        assert hip.sobol_last_info()["path"] == "terms"
        hip.sobol_set_path("gram")
        gram = hip.sobol(desc, Z, alpha, subsets, **kw)
        info = hip.sobol_last_info()
        assert info["path"] == "gram"
    finally:
        hip.sobol_set_path("auto")
    return terms, gram, info


@pytest.mark.parametrize("name", ["A", "B"])
def test_gram_of_products_matches_oracle_on_golden_cases(hip, name):
    spec, X, y, Z, noise = getattr(cases, f"case_{name}")()
    if name == "B":
        spec["dims"][2]["measure"] = ("gaussian", 0.0, 1.0)
    alpha = o.sgpr_alpha(spec, X, y, Z, noise)
    subsets, ref = o.compute_sobol_oak(spec, Z, alpha)
    terms, gram, info = _both_paths(hip, _capi.KernelDesc(spec), Z, alpha[:, 0], subsets)
    np.testing.assert_allclose(gram, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
    np.testing.assert_allclose(gram / gram.sum(), np.array(ref) / np.sum(ref), atol=1e-9)
    np.testing.assert_allclose(gram, terms, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
    assert info["pair_rows"] == len(Z) * (len(Z) + 1) // 2


@pytest.mark.parametrize("depth,share,n", [(1, True, 33), (2, False, 64), (3, True, 70), (4, True, 129), (4, False, 40), (5, True, 50),
                                           (6, True, 37)])
def test_gram_of_products_every_depth_and_variance_mode(hip, depth, share, n):
    """Depths 1..6 (halves of up to three dims), shared and per-dimension variances, every sub-kernel type as first factor,
    sizes that are not multiples of the builder's 32-row blocks; the three pairings of each order-4 term agree."""
    rng = np.random.default_rng(100 * depth + n)
    D = 7
    spec = cases.random_spec(rng, D, depth, kinds=("binary", "gaussian", "categorical", "gauss2", "uniform"), share=share)
    Z = cases.random_inputs(rng, spec, n)
    alpha = rng.standard_normal((n, 1)) * rng.uniform(0.1, 3.0, (n, 1))
    subsets, ref = o.compute_sobol_oak(spec, Z, alpha, share_var_across_orders=share)
    terms, gram, info = _both_paths(hip, _capi.KernelDesc(spec), Z, alpha[:, 0], subsets, use_order_var=share)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(gram, ref, rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(terms, ref, rtol=1e-9, atol=1e-11 * scale)
    assert info["pairing_disagreement"] < 1e-12
    if depth >= 4:
        assert info["pairing_disagreement"] > 0.0        # the redundant entries were really read


def test_gram_of_products_sign_patterns_and_odd_subset_lists(hip):
    """All-positive, all-negative, zero and single-point alphas; an arbitrary (unsorted, partial) subset list; the automatic
    choice falls back to the per-term kernel for a subset with a repeated dim and the forced Gram path refuses it."""
    rng = np.random.default_rng(5)
    spec = cases.random_spec(rng, 6, 4, kinds=("gaussian", "binary", "categorical"))
    d = _capi.KernelDesc(spec)
    n = 45
    Z = cases.random_inputs(rng, spec, n)
    subsets = [[4], [2, 0], [5, 1, 3], [3, 0, 4, 1], [1], [0, 1, 2, 3]]
    for alpha in (np.abs(rng.standard_normal(n)), -np.abs(rng.standard_normal(n)), np.zeros(n),
                  np.where(np.arange(n) == 7, 1.5, 0.0), rng.standard_normal(n)):
        ref = np.array(o.compute_sobol_oak(spec, Z, alpha.reshape(-1, 1), subsets=subsets)[1])
        terms, gram, _ = _both_paths(hip, d, Z, alpha, subsets)
        np.testing.assert_allclose(gram, ref, rtol=1e-9, atol=1e-12 * max(np.abs(ref).max(), 1e-300))
        np.testing.assert_allclose(terms, ref, rtol=1e-9, atol=1e-12 * max(np.abs(ref).max(), 1e-300))
    one = hip.sobol(d, Z[:1], np.array([0.7]), subsets)
    ref1 = np.array(o.compute_sobol_oak(spec, Z[:1], np.array([[0.7]]), subsets=subsets)[1])
    np.testing.assert_allclose(one, ref1, rtol=1e-10)
    rep = [[0, 0], [1, 2]]
    auto = hip.sobol(d, Z, rng.standard_normal(n), rep)
    assert hip.sobol_last_info()["path"] == "terms" and np.isfinite(auto).all()
    try:
        hip.sobol_set_path("gram")
        with pytest.raises(ValueError):
            hip.sobol(d, Z, np.ones(n), rep)
        with pytest.raises(ValueError):
            hip.sobol(d, Z, np.ones(n), [[0, 1, 2, 3, 4, 5, 0]])
    finally:
        hip.sobol_set_path("auto")


def test_gram_of_products_row_chunks(hip, monkeypatch):
    """Panels that do not fit one chunk accumulate into the same split partials: many small chunks == one."""
    rng = np.random.default_rng(8)
    spec = cases.random_spec(rng, 9, 4, kinds=("gaussian", "binary", "gauss2", "categorical"))
    d = _capi.KernelDesc(spec)
    n = 150
    Z = cases.random_inputs(rng, spec, n)
    alpha = rng.standard_normal(n)
    subsets = o.list_representation(9, 4)[1:]
    try:
        hip.sobol_set_path("gram")
        whole = hip.sobol(d, Z, alpha, subsets)
        monkeypatch.setenv("OAK_SOBOL_CHUNK_ROWS", "992")
        parts = hip.sobol(d, Z, alpha, subsets)
    finally:
        hip.sobol_set_path("auto")
    np.testing.assert_allclose(parts, whole, rtol=1e-12, atol=1e-14 * np.abs(whole).max())
    ref = np.array(o.compute_sobol_oak(spec, Z, alpha.reshape(-1, 1))[1])
    np.testing.assert_allclose(whole, ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max())


def test_automatic_path_choice(hip):
    """Few terms at a small n stay on the per-term kernel; thousands of terms go to the matrix pipe."""
    rng = np.random.default_rng(9)
    spec = cases.random_spec(rng, 12, 4, kinds=("gaussian",))
    d = _capi.KernelDesc(spec)
    Z = cases.random_inputs(rng, spec, 256)
    alpha = rng.standard_normal(256)
    hip.sobol(d, Z[:64], alpha[:64], [[0], [1, 2]])
    assert hip.sobol_last_info()["path"] == "terms"
    subsets = o.list_representation(12, 4)[1:]
    got = hip.sobol(d, Z, alpha, subsets)
    assert hip.sobol_last_info()["path"] == "gram"
    hip.sobol_set_path("terms")
    try:
        np.testing.assert_allclose(got, hip.sobol(d, Z, alpha, subsets), rtol=1e-9, atol=1e-12 * np.abs(got).max())
    finally:
        hip.sobol_set_path("auto")


@pytest.mark.parametrize("D,depth,expect", [(16, 4, 128), (16, 3, None)])
def test_gram_of_products_column_budget(hip, D, depth, expect):
    """A plan a few columns above a multiple of 128 is trimmed to it (no constant column: order-1 terms evaluated directly;
    a matching of pair columns spared, the affected subsets re-paired) -- D = 16 at depth 4: 135 -> 128 columns, same values."""
    rng = np.random.default_rng(77)
    spec = cases.random_spec(rng, D, depth, kinds=("gaussian", "binary", "gauss2", "categorical"))
    n = 90
    Z = cases.random_inputs(rng, spec, n)
    alpha = rng.standard_normal(n)
    subsets = o.list_representation(D, depth)[1:]
    terms, gram, info = _both_paths(hip, _capi.KernelDesc(spec), Z, alpha, subsets)
    if expect is not None:
        assert info["columns"] == expect
    np.testing.assert_allclose(gram, terms, rtol=1e-9, atol=1e-12 * np.abs(terms).max())
    pick = np.random.default_rng(1).choice(len(subsets), 150, replace=False)
    pick = sorted(set(pick) | set(range(D)))                    # every (directly evaluated) order-1 term among them
    ref = np.array(o.compute_sobol_oak(spec, Z, alpha.reshape(-1, 1), subsets=[subsets[i] for i in pick], L_cache={})[1])
    np.testing.assert_allclose(gram[pick], ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max())
    if depth >= 4:
        assert 0.0 < info["pairing_disagreement"] < 1e-12


def test_L_matrices_match_reference_executed_vectors(hip):
    """oak_sobol_L against the reference's OWN compute_L / compute_L_binary_kernel outputs (tests/golden/reference_sobol_L.npz,
    generated by executing /root/reference/oak/utils.py in the build container), and the package's f1..f4 helpers against the
    reference's."""
    from pathlib import Path
    from oak import utils as oak_utils
    d = np.load(Path(__file__).resolve().parent / "golden" / "reference_sobol_L.npz")
    assert str(d["label"]).startswith("reference-executed")
    for p, ref in zip(d["L_params"], d["L"]):
        desc = _capi.KernelDesc(dict(dims=[dict(type="rbf", lengthscale=float(p[0]), variance=1.0, measure=("gaussian", 0.0, 1.0),
                                                 active_dim=int(p[2]))],
                                     order_variances=[0.0, 1.0], max_interaction_depth=1, share_var_across_orders=True))
        got = hip.sobol_L(desc, 0, float(p[1]), float(p[3]), float(p[4]), d["L_X"])
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
        np.testing.assert_allclose(oak_utils.compute_L(d["L_X"], p[0], p[1], int(p[2]), p[3], p[4]), ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    for p, ref in zip(d["Lb_params"], d["Lb"]):
        got = oak_utils.compute_L_binary_kernel(d["Lb_X"], float(p[0]), float(p[1]), int(p[2]))
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-15)
    for k, fn in enumerate((oak_utils.f1, oak_utils.f2, oak_utils.f3, oak_utils.f4), start=1):
        for p, ref in zip(d["f_params"], d[f"f{k}"]):
            np.testing.assert_allclose(fn(d["f_x"], d["f_y"], *p), ref, rtol=1e-13, atol=0)     # (the helpers associate a few products differently)

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

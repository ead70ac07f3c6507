"""HIP normalising-flow objective (oak_flow_objective) against the oracle, and the reference's own flow test
(/root/reference/tests/test_normalising_flow.py:17-41) on the device-backed Normalizer."""
import numpy as np
import pytest

from oak import gpflow_lite as gpflow
from oak.normalising_flow import Normalizer
from oracle import flow_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,use_log", [(1, False), (100, False), (1000, True), (100003, True), (1 << 20, False)])
def test_flow_objective_and_gradient_match_oracle(hip, n, use_log):
    rng = np.random.default_rng(n)
    g = rng.normal(0.3, 1.2, size=n)
    for (s, b, k, t) in ((1.0, 0.0, 0.0, 1.0), (0.7, -0.4, 0.35, 1.6), (2.5, 0.2, -0.8, 0.6)):
        val, grad = hip.flow_objective(g, n, use_log, s, b, k, t)
        ref = flow_oracle.kl_objective(g, use_log, s, b, k, t)
        assert abs(val - ref) <= 1e-12 * max(1.0, abs(ref))
        if n <= 100003:
            fd = flow_oracle.kl_gradient_fd(g, use_log, s, b, k, t)
            np.testing.assert_allclose(grad, fd, rtol=2e-6, atol=1e-8)
        v2, g2 = hip.flow_objective(None, n, use_log, s, b, k, t)          # resident sample, deterministic
        assert v2 == val and np.array_equal(g2, grad)


def test_flow_objective_argument_errors(hip):
    g = np.zeros(10)
    with pytest.raises(ValueError):
        hip.flow_objective(g, 10, False, -1.0, 0.0, 0.0, 1.0)
    hip.flow_objective(g, 10, False, 1.0, 0.0, 0.0, 1.0)
    with pytest.raises(ValueError):
        hip.flow_objective(None, 11, False, 1.0, 0.0, 0.0, 1.0)          # no resident sample of that length


def test_normalising_flow(hip):
    """tests/test_normalising_flow.py:17-41: the fitted flow maps the sample to N(0, 1)."""
    rng = np.random.default_rng(44)
    x = rng.normal(2, 0.5, size=(100, 1))
    n = Normalizer(x, log=False)
    before = n.KL_objective()
    gpflow.Scipy().minimize(n.KL_objective, n.trainable_variables)
    y = np.asarray(n.bijector(x))
    np.testing.assert_almost_equal(0, y.mean(), decimal=2)
    np.testing.assert_almost_equal(1, y.std(), decimal=2)
    assert n.kstest()[1] > 0.05 and n.KL_objective() < before


def test_two_normalizers_share_the_context(hip):
    """Interleaved objectives of two flows re-upload their samples as needed."""
    rng = np.random.default_rng(1)
    a = Normalizer(np.exp(rng.normal(size=5000)), log=True)
    b = Normalizer(rng.normal(3, 2, size=7000), log=False)
    va, vb = a.KL_objective(), b.KL_objective()
    assert a.KL_objective() == va and b.KL_objective() == vb and a.KL_objective() == va
    ref = flow_oracle.kl_objective(a._g, True, float(a.scale.numpy()), float(a.shift.numpy()), 0.0, 1.0)
    assert abs(va - ref) <= 1e-12 * max(1.0, abs(ref))


def test_flow_fit_at_scale_is_fast_and_gaussianises(hip):
    import time
    rng = np.random.default_rng(3)
    x = np.exp(0.5 * rng.normal(size=1 << 20)) + 0.1
    t0 = time.perf_counter()
    n = Normalizer(x, log=True)
    res = gpflow.Scipy().minimize(n.KL_objective, n.trainable_variables)
    dt = time.perf_counter() - t0
    y = np.asarray(n.bijector(x))
    assert abs(y.mean()) < 0.01 and abs(y.std() - 1) < 0.01
    assert dt < 5.0, f"flow fit took {dt:.1f} s"


def test_flow_forward_matches_numpy_transforms(hip):
    """oak_flow_forward against the NumPy bijectors / scalers column by column (flows with and without log, affine, copy)."""
    rng = np.random.default_rng(8)
    N = 10007
    X = np.column_stack([np.exp(rng.normal(size=N)), rng.normal(3, 2, size=N), rng.integers(0, 2, N).astype(float),
                         rng.normal(-1, 0.3, size=N), rng.integers(0, 5, N).astype(float)])
    f0 = Normalizer(X[:, 0], log=True); f0.skewness.assign(0.2); f0.tailweight.assign(1.3); f0.shift.assign(0.1)
    f1 = Normalizer(X[:, 1], log=False); f1.skewness.assign(-0.4); f1.tailweight.assign(0.7)
    kind = np.array([2, 1, 0, 3, 0], dtype=np.int32)
    params = np.zeros((5, 5))
    for i, f in ((0, f0), (1, f1)):
        params[i] = (f.bijector.offset, float(f.scale.numpy()), float(f.shift.numpy()), float(f.skewness.numpy()), float(f.tailweight.numpy()))
    params[3] = (X[:, 3].mean(), X[:, 3].std(), 0, 0, 0)
    out = hip.flow_forward(X, kind, params)
    np.testing.assert_allclose(out[:, 0], np.asarray(f0.bijector(X[:, 0])), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(out[:, 1], np.asarray(f1.bijector(X[:, 1])), rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(out[:, 2], X[:, 2])
    np.testing.assert_allclose(out[:, 3], (X[:, 3] - X[:, 3].mean()) / X[:, 3].std(), rtol=1e-13, atol=1e-14)
    np.testing.assert_array_equal(out[:, 4], X[:, 4])
    assert hip.flow_forward(X[:0], kind, params).shape == (0, 5)
    with pytest.raises(ValueError):
        hip.flow_forward(X, kind[:3], params)


def test_model_transform_x_matches_host_formulas(hip):
    """oak_model._transform_x (flows on continuous columns, untouched discrete ones) against the NumPy bijectors."""
    from oak.model_utils import oak_model
    rng = np.random.default_rng(5)
    N = 3000
    X = np.column_stack([np.exp(rng.normal(size=N)), rng.integers(0, 2, N).astype(float), rng.normal(size=N)])
    y = (np.log(X[:, 0]) + X[:, 1] + 0.1 * rng.normal(size=N))[:, None]
    m = oak_model(num_inducing=30, binary_feature=[1], sparse=True)
    m.fit(X, y, optimise=False)
    Xt = m._transform_x(X)
    for c in (0, 2):
        np.testing.assert_allclose(Xt[:, c], np.asarray(m.input_flows[c].bijector(X[:, c])), rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(Xt[:, 1], X[:, 1])
    m2 = oak_model(num_inducing=30, binary_feature=[1], sparse=True, use_normalising_flow=False)
    m2.fit(X, y, optimise=False)
    Xt2 = m2._transform_x(X)
    np.testing.assert_allclose(Xt2[:, [0, 2]], (X[:, [0, 2]] - X[:, [0, 2]].mean(0)) / X[:, [0, 2]].std(0), rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(Xt2[:, 1], X[:, 1])

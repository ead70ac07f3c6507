"""Oracle vs the committed golden vectors (tests/golden/oracle_vectors.npz) and C oracle vs NumPy oracle."""
import numpy as np
import pytest

import cases
from conftest import GOLDEN
from oracle import c_oracle, oak_oracle as o


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN / "oracle_vectors.npz")


@pytest.mark.parametrize("name", ["A", "B"])
def test_oracle_reproduces_golden(gold, name):
    spec, X, y, Z, noise = getattr(cases, f"case_{name}")()
    np.testing.assert_array_equal(X, gold[f"{name}_X"])
    rows = 64 if name == "A" else 50
    np.testing.assert_allclose(o.oak_K(spec, X[:rows], Z), gold[f"{name}_K"], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(o.oak_K_diag(spec, X), gold[f"{name}_Kdiag"], rtol=1e-13)
    np.testing.assert_allclose(o.sgpr_elbo(spec, X, y, Z, noise), float(gold[f"{name}_elbo"]), rtol=1e-11)
    np.testing.assert_allclose(o.sgpr_alpha(spec, X, y, Z, noise), gold[f"{name}_alpha"], rtol=1e-6, atol=1e-8)
    start = 300 if name == "A" else 100
    m, v = o.sgpr_predict_f(spec, X, y, Z, noise, X[start:])
    np.testing.assert_allclose(m, gold[f"{name}_mean"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(v, gold[f"{name}_var"], rtol=1e-7, atol=1e-10)


def test_c_oracle_matches_numpy_oracle():
    for name in ("A", "B"):
        spec, X, y, Z, noise = getattr(cases, f"case_{name}")()
        np.testing.assert_allclose(c_oracle.gram(spec, X[:70], Z), o.oak_K(spec, X[:70], Z), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(c_oracle.gram(spec, Z), o.oak_K(spec, Z), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(c_oracle.gram_diag(spec, X), o.oak_K_diag(spec, X), rtol=1e-12)
        e_np = o.sgpr_elbo(spec, X, y, Z, noise)
        np.testing.assert_allclose(c_oracle.sgpr_elbo_chunked(spec, X, y, Z, noise, chunk=37), e_np, rtol=1e-11)


def test_chunking_is_exact_in_exact_arithmetic():
    """All N-dependence of the ELBO is a row sum: different chunkings agree to rounding."""
    spec, X, y, Z, noise = cases.case_A()
    a = c_oracle.sgpr_elbo_chunked(spec, X, y, Z, noise, chunk=384)
    b = c_oracle.sgpr_elbo_chunked(spec, X, y, Z, noise, chunk=50)
    np.testing.assert_allclose(a, b, rtol=1e-12)

"""The many-row triangular solve on its own (oak_bench_trsm): GPflow's `tf.linalg.triangular_solve(L, Kuf)` (oak/utils.py:189)
and the two solves of predict_f.  Checked against extended-precision substitution on sampled rows: the backward error
(residual) must be a small multiple of eps whatever the conditioning, the forward error a small multiple of eps * cond."""
import numpy as np
import pytest

from oak import _capi
from oracle import oak_oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    return _capi.default_context()


def _factor(M, D=4):
    X, y, Z = o.synthetic_problem(9000, D, M, seed=M)
    spec = o.make_spec(D, 2, lengthscales=list(np.linspace(0.8, 1.5, D)))
    Kuu = o.oak_K(spec, Z) + 1e-6 * np.eye(M)
    return spec, X, Z, np.linalg.cholesky(Kuu), np.linalg.cond(Kuu)


# n <= 256: substitution leaf; larger with >= 8192 rows: inverted 128-blocks + MFMA GEMMs (whole blocks 384, ragged 300,
# odd 301 -> staged diagonal product); fewer rows: blocked substitution
@pytest.mark.parametrize("trans", [False, True])
@pytest.mark.parametrize("M,nrhs", [(200, 9000), (300, 9000), (301, 8200), (384, 8192), (640, 8300), (384, 500), (300, 3)])
def test_rows_solve_against_extended_precision(hip, M, nrhs, trans):
    spec, X, Z, L, cond = _factor(M)
    rng = np.random.default_rng(M + nrhs)
    B = o.oak_K(spec, X[:nrhs], Z) if not trans else rng.standard_normal((nrhs, M))
    Xs, _ = hip.bench_trsm(L, B, trans=trans, reps=1)
    rows = rng.choice(nrhs, min(nrhs, 48), replace=False)
    Ll = L.astype(np.longdouble)
    A = Ll if not trans else Ll.T                      # rows x solve A x = b
    Bl = B[rows].astype(np.longdouble)
    ref = np.zeros_like(Bl)
    order = range(M) if not trans else range(M - 1, -1, -1)
    for j in order:                                    # substitution in extended precision
        ref[:, j] = (Bl[:, j] - ref @ A[j, :]) / A[j, j]
    xs = Xs[rows].astype(np.longdouble)
    resid = np.abs(xs @ A.T - Bl).max(axis=1) / np.maximum((np.abs(xs) @ np.abs(A.T)).max(axis=1), 1e-300)
    fwd = np.abs(xs - ref).max(axis=1) / np.abs(ref).max(axis=1)
    assert float(resid.max()) <= 1e-13, f"residual {float(resid.max()):.2e} (cond {cond:.1e})"
    assert float(fwd.max()) <= max(1e-13, 1e-16 * np.sqrt(cond) * 1e3), f"forward error {float(fwd.max()):.2e} (cond(L) {np.sqrt(cond):.1e})"


@pytest.mark.parametrize("trans", [False, True])
def test_every_row_of_a_large_solve(hip, trans):
    """All rows of a 2^18-row solve, three times over.  The sampled-row checks above cannot see a memory-ordering fault in the
    fused kernel's pipeline: the one r03 had touched a few rows in a million and only with the chip full of workgroups.
    trans: the same kernel on column-reversed panels (L^T x = b)."""
    M, nrhs = 512, 1 << 18
    spec, X, Z, L, cond = _factor(M)
    rng = np.random.default_rng(5)
    B = rng.standard_normal((nrhs, M))
    A = L.T if trans else L                            # rows x solve A x = b, i.e. X A^T = B
    X0, _ = hip.bench_trsm(L, B, trans=trans, reps=1)
    resid = np.abs(X0 @ A.T - B).max(axis=1) / np.maximum((np.abs(X0) @ np.abs(A.T)).max(axis=1), 1e-300)
    bad = np.flatnonzero(resid > 1e-13)
    assert bad.size == 0, f"{bad.size} rows above 1e-13, first {bad[:8]}, worst {float(resid.max()):.2e}"
    for rep in range(2):
        Xr, _ = hip.bench_trsm(L, B, trans=trans, reps=1)
        assert np.array_equal(Xr, X0), f"run {rep + 2} differs from run 1 in {int((Xr != X0).any(axis=1).sum())} rows"


@pytest.mark.parametrize("n", [31, 97, 128, 160, 577, 1030, 2050, 4100])
def test_cholesky_factor_itself(n):
    """The two-level blocked Cholesky (32-column panels in 128-column blocks: every pending depth 0 / 32 / 64 / 96 / 128, partial last
    panels and blocks, one to 33 outer blocks) through the full GP, whose entry point returns the factor: L against NumPy's on the
    oracle's Gram matrix, and L L^T against the matrix, entry by entry."""
    from oak import _capi
    from oracle import oak_oracle as o
    rng = np.random.default_rng(n)
    D = 3
    X = rng.standard_normal((n, D))
    y = rng.standard_normal((n, 1))
    spec = o.make_spec(D, 2, lengthscales=[0.7, 1.1, 1.6])
    s2 = 0.05
    ctx = _capi.HipContext(0)
    try:
        ctx.gpr_set_data(X, y)
        ctx.gpr_log_marginal(_capi.KernelDesc(spec), s2)
        L = ctx.gpr_chol(n)
    finally:
        ctx.close()
    A = o.oak_K(spec, X) + s2 * np.eye(n)
    assert np.all(np.triu(L, 1) == 0.0)
    np.testing.assert_allclose(L @ L.T, A, rtol=0, atol=1e-12 * np.abs(A).max())
    Lr = np.linalg.cholesky(A)
    np.testing.assert_allclose(L, Lr, rtol=0, atol=1e-9 * np.abs(Lr).max())     # forward error ~ cond(A) eps; cond(A) <= 1e5 here

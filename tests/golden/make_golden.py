"""Generates tests/golden/oracle_vectors.npz.

The reference (TensorFlow/GPflow) cannot be executed in this environment, so these vectors are ORACLE-DERIVED, not
reference-derived: they pin the oracle (and through it the HIP path) against silent drift, and give the GPU box
known answers without needing mpmath-scale work there.  The oracle itself is pinned by the reference's property
tests and by the mpmath definitional checks in tests/test_oracle_*.py.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import oak_oracle as o  # noqa: E402
import cases  # noqa: E402

OUT = Path(__file__).resolve().parent / "oracle_vectors.npz"


def main():
    out = {}
    # case A: BASELINE synthetic, all-continuous Gaussian-measure OAK (the benchmark's kernel family)
    spec, X, y, Z, _ = cases.case_A()
    out.update(A_X=X, A_y=y, A_Z=Z, A_noise=0.02)
    out["A_K"] = o.oak_K(spec, X[:64], Z)
    out["A_Kdiag"] = o.oak_K_diag(spec, X)
    out["A_elbo"] = o.sgpr_elbo(spec, X, y, Z, 0.02)
    out["A_alpha"] = o.sgpr_alpha(spec, X, y, Z, 0.02)
    m, v = o.sgpr_predict_f(spec, X, y, Z, 0.02, X[300:])
    out["A_mean"], out["A_var"] = m, v
    out["A_gpr_logml"] = o.gpr_log_marginal_likelihood(spec, X[:128], y[:128], 0.02)
    subs, sob = o.compute_sobol_oak(spec, Z, out["A_alpha"])
    out["A_sobol"] = np.array(sob)
    # case B: every sub-kernel type and measure, order 3
    spec, X, y, Z, _ = cases.case_B()
    out.update(B_X=X, B_y=y, B_Z=Z, B_noise=0.05)
    out["B_K"] = o.oak_K(spec, X[:50], Z)
    out["B_Kdiag"] = o.oak_K_diag(spec, X)
    out["B_elbo"] = o.sgpr_elbo(spec, X, y, Z, 0.05)
    out["B_alpha"] = o.sgpr_alpha(spec, X, y, Z, 0.05)
    m, v = o.sgpr_predict_f(spec, X, y, Z, 0.05, X[100:])
    out["B_mean"], out["B_var"] = m, v
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()

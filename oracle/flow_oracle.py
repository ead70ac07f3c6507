"""CPU restatement of the normalising-flow KL objective (TEST INFRASTRUCTURE ONLY; only tests/ may import it).

/root/reference/oak/normalising_flow.py:79-85:
    KL = 0.5 * mean(bijector(x)^2) - mean(bijector.forward_log_det_jacobian(x))
with the TFP chain SinhArcsinh(skewness, tailweight) o Scale(scale) o Shift(shift) [o Log o Shift(-offset)] (:16-55).
TensorFlow Probability is not installable here, so the chain's closed forms are written out; they are pinned by
tests/test_host_logic.py (inverse o forward = id, log-det against a central difference) and by
tests/test_oracle_flow.py (objective against numerical quadrature-free identities and its finite-difference gradient).
"""
import numpy as np


def kl_objective(g, use_log, scale, shift, skewness, tailweight):
    """g = log(x - offset) when use_log else x.  Returns the scalar objective."""
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    z = (g + shift) * scale
    u = (np.arcsinh(z) + skewness) * tailweight
    y = np.sinh(u)
    au = np.abs(u)
    log_cosh = au + np.log1p(np.exp(-2.0 * au)) - np.log(2.0)
    ld = log_cosh + np.log(tailweight) - 0.5 * np.log1p(z * z) + np.log(scale)
    if use_log:
        ld = ld - g                      # d log(x - offset) / dx = 1 / (x - offset) = exp(-g)
    return float(0.5 * np.mean(y * y) - np.mean(ld))


def kl_gradient_fd(g, use_log, scale, shift, skewness, tailweight, h=1e-6):
    """Central differences w.r.t. (scale, shift, skewness, tailweight)."""
    p = np.array([scale, shift, skewness, tailweight], dtype=np.float64)
    out = np.empty(4)
    for i in range(4):
        a, b = p.copy(), p.copy()
        a[i] += h; b[i] -= h
        out[i] = (kl_objective(g, use_log, *a) - kl_objective(g, use_log, *b)) / (2 * h)
    return out

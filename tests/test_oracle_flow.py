"""The flow oracle (oracle/flow_oracle.py, restating oak/normalising_flow.py:16-85 without TensorFlow Probability) pinned
definitionally: the KL objective against its definition written with independent numerics (mpmath, 40 digits), the
log-det term against a central difference of the transform itself, the gradient helper against a second finite-difference
scheme, and the invariances the closed forms must have."""
import mpmath as mp
import numpy as np
import pytest

from oracle import flow_oracle as fo

CASES = [(False, 1.3, -0.2, 0.4, 0.8), (True, 0.7, 0.5, -0.3, 1.4), (False, 1.0, 0.0, 0.0, 1.0), (True, 2.1, -1.0, 0.9, 0.6)]


def _sample(use_log, n=23, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.gamma(2.0, 1.5, n) + 0.3 if use_log else rng.normal(0.5, 1.7, n)
    offset = x.min() - 1.0
    return x, offset, (np.log(x - offset) if use_log else x)


def _forward_mp(x, offset, use_log, scale, shift, skew, tail):
    g = mp.log(x - offset) if use_log else mp.mpf(x)
    z = (g + shift) * scale
    return mp.sinh((mp.asinh(z) + skew) * tail)


@pytest.mark.parametrize("use_log,scale,shift,skew,tail", CASES)
def test_objective_equals_its_definition_in_extended_precision(use_log, scale, shift, skew, tail):
    """KL = mean(y^2) / 2 - mean(log |dy/dx|) with dy/dx taken by mpmath's own differentiation of the chain."""
    mp.mp.dps = 40
    x, offset, g = _sample(use_log)
    ys, lds = [], []
    for xi in x:
        f = lambda t: _forward_mp(t, mp.mpf(offset), use_log, mp.mpf(scale), mp.mpf(shift), mp.mpf(skew), mp.mpf(tail))
        ys.append(f(mp.mpf(xi)))
        lds.append(mp.log(abs(mp.diff(f, mp.mpf(xi)))))
    ref = mp.fsum([y * y for y in ys]) / (2 * len(x)) - mp.fsum(lds) / len(x)
    got = fo.kl_objective(g, use_log, scale, shift, skew, tail)
    assert abs(got - float(ref)) <= 1e-12 * max(1.0, abs(float(ref)))


@pytest.mark.parametrize("use_log,scale,shift,skew,tail", CASES)
def test_gradient_helper_agrees_with_a_richardson_difference(use_log, scale, shift, skew, tail):
    x, offset, g = _sample(use_log, seed=1)
    p = np.array([scale, shift, skew, tail])
    got = fo.kl_gradient_fd(g, use_log, *p)
    for i in range(4):
        def f(h):
            q = p.copy(); q[i] += h
            r = p.copy(); r[i] -= h
            return (fo.kl_objective(g, use_log, *q) - fo.kl_objective(g, use_log, *r)) / (2 * h)
        rich = (4 * f(5e-4) - f(1e-3)) / 3                      # O(h^4)
        assert abs(got[i] - rich) <= 1e-6 * max(1.0, abs(rich))


def test_identity_flow_and_pure_scaling():
    """skewness 0, tailweight 1 make SinhArcsinh the identity: y = scale (x + shift), log-det = log scale."""
    x = np.random.default_rng(2).normal(size=50)
    got = fo.kl_objective(x, False, 2.5, 0.3, 0.0, 1.0)
    y = 2.5 * (x + 0.3)
    assert abs(got - (0.5 * np.mean(y * y) - np.log(2.5))) <= 1e-13
    # a standard-normal sample under the identity flow sits near the entropy-free minimum 1/2
    z = np.random.default_rng(3).normal(size=200000)
    assert abs(fo.kl_objective(z, False, 1.0, 0.0, 0.0, 1.0) - 0.5) < 5e-3

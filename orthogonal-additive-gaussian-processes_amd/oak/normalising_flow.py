"""Per-feature normalising flow (host mirror of oak/normalising_flow.py:16-85).

The reference's TFP chain SinhArcsinh o Scale o Shift o Log o Shift(-offset) is restated with closed-form
log-Jacobians.  The elementwise transforms (forward, inverse, log-det: applied once per feature) are NumPy; the KL
objective that ``oak_model.fit`` minimises with L-BFGS-B -- tens of O(N) evaluations per feature -- and its gradient are
one HIP reduction kernel over the device-resident sample (``oak_flow_objective``, csrc/flow.hip).
"""
from __future__ import annotations

import numpy as np
from scipy import stats

from . import _capi
from .gpflow_lite import Module, Parameter, TensorLike, Transform

try:  # the reference imports pyplot at module level; plotting is optional here
    from matplotlib import pyplot as plt
except Exception:  # pragma: no cover
    plt = None


class Exp(Transform):
    def forward(self, u):
        return np.exp(u)

    def inverse(self, x):
        return np.log(x)

    def dforward(self, u):
        return np.exp(u)


class FlowBijector:
    """y = sinh((asinh(z) + skewness) * tailweight), z = scale * (g(x) + shift), g = log(x - offset) or identity."""

    def __init__(self, owner, log: bool, offset: float):
        self._o, self.log, self.offset = owner, log, offset

    def _inner(self, x):
        x = np.asarray(x, dtype=np.float64)
        g = np.log(x - self.offset) if self.log else x
        return (g + self._o.shift.numpy()) * self._o.scale.numpy()

    def forward(self, x):
        z = self._inner(x)
        return TensorLike(np.sinh((np.arcsinh(z) + self._o.skewness.numpy()) * self._o.tailweight.numpy()))

    __call__ = forward

    def inverse(self, y):
        y = np.asarray(y, dtype=np.float64)
        z = np.sinh(np.arcsinh(y) / self._o.tailweight.numpy() - self._o.skewness.numpy())
        g = z / self._o.scale.numpy() - self._o.shift.numpy()
        return TensorLike(np.exp(g) + self.offset if self.log else g)

    def forward_log_det_jacobian(self, x, event_ndims=0):
        x = np.asarray(x, dtype=np.float64)
        z = self._inner(x)
        tw = self._o.tailweight.numpy()
        a = np.abs((np.arcsinh(z) + self._o.skewness.numpy()) * tw)
        log_cosh = a + np.log1p(np.exp(-2.0 * a)) - np.log(2.0)      # overflow-safe log(cosh(a))
        ld = log_cosh + np.log(tw) - 0.5 * np.log1p(z * z)
        ld = ld + np.log(self._o.scale.numpy())
        if self.log:
            ld = ld - np.log(x - self.offset)
        return ld


class _KLObjective:
    """Callable objective with an analytic gradient (``gpflow_lite.Scipy`` uses ``value_and_grad`` when present, as GPflow's
    Scipy wrapper uses TensorFlow's gradients in the reference)."""
    _ORDER = ("scale", "shift", "skewness", "tailweight")      # gradient layout of oak_flow_objective

    def __init__(self, owner):
        self._o = owner

    def _evaluate(self):
        o = self._o
        ctx = _capi.default_context()
        resident = getattr(ctx, "_flow_owner", None) == o._token      # tokens are never reused (ids can be, after GC)
        val, g = ctx.flow_objective(None if resident else o._g, o._g.size, o.bijector.log, float(o.scale.numpy()),
                                    float(o.shift.numpy()), float(o.skewness.numpy()), float(o.tailweight.numpy()))
        ctx._flow_owner = o._token
        return val, g

    def __call__(self):
        return self._evaluate()[0]

    def value_and_grad(self, variables):
        val, g = self._evaluate()
        by_param = {id(getattr(self._o, name)): g[i] for i, name in enumerate(self._ORDER)}
        grads = []
        for v in variables:
            if id(v) not in by_param:
                raise ValueError("variable does not belong to this Normalizer")
            grads.append(np.asarray(by_param[id(v)] * v.transform.dforward(v.unconstrained_variable)))   # chain to unconstrained
        return val, grads


class Normalizer(Module):
    """Flow that maps the sample `x` towards N(0, 1) (oak/normalising_flow.py:30-85)."""
    _tokens = iter(range(1, 1 << 62))

    def __init__(self, x, log=True, **kwargs):
        self.x = x
        xs = np.asarray(x, dtype=np.float64)
        offset = float(np.min(xs) - 1.0) if log else 0.0
        base = np.log(xs - offset) if log else xs
        self.skewness = Parameter(0.0)
        self.tailweight = Parameter(1.0, transform=Exp())
        self.scale = Parameter(1.0 / np.std(base), transform=Exp())
        self.shift = Parameter(-np.mean(base))
        self.bijector = FlowBijector(self, bool(log), offset)
        self._g = np.ascontiguousarray(base, dtype=np.float64).reshape(-1)     # the sample the objective runs over
        self._token = next(Normalizer._tokens)
        self.KL_objective = _KLObjective(self)

    def _children(self):   # the bijector is a view on the four parameters, not a parameter container
        for key in ("scale", "shift", "skewness", "tailweight"):
            yield key, getattr(self, key)

    def kstest(self):
        s, pvalue = stats.kstest(np.asarray(self.bijector(self.x)).reshape(-1), "norm")
        print("KS test statistic is %.3f, p-value is %.8f" % (s, pvalue))
        return s, pvalue

    def plot(self, title="Normalising Flow"):
        f = plt.figure()
        ax = f.add_axes([0.3, 0.3, 0.65, 0.65])
        x, y = self.x, np.asarray(self.bijector(self.x))
        ax.plot(x, y, "k.", label="Gaussian")
        ax.legend()
        ax_x = f.add_axes([0.3, 0.05, 0.65, 0.25], sharex=ax)
        ax_x.hist(x, bins=20)
        ax_y = f.add_axes([0.05, 0.3, 0.25, 0.65], sharey=ax)
        ax_y.hist(y, bins=20, orientation="horizontal")
        ax_y.set_xlim(ax_y.get_xlim()[::-1])
        plt.title(title)

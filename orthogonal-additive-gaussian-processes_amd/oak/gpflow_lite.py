"""Minimal stand-in for the slice of the GPflow 2.2.1 API that the OAK path touches.

GPflow/TensorFlow are third-party dependencies of the reference (setup.py:13,30,33) that are not part of this
build.  This module supplies only the surface ``oak/model_utils.py`` / ``oak/utils.py`` / the reference tests
use -- ``Parameter`` (``.numpy()/.assign()``, transforms, priors), ``kernels.RBF``, ``likelihoods.Gaussian``,
``InducingPoints``, ``models.GPR/SGPR`` (``elbo``, ``predict_f``, ``predict_log_density``,
``training_loss_closure``), ``optimizers.Scipy`` -- in NumPy on the host, with every Gram / Cholesky / solve
executed by the HIP library through :mod:`oak._capi`.  It is not a general GP framework.
"""
from __future__ import annotations

import numpy as np
import scipy.optimize

from . import _capi

# ------------------------------------------------------------------------------------------------
# config / utilities
# ------------------------------------------------------------------------------------------------
_JITTER = 1e-6


class config:
    @staticmethod
    def default_float():
        return np.float64

    @staticmethod
    def default_jitter():
        return _JITTER


def default_float():
    return np.float64


def default_jitter():
    return _JITTER


class TensorLike(np.ndarray):
    """ndarray that also answers ``.numpy()`` so reference-style code written against tf.Tensor keeps working."""

    def __new__(cls, a):
        return np.asarray(a).view(cls)

    def numpy(self):
        return np.asarray(self)


# ------------------------------------------------------------------------------------------------
# transforms
# ------------------------------------------------------------------------------------------------
class Transform:
    def forward(self, u):
        return u

    def inverse(self, x):
        return x

    def dforward(self, u):
        """d theta / d u, elementwise."""
        return np.ones_like(u)


class Softplus(Transform):
    """gpflow.utilities.positive(lower): theta = lower + log(1 + e^u)."""

    def __init__(self, lower: float = 0.0):
        self.lower = float(lower)

    def forward(self, u):
        return self.lower + np.logaddexp(0.0, u)

    def inverse(self, x):
        x = np.asarray(x, dtype=np.float64) - self.lower
        if np.any(x <= 0):
            raise ValueError("value outside the support of the positive transform")
        return x + np.log(-np.expm1(-x))

    def dforward(self, u):
        return 1.0 / (1.0 + np.exp(-u))


class Sigmoid(Transform):
    """tfp.bijectors.Sigmoid(low, high): theta = low + (high - low) * sigmoid(u)  (oak/oak_kernel.py:24-33)."""

    def __init__(self, low: float, high: float):
        self.low, self.high = float(low), float(high)

    def forward(self, u):
        return self.low + (self.high - self.low) / (1.0 + np.exp(-u))

    def inverse(self, x):
        y = (np.asarray(x, dtype=np.float64) - self.low) / (self.high - self.low)
        if np.any((y <= 0) | (y >= 1)):
            raise ValueError("value outside the bounds of the sigmoid transform")
        return np.log(y) - np.log1p(-y)

    def dforward(self, u):
        s = 1.0 / (1.0 + np.exp(-u))
        return (self.high - self.low) * s * (1.0 - s)


def positive(lower=None):
    return Softplus(0.0 if lower is None else lower)


class Gamma:
    """tfd.Gamma(concentration, rate).log_prob (oak/model_utils.py:165)."""

    def __init__(self, concentration, rate):
        self.concentration, self.rate = float(concentration), float(rate)

    def log_prob(self, x):
        from scipy.special import gammaln
        a, b = self.concentration, self.rate
        return a * np.log(b) - gammaln(a) + (a - 1.0) * np.log(x) - b * x

    def dlog_prob(self, x):
        return (self.concentration - 1.0) / x - self.rate


# ------------------------------------------------------------------------------------------------
# Parameter / Module
# ------------------------------------------------------------------------------------------------
class Parameter:
    """Constrained value with an unconstrained representation (gpflow.Parameter)."""

    def __init__(self, value, transform: Transform = None, prior=None, trainable: bool = True, dtype=None, name=None):
        if isinstance(value, Parameter):
            value = value.numpy()
        self.transform = transform if transform is not None else Transform()
        self.prior = prior
        self.trainable = bool(trainable)
        self.name = name
        self._u = np.array(self.transform.inverse(np.asarray(value, dtype=np.float64)), dtype=np.float64)

    def numpy(self):
        v = self.transform.forward(self._u)
        return np.array(v, dtype=np.float64) if np.ndim(v) else np.float64(v)

    def assign(self, value):
        value = np.asarray(value.numpy() if isinstance(value, Parameter) else value, dtype=np.float64)
        if value.shape != self._u.shape:
            value = np.broadcast_to(value, self._u.shape) if value.size == 1 else value.reshape(self._u.shape)
        self._u = np.array(self.transform.inverse(value), dtype=np.float64)
        return self

    @property
    def unconstrained_variable(self):
        return self._u

    @property
    def shape(self):
        return self._u.shape

    def __float__(self):
        return float(self.numpy())

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.numpy(), dtype=dtype)

    def __mul__(self, other):
        return self.numpy() * other

    __rmul__ = __mul__

    def __repr__(self):
        return f"<Parameter {self.name or ''} value={self.numpy()!r} trainable={self.trainable}>"

    def log_prior_density(self):
        """prior on the constrained value, no Jacobian (gpflow PriorOn.CONSTRAINED, recalled)."""
        if self.prior is None:
            return 0.0
        return float(np.sum(self.prior.log_prob(self.numpy())))


def _as_value(x):
    """numeric value of a Parameter / array / scalar (the reference sometimes overwrites a Parameter with a constant
    tensor, oak_kernel.py:164-166,179,187)."""
    return x.numpy() if isinstance(x, Parameter) else np.asarray(x, dtype=np.float64)


class Module:
    """Parameter container; traversal order mirrors tf.Module (attributes sorted by name, sequences in order)."""

    def _children(self):
        for key in sorted(vars(self)):
            if key.startswith("_"):
                continue
            yield key, getattr(self, key)

    def _walk(self, seen=None, prefix=""):
        seen = set() if seen is None else seen
        if id(self) in seen:
            return
        seen.add(id(self))
        for key, val in self._children():
            yield from _walk_value(val, seen, f"{prefix}.{key}" if prefix else key)

    @property
    def parameters(self):
        return tuple(p for _, p in self._walk())

    @property
    def trainable_parameters(self):
        return tuple(p for p in self.parameters if p.trainable)

    @property
    def trainable_variables(self):
        return tuple(p for p in self.parameters if p.trainable)

    def named_parameters(self):
        return list(self._walk())


def _walk_value(val, seen, path):
    if isinstance(val, Parameter):
        if id(val) not in seen:
            seen.add(id(val))
            yield path, val
    elif isinstance(val, Module):
        yield from val._walk(seen, path)
    elif isinstance(val, (list, tuple)):
        for i, v in enumerate(val):
            yield from _walk_value(v, seen, f"{path}[{i}]")


def set_trainable(obj, flag: bool):
    if isinstance(obj, Parameter):
        obj.trainable = bool(flag)
    else:
        for p in obj.parameters:
            p.trainable = bool(flag)


def print_summary(module, fmt=None):
    rows = [(name, type(p.transform).__name__, p.trainable, np.array2string(np.asarray(p.numpy()), precision=5))
            for name, p in module.named_parameters()]
    width = max([len(r[0]) for r in rows] + [4])
    print(f"{'name'.ljust(width)}  transform  trainable  value")
    for r in rows:
        print(f"{r[0].ljust(width)}  {r[1]:<9}  {str(r[2]):<9}  {r[3]}")


def to_default_float(x):
    return np.asarray(x, dtype=np.float64)


class utilities:
    positive = staticmethod(positive)
    print_summary = staticmethod(print_summary)
    to_default_float = staticmethod(to_default_float)
    set_trainable = staticmethod(set_trainable)


# ------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------
class Kernel(Module):
    """gpflow.kernels.Kernel protocol: K, K_diag, __call__(X, X2, full_cov, presliced), active_dims slicing."""

    def __init__(self, active_dims=None, name=None):
        if active_dims is None:
            self._active_dims = slice(None)
        elif isinstance(active_dims, slice):
            self._active_dims = active_dims
        else:
            self._active_dims = np.array(list(active_dims), dtype=int)
        self.name = name

    @property
    def active_dims(self):
        return self._active_dims

    @active_dims.setter
    def active_dims(self, value):
        if value is None:
            value = slice(None)
        self._active_dims = value if isinstance(value, slice) else np.array(list(value), dtype=int)

    def slice(self, X, X2=None):
        dims = self._active_dims
        X = np.asarray(X, dtype=np.float64)
        if X2 is not None:
            X2 = np.asarray(X2, dtype=np.float64)
        if isinstance(dims, slice):
            return X[..., dims], (None if X2 is None else X2[..., dims])
        return X[..., dims], (None if X2 is None else X2[..., dims])

    def K(self, X, X2=None):
        raise NotImplementedError

    def K_diag(self, X):
        raise NotImplementedError

    def __call__(self, X, X2=None, *, full_cov=True, presliced=False):
        if (not full_cov) and (X2 is not None):
            raise ValueError("Ambiguous inputs: `not full_cov` and `X2` are not compatible.")
        if not presliced:
            X, X2 = self.slice(X, X2)
        if not full_cov:
            return self.K_diag(X)
        return self.K(X, X2)


def _ctx():
    return _capi.default_context()


def _active_communicator():
    """The job's communicator when this process is one rank of several (oak.distributed.init_from_env), else None."""
    from . import distributed
    comm = distributed.current()
    return comm if (comm is not None and comm.active) else None


def _shard_rows(comm, hip, X, Y):
    """This rank's contiguous row block of (X, Y) and the bookkeeping that goes with it: the context joins the communicator
    (collective) and learns the row count of the whole problem (what the auto route's size rule looks at)."""
    if comm is None:
        return X, Y
    lo, hi = comm.bounds(len(X))
    if getattr(hip, "_oak_comm_attached", None) is not comm:
        comm.attach(hip)
        hip._oak_comm_attached = comm
    hip.sgpr_set_global_rows(len(X))
    return X[lo:hi], Y[lo:hi]


class RBF(Kernel):
    """gpflow.kernels.RBF (SquaredExponential, isotropic lengthscale): variance * exp(-|x - z|^2 / (2 l^2))."""

    def __init__(self, variance=1.0, lengthscales=1.0, active_dims=None, name=None):
        super().__init__(active_dims=active_dims, name=name)
        self.variance = Parameter(variance, transform=positive())
        self.lengthscales = Parameter(lengthscales, transform=positive())

    def _spec(self, ncols):
        """product of ncols one-dimensional RBFs == one isotropic RBF: K = e_ncols(k_1..k_ncols)."""
        if ncols > _capi.MAX_DEPTH:
            raise NotImplementedError(f"stand-alone RBF over {ncols} > {_capi.MAX_DEPTH} columns is not supported by the HIP path")
        l = float(np.asarray(_as_value(self.lengthscales)).reshape(-1)[0])
        v = float(np.asarray(_as_value(self.variance)).reshape(-1)[0])
        dims = [dict(type="rbf", lengthscale=l, variance=(v if d == 0 else 1.0), measure=None, active_dim=d) for d in range(ncols)]
        return dict(dims=dims, order_variances=[0.0] * ncols + [1.0], max_interaction_depth=ncols, share_var_across_orders=True)

    def K(self, X, X2=None):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        desc = _capi.KernelDesc(self._spec(X.shape[1]))
        return TensorLike(_ctx().gram(desc, X, None if X2 is None else np.atleast_2d(np.asarray(X2, dtype=np.float64))))

    def K_diag(self, X):
        v = float(np.asarray(_as_value(self.variance)).reshape(-1)[0])
        return TensorLike(np.full(np.asarray(X).shape[0], v))


SquaredExponential = RBF


class kernels:
    Kernel = Kernel
    RBF = RBF
    SquaredExponential = RBF


# ------------------------------------------------------------------------------------------------
# likelihood / inducing variables
# ------------------------------------------------------------------------------------------------
class Gaussian(Module):
    """gpflow.likelihoods.Gaussian: variance with softplus + 1e-6 lower bound (recalled)."""

    DEFAULT_VARIANCE_LOWER_BOUND = 1e-6

    def __init__(self, variance=1.0, variance_lower_bound=DEFAULT_VARIANCE_LOWER_BOUND):
        self.variance = Parameter(variance, transform=positive(lower=variance_lower_bound))

    def predict_log_density(self, Fmu, Fvar, Y):
        var = Fvar + self.variance.numpy()
        return np.sum(-0.5 * (np.log(2 * np.pi) + np.log(var) + np.square(Fmu - np.asarray(Y)) / var), axis=-1)


def inv_logit(x, jitter=1e-3):
    """The classification example's link (examples/uci/uci_classification_train.py:43-45): sigmoid(x)(1 - 2 jitter) + jitter."""
    x = np.asarray(x, dtype=np.float64)
    return TensorLike(1.0 / (1.0 + np.exp(-x)) * (1 - 2 * jitter) + jitter)


def inv_probit(x, jitter=1e-3):
    """gpflow.likelihoods.utils.inv_probit: Phi(x)(1 - 2 jitter) + jitter."""
    from scipy.special import erf
    x = np.asarray(x, dtype=np.float64)
    return TensorLike(0.5 * (1.0 + erf(x / np.sqrt(2.0))) * (1 - 2 * jitter) + jitter)


inv_logit._oak_link = ("logit", 1e-3)
inv_probit._oak_link = ("probit", 1e-3)


class Bernoulli(Module):
    """gpflow.likelihoods.Bernoulli(invlink): log p(y | f) = log(where(y == 1, p, 1 - p)), p = invlink(f); expectations
    by 20-node Gauss-Hermite quadrature.  The quadrature runs inside the device kernels, which know two links:
    :func:`inv_logit` (the reference's classification example) and :func:`inv_probit` (GPflow's default)."""

    num_gauss_hermite_points = 20

    def __init__(self, invlink=inv_probit):
        link = getattr(invlink, "_oak_link", None)
        if link is None:
            raise NotImplementedError("Bernoulli: pass gpflow_lite.inv_logit or gpflow_lite.inv_probit as the inverse link")
        self.invlink = invlink
        self._link, self._link_eps = link


class likelihoods:
    Gaussian = Gaussian
    Bernoulli = Bernoulli


class InducingPoints(Module):
    def __init__(self, Z, name=None):
        self.Z = Parameter(np.asarray(Z, dtype=np.float64))
        self.name = name

    def __len__(self):
        return self.Z.shape[0]

    @property
    def num_inducing(self):
        return self.Z.shape[0]


class inducing_variables:
    InducingPoints = InducingPoints


# ------------------------------------------------------------------------------------------------
# models
# ------------------------------------------------------------------------------------------------
class _LossClosure:
    """Callable returned by ``training_loss_closure``; optimizers use ``value_and_grad`` when available."""

    def __init__(self, model):
        self.model = model

    def __call__(self):
        return self.model.training_loss()

    def value_and_grad(self, variables):
        return self.model._training_loss_and_grad(variables)


class GPModel(Module):
    def __init__(self, data, kernel, mean_function=None, noise_variance=1.0):
        X, Y = data
        self.data = (np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64).reshape(len(X), -1))
        # P output columns are GPflow's independent outputs sharing kernel and noise: bound, gradient and posterior mean are sums /
        # stacks over single-output problems.  The sparse model evaluates them in ONE device pass (the y-independent statistics --
        # Kuf panel, Phi, both Cholesky factors, the M x M adjoints -- are shared: oak_sgpr_set_extra_targets); the full GP, a
        # small problem by construction, runs one pass per column (`_outputs`).  The reference itself only ever passes one column.
        self._P = self.data[1].shape[1]
        if self._P < 1:
            raise ValueError("Y has no columns")
        self._col = 0                                                  # the target column currently on the device
        if mean_function is not None:
            raise NotImplementedError("only the zero mean function is supported (model_utils.py:152,159 pass None)")
        self.kernel = kernel
        self.likelihood = Gaussian(noise_variance)
        self._hip = _capi.HipContext(_capi.default_context().device)   # own device state (data stay resident)
        self._comm = _active_communicator()                            # row sharding (oak/distributed.py); None: single process

    def mean_function(self, X):
        return np.zeros((np.asarray(X).shape[0], self._P))

    def _set_column(self, p):                                          # subclasses: put target column p on the device
        raise NotImplementedError

    def _outputs(self):
        """Iterate over the output columns, each one on the device while the caller's loop body runs."""
        for p in range(self._P):
            if p != self._col:
                self._set_column(p)
                self._col = p
            yield p

    # -- kernel -> POD description ---------------------------------------------------------------
    def _spec(self):
        from .oak_kernel import kernel_to_spec
        return kernel_to_spec(self.kernel)

    def _desc(self):
        return _capi.KernelDesc(self._spec())

    # -- objective ---------------------------------------------------------------------------------
    def log_prior_density(self):
        return float(sum(p.log_prior_density() for p in self.trainable_parameters))

    def log_posterior_density(self):
        return self.maximum_log_likelihood_objective() + self.log_prior_density()

    def training_loss(self):
        return -(self.maximum_log_likelihood_objective() + self.log_prior_density())

    def training_loss_closure(self, compile=True):
        return _LossClosure(self)

    def predict_log_density(self, data, full_cov=False, full_output_cov=False):
        X, Y = data
        mean, var = self.predict_f(X)
        return self.likelihood.predict_log_density(mean, var, np.asarray(Y, dtype=np.float64).reshape(len(mean), -1))

    def predict_y(self, Xnew):
        mean, var = self.predict_f(Xnew)
        return mean, var + self.likelihood.variance.numpy()

    # -- gradients w.r.t. the unconstrained trainable variables -----------------------------------------
    def _objective_and_constrained_grad(self):
        raise NotImplementedError

    def _training_loss_and_grad(self, variables):
        """loss and d loss / d u for each variable in ``variables`` (analytic, HIP backward pass)."""
        from .oak_kernel import scatter_gradient
        self._want_extra = variables
        obj, gvec, desc = self._objective_and_constrained_grad()
        grads = scatter_gradient(self.kernel, self.likelihood, desc, gvec, variables)
        extra = getattr(self, "_extra_grads", {})                    # e.g. the inducing inputs (SGPR, zfixed=False)
        grads = [extra.get(id(p), g) if g is None else g for p, g in zip(variables, grads)]
        loss = -(obj + self.log_prior_density())
        out = []
        for p, g in zip(variables, grads):
            if g is None:
                raise NotImplementedError(f"no analytic gradient for parameter {p!r}")
            g = np.asarray(g, dtype=np.float64).reshape(p.shape)
            if p.prior is not None and hasattr(p.prior, "dlog_prob"):
                g = g + p.prior.dlog_prob(p.numpy())
            out.append(-(g * p.transform.dforward(p.unconstrained_variable)))
        return loss, out


class GPR(GPModel):
    """gpflow.models.GPR (constructed at oak/model_utils.py:159)."""

    def __init__(self, data, kernel, mean_function=None, noise_variance=1.0):
        super().__init__(data, kernel, mean_function, noise_variance)
        self._hip.gpr_set_data(self.data[0], self.data[1][:, 0])

    def _set_column(self, p):
        self._hip.gpr_set_targets(self.data[1][:, p])

    def log_marginal_likelihood(self):
        desc, s2 = self._desc(), float(self.likelihood.variance.numpy())
        return float(sum(self._hip.gpr_log_marginal(desc, s2) for _ in self._outputs()))

    def maximum_log_likelihood_objective(self):
        return self.log_marginal_likelihood()

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        if full_cov:
            raise NotImplementedError("full_cov=True is not on the OAK path")
        desc, s2 = self._desc(), float(self.likelihood.variance.numpy())
        Xnew = np.asarray(Xnew, dtype=np.float64)
        means, var = [], None
        for _ in self._outputs():
            self._hip.gpr_log_marginal(desc, s2)
            mean, var = self._hip.gpr_predict(desc, Xnew)
            means.append(mean)
        return TensorLike(np.stack(means, axis=1)), TensorLike(np.tile(var[:, None], (1, self._P)))

    def alpha(self):
        """cholesky_solve(L, Y) of oak/utils.py:206-211 (one column per output)."""
        desc, s2, cols = self._desc(), float(self.likelihood.variance.numpy()), []
        for _ in self._outputs():
            self._hip.gpr_log_marginal(desc, s2)
            cols.append(self._hip.gpr_alpha(self.data[0].shape[0]))
        return TensorLike(np.stack(cols, axis=1))

    def effective_L(self):
        """chol(K + noise I) of oak/utils.py:206-211."""
        self._hip.gpr_log_marginal(self._desc(), float(self.likelihood.variance.numpy()))
        return self._hip.gpr_chol(self.data[0].shape[0])

    def _objective_and_constrained_grad(self):
        desc, s2 = self._desc(), float(self.likelihood.variance.numpy())
        obj, g = 0.0, 0.0
        for _ in self._outputs():
            o_p, g_p = self._hip.gpr_log_marginal_grad(desc, s2)
            obj, g = obj + o_p, g + g_p
        return obj, g, desc


class SGPR(GPModel):
    """gpflow.models.SGPR (constructed at oak/model_utils.py:149-157): collapsed Titsias bound."""

    SHARDED_PREDICT_MIN_ROWS = 4096      # per rank: below it a replicated prediction is cheaper than the gather

    def __init__(self, data, kernel, inducing_variable, mean_function=None, noise_variance=1.0, num_latent_gps=None):
        super().__init__(data, kernel, mean_function, noise_variance)
        if not isinstance(inducing_variable, InducingPoints):
            inducing_variable = InducingPoints(inducing_variable)
        self.inducing_variable = inducing_variable
        # under a communicator every rank is handed the same (X, Y) and keeps its contiguous row block on the device: the
        # statistics and the gradient record are summed over the ranks inside the library, the O(M^3) tail is replicated, so
        # objective and gradient are bit-identical on all ranks and an optimiser simply runs replicated
        Xl, Yl = _shard_rows(self._comm, self._hip, self.data[0], self.data[1])
        self._Yl = Yl                                                  # this rank's rows, all output columns
        self._hip.sgpr_set_data(Xl, Yl[:, 0])
        if self._P > 1:
            self._hip.sgpr_set_extra_targets(Yl[:, 1:])                # outputs 1 .. P-1 share every y-independent statistic
        self._z_sent = None
        self.route = "auto"

    def _posteriors(self):
        """Iterate over the outputs with that output's posterior selected on the device (one evaluation serves all of them)."""
        for p in range(self._P):
            self._hip.sgpr_select_output(p)
            yield p
        if self._P > 1:
            self._hip.sgpr_select_output(0)

    def _sync_Z(self):
        Z = self.inducing_variable.Z.numpy()
        if self._z_sent is None or self._z_sent.shape != Z.shape or not np.array_equal(self._z_sent, Z):
            self._hip.sgpr_set_inducing(Z)
            self._z_sent = Z.copy()
        self._hip.sgpr_set_route(self.route)

    def elbo(self):
        self._sync_Z()
        desc, s2 = self._desc(), float(self.likelihood.variance.numpy())
        return float(self._hip.sgpr_elbo(desc, s2, default_jitter()))   # the sum over the output columns

    def maximum_log_likelihood_objective(self):
        return self.elbo()

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        if full_cov:
            raise NotImplementedError("full_cov=True is not on the OAK path")
        self._sync_Z()
        desc, s2 = self._desc(), float(self.likelihood.variance.numpy())
        Xnew = np.asarray(Xnew, dtype=np.float64)
        means, var = [], None
        self._hip.sgpr_elbo(desc, s2, default_jitter())
        for _ in self._posteriors():
            if self._comm is not None and len(Xnew) >= self.SHARDED_PREDICT_MIN_ROWS * self._comm.world:
                # test rows are independent: each rank predicts its block from the replicated posterior, one gather
                from . import distributed
                mean, var = distributed.sharded_predict(self._hip, desc, Xnew, self._comm.rank, self._comm.world, comm=self._comm)
            else:
                mean, var = self._hip.sgpr_predict(desc, Xnew)
            means.append(mean)
        return TensorLike(np.stack(means, axis=1)), TensorLike(np.tile(var[:, None], (1, self._P)))

    def alpha(self):
        """alpha of oak/utils.py:180-198 (one column per output)."""
        self._sync_Z()
        desc, s2, cols = self._desc(), float(self.likelihood.variance.numpy()), []
        self._hip.sgpr_elbo(desc, s2, default_jitter())
        for _ in self._posteriors():
            cols.append(self._hip.sgpr_alpha(len(self.inducing_variable)))
        return TensorLike(np.stack(cols, axis=1))

    def effective_L(self):
        """inv(L^-1 - LB^-1 L^-1) of oak/utils.py:199-204."""
        self.elbo()
        return self._hip.sgpr_effective_L(len(self.inducing_variable))

    def _objective_and_constrained_grad(self):
        self._sync_Z()
        desc = self._desc()
        Zp = self.inducing_variable.Z
        want_z = any(v is Zp for v in getattr(self, "_want_extra", ()))
        self._extra_grads = {}
        s2 = float(self.likelihood.variance.numpy())
        # one call: bound and gradient summed over the output columns (shared statistics, oak_sgpr_set_extra_targets)
        if want_z:  # trainable inducing inputs (create_model_oak(zfixed=False)): one more pass over the pairs
            Z = Zp.numpy()
            obj, g, gz = self._hip.sgpr_elbo_grad_z(desc, s2, Z.shape[0], Z.shape[1], default_jitter())
            self._extra_grads[id(Zp)] = gz
        else:
            obj, g = self._hip.sgpr_elbo_grad(desc, s2, default_jitter())
        return obj, g, desc


class _SVGPPosterior:
    """The two members of gpflow's posterior object that oak/utils.py:174-179 reads."""

    def __init__(self, model):
        self._m = model

    @property
    def alpha(self):
        return TensorLike(self._m._posterior(get_L=False)[:, None])

    @property
    def Qinv(self):
        """[1, M, M]: Lm^-T (I - diag(q_sqrt^2)) Lm^-1, rebuilt from the factor the device returns (inv(Qinv) = L L^T)."""
        _, L = self._m._posterior(get_L=True)
        Linv = np.linalg.inv(L)
        return TensorLike((Linv.T @ Linv)[None])


class SVGP(Module):
    """gpflow.models.SVGP as the classification example builds it (examples/uci/uci_classification_train.py:108-116):
    ``whiten=True, q_diag=True``, one latent, Bernoulli likelihood, full-batch data handed to ``elbo`` /
    ``training_loss_closure``.  Other configurations are not on the OAK path and raise NotImplementedError."""

    def __init__(self, kernel, likelihood, inducing_variable, *, mean_function=None, num_latent_gps=1, q_diag=False,
                 q_mu=None, q_sqrt=None, whiten=True, num_data=None):
        if not (whiten and q_diag) or num_latent_gps != 1 or mean_function is not None:
            raise NotImplementedError("SVGP: only whiten=True, q_diag=True, one latent, zero mean (the reference's use)")
        if not isinstance(likelihood, Bernoulli):
            raise NotImplementedError("SVGP: only the Bernoulli likelihood is on the OAK path")
        if num_data is not None:
            raise NotImplementedError("SVGP: minibatch scaling (num_data) is not used by the reference")
        self.kernel = kernel
        self.likelihood = likelihood
        if not isinstance(inducing_variable, InducingPoints):
            inducing_variable = InducingPoints(inducing_variable)
        self.inducing_variable = inducing_variable
        M = len(inducing_variable)
        self.q_mu = Parameter(np.zeros((M, 1)) if q_mu is None else np.asarray(q_mu, dtype=np.float64).reshape(M, 1))
        self.q_sqrt = Parameter(np.ones((M, 1)) if q_sqrt is None else np.asarray(q_sqrt, dtype=np.float64).reshape(M, 1),
                                transform=positive())
        self.data = None                 # the example assigns ``model.data`` before asking for Sobol indices (:150)
        self._hip = _capi.HipContext(_capi.default_context().device)
        self._comm = _active_communicator()
        self._z_sent = self._data_sent = self._data_obj = None

    # -- device state -----------------------------------------------------------------------------
    def _spec(self):
        from .oak_kernel import kernel_to_spec
        return kernel_to_spec(self.kernel)

    def _desc(self):
        return _capi.KernelDesc(self._spec())

    def _sync_Z(self):
        Z = self.inducing_variable.Z.numpy()
        if self._z_sent is None or self._z_sent.shape != Z.shape or not np.array_equal(self._z_sent, Z):
            self._hip.sgpr_set_inducing(Z)
            self._z_sent = Z.copy()

    def _sync_data(self, data):
        if self._data_obj is not None and data[0] is self._data_obj[0] and data[1] is self._data_obj[1]:
            return                       # the closure hands over the same arrays at every evaluation (treated as immutable)
        self._data_obj = (data[0], data[1])
        X = np.asarray(data[0], dtype=np.float64)
        Y = np.asarray(data[1], dtype=np.float64).reshape(len(X), -1)
        if Y.shape[1] != 1:
            raise NotImplementedError("the HIP path supports a single output column")
        key = self._data_sent
        if key is None or key[0].shape != X.shape or not (np.array_equal(key[0], X) and np.array_equal(key[1], Y)):
            Xl, Yl = _shard_rows(self._comm, self._hip, X, Y)           # row block of this rank under a communicator
            self._hip.sgpr_set_data(Xl, Yl)
            self._data_sent = (X.copy(), Y.copy())
            if self._z_sent is not None and self._z_sent.shape[1] != X.shape[1]:
                self._z_sent = None          # the library drops inducing inputs of another column count

    def _q(self):
        return self.q_mu.numpy().reshape(-1), self.q_sqrt.numpy().reshape(-1)

    def _lik(self):
        return dict(link=self.likelihood._link, link_eps=self.likelihood._link_eps, jitter=default_jitter(),
                    n_gh=self.likelihood.num_gauss_hermite_points)

    # -- objective -----------------------------------------------------------------------------------
    def prior_kl(self):
        m, s = self._q()
        return float(0.5 * (np.sum(m * m) - m.size - np.sum(np.log(s * s)) + np.sum(s * s)))

    def elbo(self, data):
        self._sync_data(data)
        self._sync_Z()
        return self._hip.svgp_elbo(self._desc(), *self._q(), **self._lik())

    def maximum_log_likelihood_objective(self, data):
        return self.elbo(data)

    def log_prior_density(self):
        return float(sum(p.log_prior_density() for p in self.trainable_parameters))

    def training_loss(self, data):
        return -(self.elbo(data) + self.log_prior_density())

    def training_loss_closure(self, data, compile=True):
        model = self

        class _Closure:
            def __call__(self):
                return model.training_loss(data)

            def value_and_grad(self, variables):
                return model._training_loss_and_grad(data, variables)

        return _Closure()

    def _training_loss_and_grad(self, data, variables):
        from .oak_kernel import scatter_gradient
        self._sync_data(data)
        self._sync_Z()
        desc = self._desc()
        obj, gvec, gm, gs = self._hip.svgp_elbo(desc, *self._q(), grad=True, **self._lik())
        grads = scatter_gradient(self.kernel, self.likelihood, desc, gvec, variables)
        extra = {id(self.q_mu): gm, id(self.q_sqrt): gs}
        loss = -(obj + self.log_prior_density())
        out = []
        for p, g in zip(variables, grads):
            g = extra.get(id(p), g)
            if g is None:
                raise NotImplementedError(f"no analytic gradient for parameter {p!r}")
            g = np.asarray(g, dtype=np.float64).reshape(p.shape)
            if p.prior is not None and hasattr(p.prior, "dlog_prob"):
                g = g + p.prior.dlog_prob(p.numpy())
            out.append(-(g * p.transform.dforward(p.unconstrained_variable)))
        return loss, out

    # -- predictions -----------------------------------------------------------------------------------
    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        if full_cov:
            raise NotImplementedError("full_cov=True is not on the OAK path")
        self._sync_Z()
        mean, var = self._hip.svgp_predict(self._desc(), *self._q(), np.asarray(Xnew, dtype=np.float64), jitter=default_jitter())
        return TensorLike(mean[:, None]), TensorLike(var[:, None])

    def predict_log_density(self, data, full_cov=False, full_output_cov=False):
        X, Y = data
        self._sync_Z()
        _, _, ld = self._hip.svgp_predict(self._desc(), *self._q(), np.asarray(X, dtype=np.float64),
                                          np.asarray(Y, dtype=np.float64).reshape(-1), **self._lik())
        return TensorLike(ld)

    def _posterior(self, get_L=True):
        self._sync_Z()
        return self._hip.svgp_posterior(self._desc(), *self._q(), jitter=default_jitter(), get_L=get_L)

    def posterior(self, precompute_cache=None):
        return _SVGPPosterior(self)

    def alpha(self):
        return TensorLike(self._posterior(get_L=False)[:, None])

    def effective_L(self):
        """chol(inv(posterior.Qinv[0])) of oak/utils.py:174-179."""
        return self._posterior(get_L=True)[1]


class models:
    GPR = GPR
    SGPR = SGPR
    SVGP = SVGP
    GPModel = GPModel
    BayesianModel = GPModel


# ------------------------------------------------------------------------------------------------
# optimizer
# ------------------------------------------------------------------------------------------------
class Scipy:
    """gpflow.optimizers.Scipy: scipy.optimize.minimize over the packed unconstrained variables."""

    def minimize(self, closure, variables, method="L-BFGS-B", step_callback=None, compile=True, on_linalg_error="raise",
                 **scipy_kwargs):
        """``on_linalg_error="raise"`` (default) lets a failed Cholesky at a trial point abort the optimisation, as the
        reference does (TensorFlow raises InvalidArgumentError through GPflow's Scipy wrapper; callers wrap ``fit`` in
        try/except, examples/uci/uci_classification_train.py:146-159).  ``"inf"`` is an extension: the trial point is
        reported to scipy as +inf with a zero gradient so that the line search backtracks instead."""
        if on_linalg_error not in ("raise", "inf"):
            raise ValueError("on_linalg_error must be 'raise' or 'inf'")
        variables = tuple(variables)
        if not variables:
            raise ValueError("no variables to optimise")
        shapes = [v.shape for v in variables]
        sizes = [int(np.prod(s)) if s else 1 for s in shapes]

        def unpack(x):
            pos = 0
            for v, s, n in zip(variables, shapes, sizes):
                v._u = np.array(x[pos:pos + n].reshape(s), dtype=np.float64)
                pos += n

        x0 = np.concatenate([np.asarray(v.unconstrained_variable, dtype=np.float64).reshape(-1) for v in variables])
        has_grad = hasattr(closure, "value_and_grad")

        def fun(x):
            unpack(x)
            try:
                if has_grad:
                    loss, grads = closure.value_and_grad(variables)
                    return float(loss), np.concatenate([np.asarray(g, dtype=np.float64).reshape(-1) for g in grads])
                return float(closure())
            except _capi.NotPositiveDefiniteError:
                if on_linalg_error == "raise":
                    raise
                return (np.inf, np.zeros_like(x)) if has_grad else np.inf

        res = scipy.optimize.minimize(fun, x0, jac=True if has_grad else None, method=method, **scipy_kwargs)
        unpack(res.x)
        return res


class optimizers:
    Scipy = Scipy

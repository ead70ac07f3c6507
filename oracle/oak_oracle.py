"""fp64 NumPy/SciPy restatement of the reference's OAK Gram -> SGPR/GPR -> Sobol path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference file:line (relative to the upstream repo root) whose arithmetic it
restates, in the reference's own operation order: D materialised per-dimension
matrices -> power sums -> Newton-Girard -> weighted sum, and GPflow's SGPR
algebra (L, A, AAT, B, LB, c).  GPflow / TensorFlow internals that are not in the
reference tree are restated from GPflow 2.2.1 as recalled in SURVEY.md section 8
(rows a7-a11, a14) and are marked "gpflow 2.2.1 (recalled)".

PARITY UNPINNED against reference-executed outputs for the ELBO scalar, predictive variance, GPR
log-marginal and the prior / transform conventions (TensorFlow / GPflow cannot be installed here, and
the reference's own tests hold no numbers for them); the Gram entries, alpha / predictive mean and the
Sobol indices ARE pinned by the reference's exact property tests (tests/test_oracle_reference_properties.py,
tests/test_oracle_sobol.py), everything else by 50-digit definitional restatements
(tests/test_oracle_definitional.py).  See oracle/__init__.py and DESIGN.md section 3.

Kernel description used throughout (``spec``): a plain dict
    {"dims": [dim_spec, ...], "order_variances": [s0..sR] (or [s0] when not shared),
     "max_interaction_depth": R, "share_var_across_orders": bool}
with dim_spec one of
    {"type": "rbf", "lengthscale": l, "variance": s2,
     "measure": None | ("gaussian", mu, var) | ("uniform", a, b)
                | ("empirical", loc[K,1], w[K,1]) | ("mog", means[K], vars[K], w[K])}
    {"type": "binary", "p0": p0, "variance": s2}
    {"type": "categorical", "p": p[C,1], "W": W[C,r], "kappa": kappa[C], "variance": s2}
"""
from __future__ import annotations

import itertools
from functools import reduce
from math import erf as _erf

import numpy as np
import scipy.linalg as sla
from scipy.special import erf

JITTER = 1e-6  # gpflow.config.default_jitter()  (gpflow 2.2.1, recalled)


# --------------------------------------------------------------------------------------
# parameter transforms
# --------------------------------------------------------------------------------------
def softplus(u):
    """gpflow.utilities.positive() forward (tfp Softplus): log(1+e^u)."""
    u = np.asarray(u, dtype=np.float64)
    return np.logaddexp(0.0, u)


def softplus_inv(x):
    x = np.asarray(x, dtype=np.float64)
    return x + np.log(-np.expm1(-x))


def sigmoid_bounded(u, low, high):
    """oak/oak_kernel.py:24-33 -- tfb.Sigmoid(low, high).forward."""
    u = np.asarray(u, dtype=np.float64)
    return low + (high - low) / (1.0 + np.exp(-u))


def sigmoid_bounded_inv(x, low, high):
    y = (np.asarray(x, dtype=np.float64) - low) / (high - low)
    return np.log(y) - np.log1p(-y)


# --------------------------------------------------------------------------------------
# base RBF (gpflow.kernels.RBF, gpflow 2.2.1 recalled): variance*exp(-0.5*square_distance)
# --------------------------------------------------------------------------------------
def rbf_K(X, X2, lengthscale, variance):
    """gpflow SquaredExponential.K: scaled square_distance in the |x|^2+|z|^2-2xz form."""
    Xs = np.asarray(X, dtype=np.float64) / lengthscale
    if X2 is None:
        sq = np.sum(Xs * Xs, axis=-1, keepdims=True)
        r2 = -2.0 * Xs @ Xs.T + sq + sq.T
    else:
        X2s = np.asarray(X2, dtype=np.float64) / lengthscale
        r2 = -2.0 * Xs @ X2s.T + np.sum(Xs * Xs, -1)[:, None] + np.sum(X2s * X2s, -1)[None, :]
    return variance * np.exp(-0.5 * r2)


def rbf_K_diag(X, variance):
    return np.full(np.asarray(X).shape[0], float(variance))


# --------------------------------------------------------------------------------------
# constrained RBF: cov_X_s / var_s for the four measures (oak/ortho_rbf_kernel.py:47-152)
# --------------------------------------------------------------------------------------
def cov_X_s(X, dim):
    l, s2, meas = dim["lengthscale"], dim["variance"], dim["measure"]
    X = np.asarray(X, dtype=np.float64)
    assert X.ndim == 2 and X.shape[1] == 1
    kind = meas[0]
    if kind == "uniform":  # :49-63
        a, b = meas[1], meas[2]
        return (s2 * l / (b - a) * np.sqrt(np.pi / 2)
                * (erf((b - X) / np.sqrt(2) / l) - erf((a - X) / np.sqrt(2) / l)))
    if kind == "gaussian":  # :82-92
        mu, var = meas[1], meas[2]
        return s2 * l / np.sqrt(l ** 2 + var) * np.exp(-0.5 * ((X - mu) ** 2) / (l ** 2 + var))
    if kind == "empirical":  # :101-107
        loc, w = np.asarray(meas[1], dtype=np.float64), np.asarray(meas[2], dtype=np.float64)
        return rbf_K(X, loc, l, s2) @ w
    if kind == "mog":  # :124-136
        mu, var, w = (np.asarray(m, dtype=np.float64) for m in meas[1:4])
        tmp = np.exp(-0.5 * ((X - mu) ** 2) / (l ** 2 + var)) / np.sqrt(l ** 2 + var)
        return s2 * l * (tmp @ w.reshape(-1, 1))
    raise NotImplementedError(kind)


def var_s(dim):
    l, s2, meas = dim["lengthscale"], dim["variance"], dim["measure"]
    kind = meas[0]
    if kind == "uniform":  # :65-78
        a, b = meas[1], meas[2]
        y = (b - a) / np.sqrt(2) / l
        return (2.0 / ((b - a) ** 2) * s2 * l ** 2
                * (np.sqrt(np.pi) * y * _erf(y) + np.exp(-np.square(y)) - 1.0))
    if kind == "gaussian":  # :94-97
        return s2 * l / np.sqrt(l ** 2 + 2 * meas[2])
    if kind == "empirical":  # :109-120
        loc, w = np.asarray(meas[1], dtype=np.float64), np.asarray(meas[2], dtype=np.float64)
        return float(np.squeeze(w.T @ rbf_K(loc, None, l, s2) @ w))
    if kind == "mog":  # :138-152
        mu, var, w = (np.asarray(m, dtype=np.float64) for m in meas[1:4])
        dists = np.square(mu[:, None] - mu[None, :])
        scales = np.square(l) + var[:, None] + var[None, :]
        tmp = s2 * l / np.sqrt(scales) * np.exp(-0.5 * dists / scales)
        return float(np.squeeze(w[None, :] @ tmp @ w[:, None]))
    raise NotImplementedError(kind)


def ortho_rbf_K(X, X2, dim):
    """oak/ortho_rbf_kernel.py:157-172 (measure None = plain RBF, oak_kernel.py:199-210)."""
    if dim["measure"] is None:
        return rbf_K(X, X2, dim["lengthscale"], dim["variance"])
    cx = cov_X_s(X, dim)
    cx2 = cx if X2 is None else cov_X_s(X2, dim)
    return rbf_K(X, X2, dim["lengthscale"], dim["variance"]) - (cx @ cx2.T) / var_s(dim)


def ortho_rbf_K_diag(X, dim):
    """oak/ortho_rbf_kernel.py:174-177."""
    if dim["measure"] is None:
        return rbf_K_diag(X, dim["variance"])
    cx = cov_X_s(X, dim)
    return rbf_K_diag(X, dim["variance"]) - np.square(cx[:, 0]) / var_s(dim)


# --------------------------------------------------------------------------------------
# binary / categorical tables (oak/ortho_binary_kernel.py:29-59, ortho_categorical_kernel.py:34-74)
# --------------------------------------------------------------------------------------
def binary_table(dim):
    p0 = dim["p0"]
    p1 = 1.0 - p0
    return np.array([[np.square(p1), -p0 * p1], [-p0 * p1, np.square(p0)]]) * dim["variance"]


def binary_diag_table(dim):
    p0 = dim["p0"]
    p1 = 1.0 - p0
    return np.array([np.square(p1), np.square(p0)]) * dim["variance"]


def categorical_table(dim):
    W, kappa = np.asarray(dim["W"], dtype=np.float64), np.asarray(dim["kappa"], dtype=np.float64)
    p = np.asarray(dim["p"], dtype=np.float64).reshape(-1, 1)
    A = W @ W.T + np.diag(kappa)
    Ap = A @ p
    B = A - (Ap @ Ap.T) / (p.T @ Ap)[0]
    return B * dim["variance"]


def categorical_diag_table(dim):
    W, kappa = np.asarray(dim["W"], dtype=np.float64), np.asarray(dim["kappa"], dtype=np.float64)
    p = np.asarray(dim["p"], dtype=np.float64).reshape(-1, 1)
    A = W @ W.T + np.diag(kappa)
    Ap = A @ p
    A_diag = np.sum(np.square(W), 1) + kappa
    B_diag = A_diag - np.sum(np.square(Ap), 1) / (p.T @ Ap)[0]
    return B_diag * dim["variance"]


def _gather_K(B, X, X2):
    xi = np.asarray(X)[..., 0].astype(np.int32)  # tf.cast(float -> int32) truncates
    x2i = xi if X2 is None else np.asarray(X2)[..., 0].astype(np.int32)
    return B[:, x2i].T[:, xi].T if False else B[np.ix_(xi, x2i)]


def base_K(X, X2, dim):
    """k_d(X[:, [d]], X2[:, [d]]) for one already-sliced column."""
    t = dim["type"]
    if t == "rbf":
        return ortho_rbf_K(X, X2, dim)
    if t == "binary":
        return _gather_K(binary_table(dim), X, X2)
    if t == "categorical":
        return _gather_K(categorical_table(dim), X, X2)
    raise NotImplementedError(t)


def base_K_diag(X, dim):
    t = dim["type"]
    if t == "rbf":
        return ortho_rbf_K_diag(X, dim)
    xi = np.asarray(X)[..., 0].astype(np.int32)
    if t == "binary":
        return binary_diag_table(dim)[xi]
    if t == "categorical":
        return categorical_diag_table(dim)[xi]
    raise NotImplementedError(t)


# --------------------------------------------------------------------------------------
# OAK composite kernel (oak/oak_kernel.py:223-278)
# --------------------------------------------------------------------------------------
def compute_additive_terms(kernel_matrices, max_interaction_depth):
    """Power sums + Newton-Girard exactly as oak/oak_kernel.py:236-249."""
    s = [reduce(np.add, [np.power(k, p) for k in kernel_matrices])
         for p in range(max_interaction_depth + 1)]
    e = [np.ones_like(kernel_matrices[0])]
    for n in range(1, max_interaction_depth + 1):
        e.append((1.0 / n) * reduce(
            np.add, [((-1) ** (k - 1)) * e[n - k] * s[k] for k in range(1, n + 1)]))
    return e


def _combine(spec, additive_terms):
    v = spec["order_variances"]
    if spec.get("share_var_across_orders", True):  # :256-260
        return reduce(np.add, [sigma2 * k for sigma2, k in zip(v, additive_terms)])
    return reduce(np.add, [v[0] * additive_terms[0]] + additive_terms[1:])  # :262-265


def active_col(spec, d):
    return spec["dims"][d].get("active_dim", d)


def active_cols(spec, d):
    """Columns sub-kernel d reads: gpflow slices ``active_dims`` before calling K (oak/oak_kernel.py:253,268); a grouped
    sub-kernel (active_dims=[[0, 1], ...], :74-82) is an unconstrained RBF over several columns."""
    dim = spec["dims"][d]
    return [int(c) for c in dim["active_dims"]] if dim.get("active_dims") is not None else [active_col(spec, d)]


def oak_K(spec, X, X2=None):
    """oak/oak_kernel.py:251-265."""
    X = np.asarray(X, dtype=np.float64)
    mats = []
    for d, dim in enumerate(spec["dims"]):
        c = active_cols(spec, d)
        if len(c) > 1:
            assert dim["type"] == "rbf" and dim["measure"] is None, "only the unconstrained RBF is multi-dimensional"
            mats.append(rbf_K(X[:, c], None if X2 is None else np.asarray(X2, dtype=np.float64)[:, c], dim["lengthscale"], dim["variance"]))
            continue
        c = c[0]
        mats.append(base_K(X[:, c:c + 1], None if X2 is None else np.asarray(X2, dtype=np.float64)[:, c:c + 1], dim))
    return _combine(spec, compute_additive_terms(mats, spec["max_interaction_depth"]))


def oak_K_diag(spec, X):
    """oak/oak_kernel.py:267-278."""
    X = np.asarray(X, dtype=np.float64)
    diags = [base_K_diag(X[:, active_col(spec, d):active_col(spec, d) + 1], dim)
             for d, dim in enumerate(spec["dims"])]
    return _combine(spec, compute_additive_terms(diags, spec["max_interaction_depth"]))


def component_K(spec, subset, X, X2=None, share_var_across_orders=True):
    """KernelComponenent.K, oak/oak_kernel.py:300-320."""
    X = np.asarray(X, dtype=np.float64)
    n2 = X.shape[0] if X2 is None else np.asarray(X2).shape[0]
    if len(subset) == 0:
        return spec["order_variances"][0] * np.ones((X.shape[0], n2))
    mats = []
    for d in subset:
        c = active_cols(spec, d)
        if len(c) > 1:      # a grouped sub-kernel: the unconstrained RBF over the group's columns, as in oak_K
            dim = spec["dims"][d]
            mats.append(rbf_K(X[:, c], None if X2 is None else np.asarray(X2, dtype=np.float64)[:, c], dim["lengthscale"], dim["variance"]))
            continue
        c = c[0]
        mats.append(base_K(X[:, c:c + 1], None if X2 is None else np.asarray(X2, dtype=np.float64)[:, c:c + 1],
                           spec["dims"][d]))
    vn = spec["order_variances"][len(subset)] if share_var_across_orders else 1.0
    return vn * np.prod(mats, axis=0)


def component_K_diag(spec, subset, X, share_var_across_orders=True):
    """KernelComponenent.K_diag, oak/oak_kernel.py:322-335."""
    X = np.asarray(X, dtype=np.float64)
    if len(subset) == 0:
        return spec["order_variances"][0] * np.ones(X.shape[0])
    diags = [base_K_diag(X[:, active_col(spec, d):active_col(spec, d) + 1], spec["dims"][d]) for d in subset]
    vn = spec["order_variances"][len(subset)] if share_var_across_orders else 1.0
    return vn * np.prod(diags, axis=0)


def list_representation(num_dims, max_interaction_depth):
    """Subset enumeration order of get_list_representation, oak/oak_kernel.py:347-362."""
    out = [[]]
    if max_interaction_depth > 0:
        for ii in range(1, max_interaction_depth + 1):
            out += [list(t) for t in itertools.combinations(range(num_dims), ii)]
    return out


# --------------------------------------------------------------------------------------
# SGPR (gpflow.models.SGPR 2.2.1 recalled; op order mirrored in-tree at oak/utils.py:182-198)
# --------------------------------------------------------------------------------------
def sgpr_common(spec, X, Y, Z, noise_variance, jitter=JITTER):
    err = np.asarray(Y, dtype=np.float64)  # mean_function = None -> zero
    kuf = oak_K(spec, Z, X)                                           # utils.py:184
    kuu = oak_K(spec, Z) + jitter * np.eye(np.asarray(Z).shape[0])    # utils.py:185
    sigma = np.sqrt(noise_variance)                                   # :187
    L = np.linalg.cholesky(kuu)                                       # :188
    A = sla.solve_triangular(L, kuf, lower=True) / sigma              # :189
    AAT = A @ A.T
    B = AAT + np.eye(L.shape[0])                                      # :190-192
    LB = np.linalg.cholesky(B)                                        # :193
    Aerr = A @ err                                                    # :194
    c = sla.solve_triangular(LB, Aerr, lower=True) / sigma            # :195
    return dict(L=L, A=A, AAT=AAT, B=B, LB=LB, Aerr=Aerr, c=c, err=err, kuf=kuf, kuu=kuu)


def sgpr_elbo(spec, X, Y, Z, noise_variance, jitter=JITTER):
    """gpflow SGPR.elbo (2.2.1 recalled) -- SURVEY 8a row a8."""
    cc = sgpr_common(spec, X, Y, Z, noise_variance, jitter)
    N, P = cc["err"].shape
    Kdiag = oak_K_diag(spec, X)
    bound = -0.5 * N * P * np.log(2 * np.pi)
    bound += -P * np.sum(np.log(np.diag(cc["LB"])))
    bound -= 0.5 * N * P * np.log(noise_variance)
    bound += -0.5 * np.sum(np.square(cc["err"])) / noise_variance
    bound += 0.5 * np.sum(np.square(cc["c"]))
    bound += -0.5 * P * np.sum(Kdiag) / noise_variance
    bound += 0.5 * P * np.sum(np.diag(cc["AAT"]))
    return float(bound)


def sgpr_elbo_terms(spec, X, Y, Z, noise_variance, jitter=JITTER):
    """The kernel-dependent pieces of the bound above, one by one (P = 1), named as HipContext.sgpr_last_terms names
    them: a relative bound on the total is dominated by the data-only terms at large N, these are not."""
    cc = sgpr_common(spec, X, Y, Z, noise_variance, jitter)
    return dict(sum_log_diag_LB=float(np.sum(np.log(np.diag(cc["LB"])))), cTc=float(np.sum(np.square(cc["c"]))),
                tr_AAT=float(np.sum(np.diag(cc["AAT"]))), kappa=float(np.sum(oak_K_diag(spec, X))),
                yy=float(np.sum(np.square(cc["err"]))), n_rows=float(cc["err"].shape[0]),
                logdet_Kuu=float(2.0 * np.sum(np.log(np.diag(cc["L"])))))


def sgpr_alpha(spec, X, Y, Z, noise_variance, jitter=JITTER):
    """oak/utils.py:197-198."""
    cc = sgpr_common(spec, X, Y, Z, noise_variance, jitter)
    tmp1 = np.linalg.solve(cc["LB"].T, cc["c"])
    return np.linalg.solve(cc["L"].T, tmp1)


def sgpr_predict_f(spec, X, Y, Z, noise_variance, Xnew, jitter=JITTER):
    """gpflow SGPR.predict_f full_cov=False (2.2.1 recalled) -- SURVEY 8a row a9."""
    cc = sgpr_common(spec, X, Y, Z, noise_variance, jitter)
    Kus = oak_K(spec, Z, Xnew)
    tmp1 = sla.solve_triangular(cc["L"], Kus, lower=True)
    tmp2 = sla.solve_triangular(cc["LB"], tmp1, lower=True)
    mean = tmp2.T @ cc["c"]
    var = oak_K_diag(spec, Xnew) + np.sum(np.square(tmp2), 0) - np.sum(np.square(tmp1), 0)
    return mean, np.tile(var[:, None], [1, cc["err"].shape[1]])


def gaussian_log_density(y, mean, var):
    return -0.5 * (np.log(2 * np.pi) + np.log(var) + np.square(mean - y) / var)


def predict_log_density(mean, var, Y, noise_variance):
    """gpflow Gaussian.predict_log_density: logdensity(Y, Fmu, Fvar + sigma2) summed over outputs."""
    return np.sum(gaussian_log_density(np.asarray(Y), mean, var + noise_variance), axis=-1)


# --------------------------------------------------------------------------------------
# GPR (gpflow.models.GPR 2.2.1 recalled; in-tree mirror oak/utils.py:206-211)
# --------------------------------------------------------------------------------------
def gpr_log_marginal_likelihood(spec, X, Y, noise_variance):
    K = oak_K(spec, X)
    ks = K + noise_variance * np.eye(K.shape[0])
    L = np.linalg.cholesky(ks)
    Y = np.asarray(Y, dtype=np.float64)
    a = sla.solve_triangular(L, Y, lower=True)
    N, P = Y.shape
    return float(-0.5 * np.sum(np.square(a)) - P * np.sum(np.log(np.diag(L))) - 0.5 * N * P * np.log(2 * np.pi))


def gpr_alpha(spec, X, Y, noise_variance):
    """oak/utils.py:208-211."""
    K = oak_K(spec, X)
    L = np.linalg.cholesky(K + np.eye(K.shape[0]) * noise_variance)
    return sla.cho_solve((L, True), np.asarray(Y, dtype=np.float64))


def gpr_predict_f(spec, X, Y, noise_variance, Xnew):
    """gpflow GPR.predict_f -> base_conditional(full_cov=False, white=False)."""
    Kmm = oak_K(spec, X) + noise_variance * np.eye(np.asarray(X).shape[0])
    Kmn = oak_K(spec, X, Xnew)
    Lm = np.linalg.cholesky(Kmm)
    A = sla.solve_triangular(Lm, Kmn, lower=True)
    fvar = oak_K_diag(spec, Xnew) - np.sum(np.square(A), 0)
    A = sla.solve_triangular(Lm.T, A, lower=False)
    fmean = A.T @ np.asarray(Y, dtype=np.float64)
    return fmean, np.tile(fvar[:, None], [1, np.asarray(Y).shape[1]])


# --------------------------------------------------------------------------------------
# training objective (oak/model_utils.py:161-173; gpflow prior on constrained value, recalled)
# --------------------------------------------------------------------------------------
def gamma_log_prob(x, concentration=1.0, rate=0.2):
    from scipy.special import gammaln
    return concentration * np.log(rate) - gammaln(concentration) + (concentration - 1.0) * np.log(x) - rate * x


def log_prior(spec, use_sparsity_prior=True):
    if not (use_sparsity_prior and spec.get("share_var_across_orders", True)):
        return 0.0
    return float(np.sum([gamma_log_prob(v) for v in spec["order_variances"]]))


# --------------------------------------------------------------------------------------
# Sobol path (oak/utils.py:116-165, 221-335, 338-435)
# --------------------------------------------------------------------------------------
def f1(x, y, sigma, lengthscales, delta, mu):  # utils.py:116-124
    return (sigma ** 4 * lengthscales / np.sqrt(lengthscales ** 2 + 2 * delta ** 2)
            * np.exp(-((x - y) ** 2) / (4 * lengthscales ** 2))
            * np.exp(-((mu - (x + y) / 2) ** 2) / (2 * delta ** 2 + lengthscales ** 2)))


def f2(x, y, sigma, lengthscales, delta, mu):  # utils.py:127-144
    M = 1 / (lengthscales ** 2) + 1 / (lengthscales ** 2 + delta ** 2)
    m = 1 / M * (mu / (lengthscales ** 2 + delta ** 2) + x / lengthscales ** 2)
    C = x ** 2 / (lengthscales ** 2) + mu ** 2 / (lengthscales ** 2 + delta ** 2) - m ** 2 * M
    return (sigma ** 4 * lengthscales
            * np.sqrt((lengthscales ** 2 + 2 * delta ** 2) / (delta ** 2 * M + 1))
            * np.exp(-C / 2) / (lengthscales ** 2 + delta ** 2)
            * np.exp(-((y - mu) ** 2) / (2 * (lengthscales ** 2 + delta ** 2)))
            * np.exp(-((m - mu) ** 2) / (2 * (1 / M + delta ** 2))))


def f3(x, y, sigma, lengthscales, delta, mu):  # utils.py:147-149
    return f2(y, x, sigma, lengthscales, delta, mu)


def f4(x, y, sigma, lengthscales, delta, mu):  # utils.py:152-165
    return (sigma ** 4 * lengthscales ** 2 * (lengthscales ** 2 + 2 * delta ** 2)
            * np.sqrt((lengthscales ** 2 + delta ** 2) / (lengthscales ** 2 + 3 * delta ** 2))
            / ((lengthscales ** 2 + delta ** 2) ** 2)
            * np.exp(-((x - mu) ** 2 + (y - mu) ** 2) / (2 * (lengthscales ** 2 + delta ** 2))))


def compute_L(X, lengthscale, variance, dim, delta, mu):  # utils.py:221-240
    N = X.shape[0]
    sigma = np.sqrt(variance)
    x = np.repeat(X[:, dim], N)
    y = np.tile(X[:, dim], N)
    L = (f1(x, y, sigma, lengthscale, delta, mu) - f2(x, y, sigma, lengthscale, delta, mu)
         - f3(x, y, sigma, lengthscale, delta, mu) + f4(x, y, sigma, lengthscale, delta, mu))
    return np.reshape(L, (N, N))


def compute_L_binary_kernel(X, p0, variance, dim):  # utils.py:243-272
    assert 0 <= p0 <= 1
    N = X.shape[0]
    x = np.repeat(X[:, dim], N)
    y = np.tile(X[:, dim], N)
    p1 = 1 - p0
    L = variance * (p0 * (p1 ** 2 * (1 - x) - p0 * p1 * x) * (p1 ** 2 * (1 - y) - p0 * p1 * y)
                    + p1 * (-p0 * p1 * (1 - x) + p0 ** 2 * x) * (-p0 * p1 * (1 - y) + p0 ** 2 * y))
    return np.reshape(L, (N, N))


def compute_L_categorical_kernel(X, W, kappa, p, variance, dim):  # utils.py:275-309
    p = np.asarray(p, dtype=np.float64).reshape(-1, 1)
    assert np.abs(p.sum() - 1) < 1e-6
    B = categorical_table(dict(W=W, kappa=kappa, p=p, variance=variance))
    xi = np.asarray(X)[:, dim].astype(np.int32)
    K = B[:, xi]            # K[c, n] = B[c, x_n]   (gather/transposes of :303-305)
    return K.T @ (K * p)    # :307


def compute_L_empirical_measure(loc, w, dim_spec, z):  # utils.py:312-335
    kxu = ortho_rbf_K(np.asarray(loc, dtype=np.float64), np.asarray(z, dtype=np.float64), dim_spec)
    w = np.reshape(np.asarray(w, dtype=np.float64), [1, -1])
    return (w * kxu.T) @ kxu


def compute_sobol_oak(spec, Xc, alpha, delta=1.0, mu=0.0, share_var_across_orders=True, subsets=None, L_cache=None):
    """oak/utils.py:338-435. Xc = Z (sparse) or training X (full); alpha [rows(Xc),1].
    ``subsets``: evaluate only these terms (the reference always walks the whole list; the per-term arithmetic is unchanged).
    ``L_cache``: a dict that memoises the per-dimension matrices by (dim, v) -- the reference recomputes them per term; the
    values are identical, a checker at n = 2048 just cannot afford 4 closed-form n x n evaluations per term."""
    Xc = np.asarray(Xc, dtype=np.float64)
    D = len(spec["dims"])
    if subsets is None:
        subsets = list_representation(D, spec["max_interaction_depth"])[1:]
    N = Xc.shape[0]

    def L_of(d, v):
        key = (d, float(v))
        if L_cache is not None and key in L_cache:
            return L_cache[key]
        dim = spec["dims"][d]
        col = active_col(spec, d)
        if dim["type"] == "rbf":
            kind = None if dim["measure"] is None else dim["measure"][0]
            if kind not in ("empirical", "mog"):                    # :388-400
                Ld = compute_L(Xc, dim["lengthscale"], v, col, delta, mu)
            elif kind == "empirical":                                 # :402-412
                Ld = compute_L_empirical_measure(dim["measure"][1], dim["measure"][2], dim, Xc[:, col].reshape(-1, 1))
            else:
                raise NotImplementedError                             # :413-414
        elif dim["type"] == "binary":                                 # :416-418
            Ld = compute_L_binary_kernel(Xc, dim["p0"], v, col)
        elif dim["type"] == "categorical":                            # :420-424
            Ld = compute_L_categorical_kernel(Xc, dim["W"], dim["kappa"], dim["p"], v, col)
        else:
            raise NotImplementedError
        if L_cache is not None:
            L_cache[key] = Ld
        return Ld

    sobol = []
    for S in subsets:
        L = np.ones((N, N))
        n_order = len(S)
        for j, d in enumerate(S):
            if share_var_across_orders:
                v = spec["order_variances"][n_order] if j < 1 else 1.0   # :376-380
            else:
                v = spec["dims"][d]["variance"]
            dim = spec["dims"][d]
            if dim["type"] == "rbf" and dim["measure"] is not None and dim["measure"][0] == "empirical":
                L = v ** 2 * L * L_of(d, 1.0)                             # :402-412 (the variance multiplies outside)
            else:
                L = L * L_of(d, v)
        sobol.append(float((alpha.T @ L @ alpha)[0, 0]))                  # :429-432
    return subsets, sobol


def prediction_components(spec, Xc, alpha, X, share_var_across_orders=True):
    """get_prediction_component, oak/utils.py:491-530."""
    X = np.asarray(X, dtype=np.float64)
    D = X.shape[1]
    subsets = list_representation(D, spec["max_interaction_depth"])[1:]
    out = []
    for S in subsets:
        Kxx = np.ones((X.shape[0], alpha.shape[0]))
        for idx in S:
            Kxx = Kxx * base_K(X[:, idx:idx + 1], np.asarray(Xc, dtype=np.float64)[:, idx:idx + 1], spec["dims"][idx])
        if share_var_across_orders:
            Kxx = Kxx * spec["order_variances"][len(S)]
        out.append((Kxx @ alpha)[:, 0])
    return out


# --------------------------------------------------------------------------------------
# helpers to build specs the way OAKKernel.__init__ does (oak/oak_kernel.py:59-221)
# --------------------------------------------------------------------------------------
def make_spec(num_dims, max_interaction_depth, *, constrain_orthogonal=True, p0=None, p=None,
              lengthscales=None, order_variances=None, empirical_locations=None, empirical_weights=None,
              gmm_measures=None, share_var_across_orders=True, base_variances=None, cat_W=None, cat_kappa=None):
    p0 = [None] * num_dims if p0 is None else p0
    p = [None] * num_dims if p is None else p
    dims = []
    for d in range(num_dims):
        bv = 1.0 if base_variances is None else base_variances[d]
        if p0[d] is None and p[d] is None:
            l = 1.0 if lengthscales is None else lengthscales[d]
            if not constrain_orthogonal:
                meas = None
            elif empirical_locations is not None and empirical_locations[d] is not None:
                loc = np.asarray(empirical_locations[d], dtype=np.float64).reshape(-1, 1)
                w = (np.ones((loc.shape[0], 1)) / loc.shape[0] if empirical_weights is None or empirical_weights[d] is None
                     else np.asarray(empirical_weights[d], dtype=np.float64).reshape(-1, 1))
                meas = ("empirical", loc, w)
            elif gmm_measures is not None and gmm_measures[d] is not None:
                g = gmm_measures[d]
                meas = ("mog", np.asarray(g[0], float), np.asarray(g[1], float), np.asarray(g[2], float))
            else:
                meas = ("gaussian", 0.0, 1.0)  # oak_kernel.py:84,160
            dims.append(dict(type="rbf", lengthscale=float(l), variance=float(bv), measure=meas))
        elif p[d] is not None:
            pd = np.asarray(p[d], dtype=np.float64).reshape(-1, 1)
            W = cat_W[d] if cat_W is not None and cat_W[d] is not None else np.zeros((pd.shape[0], 2))
            kap = cat_kappa[d] if cat_kappa is not None and cat_kappa[d] is not None else np.ones(pd.shape[0])
            dims.append(dict(type="categorical", p=pd, W=np.asarray(W, float), kappa=np.asarray(kap, float), variance=float(bv)))
        else:
            dims.append(dict(type="binary", p0=float(p0[d]), variance=float(bv)))
    R = max_interaction_depth
    if order_variances is None:
        order_variances = [1.0] * (R + 1 if share_var_across_orders else 1)
    return dict(dims=dims, order_variances=[float(v) for v in order_variances], max_interaction_depth=R,
                share_var_across_orders=bool(share_var_across_orders))


def synthetic_problem(N, D, M, seed=20240601):
    """BASELINE.md section 3 / SURVEY 8(d) synthetic inputs."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    eps = rng.standard_normal(N)
    y = np.sum(np.sin(X), axis=1) + 0.5 * X[:, 0] * X[:, 1 % D] + 0.1 * eps
    y = (y - y.mean()) / y.std()
    return X, y.reshape(-1, 1), X[:M].copy()

"""Degenerate sizes through the C ABI against the oracle (dev probe; uses the oracle, hence under tests/dev)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oak import _capi
from oracle import oak_oracle as o, svgp_oracle as sv
ctx = _capi.default_context()
rng = np.random.default_rng(0)
for (N, M, D, R) in [(1, 1, 1, 1), (2, 1, 1, 0), (1, 2, 2, 2), (3, 3, 1, 1), (5, 2, 3, 3), (64, 1, 2, 1), (65, 33, 2, 2)]:
    X = rng.standard_normal((N, D)); Z = rng.standard_normal((M, D)); y = rng.standard_normal((N, 1))
    spec = o.make_spec(D, R, lengthscales=list(rng.uniform(0.8, 1.5, D)), order_variances=list(rng.uniform(0.5, 1.5, R + 1)))
    d = _capi.KernelDesc(spec)
    out = []
    for route in ("phi", "whitened"):
        ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route(route)
        e = ctx.sgpr_elbo(d, 0.1); er = o.sgpr_elbo(spec, X, y, Z, 0.1)
        e2, g = ctx.sgpr_elbo_grad(d, 0.1)
        m, v = ctx.sgpr_predict(d, X); mr, vr = o.sgpr_predict_f(spec, X, y, Z, 0.1, X)
        out.append((abs(e - er) / abs(er), abs(e2 - er) / abs(er), np.abs(m - np.asarray(mr).ravel()).max(), np.abs(v - np.asarray(vr).ravel()).max(), np.isfinite(g).all()))
    yb = (y > 0).astype(float)
    ctx.sgpr_set_data(X, yb)
    q_mu, q_sqrt = rng.standard_normal(M), rng.uniform(0.3, 1.0, M)
    es, gs, gm, gq = ctx.svgp_elbo(d, q_mu, q_sqrt, grad=True); esr = sv.svgp_elbo(spec, X, yb.ravel(), Z, q_mu, q_sqrt)
    ctx.gpr_set_data(X, y)
    lg = ctx.gpr_log_marginal(d, 0.1); lgr = o.gpr_log_marginal_likelihood(spec, X, y, 0.1)
    print((N, M, D, R), ["%.1e" % t for t in out[0][:4]], out[0][4], ["%.1e" % t for t in out[1][:4]], "svgp %.1e" % (abs(es - esr) / abs(esr)),
          np.isfinite(gs).all() and np.isfinite(gm).all(), "gpr %.1e" % (abs(lg - lgr) / abs(lgr)))

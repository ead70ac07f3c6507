"""Run-to-run repeatability of prediction at the headline size: objective, posterior mean (large and small batches), alpha, and the
whitened posterior against the phi posterior, several times over (what exposed the memory-ordering fault of r03's first fused
triangular solve: DESIGN.md section 5b).  python tools/dev_predict_flaky.py"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
from oracle import oak_oracle as o
N, D, M = 1 << 20, 16, 1024
X, y, Z = o.synthetic_problem(N, D, M)
spec = o.make_spec(D, 2)
ctx = _capi.HipContext(0)
d = _capi.KernelDesc(spec)
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
rng = np.random.default_rng(17)
Xs = rng.standard_normal((1 << 18, D))
idx = rng.choice(Xs.shape[0], 4096, replace=False)
res = []
for rep in range(4):
    e = ctx.sgpr_elbo(d, 0.01)
    t = ctx.sgpr_last_terms()
    mean, var = ctx.sgpr_predict(d, Xs)
    alpha = ctx.sgpr_alpha(M)
    Ks = ctx.gram(d, Xs[idx], Z)
    m2, v2 = ctx.sgpr_predict(d, Xs[idx])
    res.append((e, mean[idx].copy(), alpha.copy(), m2.copy(), t))
    print(rep, "elbo", repr(e), "mean-vs-K.alpha", np.abs(mean[idx] - Ks @ alpha).max(), "small-vs-big batch", np.abs(m2 - mean[idx]).max(),
          "cTc", repr(t["cTc"]), "sumlogLB", repr(t["sum_log_diag_LB"]), flush=True)
for rep in range(1, 4):
    print("rep", rep, "vs 0: elbo", res[rep][0] - res[0][0], "mean", np.abs(res[rep][1] - res[0][1]).max(), "alpha", np.abs(res[rep][2] - res[0][2]).max(),
          "m2", np.abs(res[rep][3] - res[0][3]).max())
# whitened posterior against the phi posterior, all rows of a fresh batch, several times
m_phi, v_phi = ctx.sgpr_predict(d, Xs[:65536])
for rep in range(6):
    ctx.sgpr_set_route("whitened")
    ew = ctx.sgpr_elbo(d, 0.01)
    mw, vw = ctx.sgpr_predict(d, Xs[:65536])
    ctx.sgpr_set_route("phi")
    print("whitened rep", rep, "elbo", repr(ew), "max |mean_w - mean_phi|", np.abs(mw - m_phi).max(), "var", np.abs(vw - v_phi).max(), flush=True)

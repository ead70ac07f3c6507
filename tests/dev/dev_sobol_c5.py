"""Developer script: Sobol indices at BASELINE config-5 scale (D=32 mixed, M=2048, order 4 -> 41 448 terms)."""
import sys, time; from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
import numpy as np
import bench
from oak import _capi
from oracle import oak_oracle as o
N, D, M, R = 32768, 32, 2048, 4
X, y, Z = bench.synthetic(N, D, M, mixed=True)
spec = bench.make_spec(D, R, mixed=True)
d = _capi.KernelDesc(spec)
ctx = _capi.HipContext(0); ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
e = ctx.sgpr_elbo(d, 0.01); alpha = ctx.sgpr_alpha(M)
subsets = o.list_representation(D, R)[1:]
print("terms", len(subsets))
t = time.time(); s = ctx.sobol(d, Z, alpha, subsets); dt = time.time() - t
print("sobol time %.2f s  sum %.6g  min %.3g" % (dt, s.sum(), s.min()))
# spot-check 6 terms against the oracle formula on the same alpha
idx = [0, 31, 40, 600, 6000, 41447]
ref = []
for i in idx:
    S = subsets[i]
    sp = dict(spec); 
    L = np.ones((M, M))
    for j, dd in enumerate(S):
        dim = spec["dims"][dd]; v = spec["order_variances"][len(S)] if j == 0 else 1.0
        if dim["type"] == "rbf": L = L * o.compute_L(Z, dim["lengthscale"], v, dd, 1.0, 0.0)
        elif dim["type"] == "binary": L = L * o.compute_L_binary_kernel(Z, dim["p0"], v, dd)
        else: L = L * o.compute_L_categorical_kernel(Z, dim["W"], dim["kappa"], dim["p"], v, dd)
    ref.append(float(alpha @ L @ alpha))
print("spot check rel err", np.max(np.abs(s[idx] - np.array(ref)) / np.maximum(np.abs(ref), 1e-300)))

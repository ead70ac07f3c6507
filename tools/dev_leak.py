"""Create / use / destroy contexts in a loop and watch free device memory and host RSS:  python tools/dev_leak.py [iters]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
import psutil
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N, D, M, R = 20000, 6, 256, 2
X, y, Z = bench.synthetic(N, D, M)
spec = bench.make_spec(D, R)
yb = (y > 0).astype(float)
rng = np.random.default_rng(0)
q_mu, q_sqrt = 0.3 * rng.standard_normal(M), rng.uniform(0.3, 0.9, M)
base = _capi.default_context()
proc = psutil.Process(os.getpid())
for it in range(iters):
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
    ctx.sgpr_elbo_grad(d, 0.01, 1e-6)
    ctx.sgpr_predict(d, X[:5000])
    ctx.sgpr_set_data(X, yb)
    ctx.svgp_elbo(d, q_mu, q_sqrt, grad=True)
    ctx.svgp_predict(d, q_mu, q_sqrt, X[:5000], yb[:5000].ravel())
    ctx.kmeans(X, Z.copy(), max_iter=5)
    ctx.close()
    if it % 10 == 0 or it == iters - 1:
        free, total = base.mem_info()
        print(f"iter {it}: device free {free/2**30:.3f} GiB, host rss {proc.memory_info().rss/2**20:.0f} MiB", flush=True)

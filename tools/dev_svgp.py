"""Timing of one SVGP ELBO (+ gradient) evaluation through the C ABI:  python tools/dev_svgp.py [N] [D] [M] [R]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
N, D, M, R = [int(a) for a in sys.argv[1:5]] + [200000, 8, 512, 2][len(sys.argv) - 1:]
X, y, Z = bench.synthetic(N, D, M)
y = (y > 0).astype(float)
spec = bench.make_spec(D, R)
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
d = _capi.KernelDesc(spec)
rng = np.random.default_rng(0)
q_mu, q_sqrt = 0.3 * rng.standard_normal(M), rng.uniform(0.3, 1.0, M)
for grad in (False, True):
    ctx.svgp_elbo(d, q_mu, q_sqrt, grad=grad)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = ctx.svgp_elbo(d, q_mu, q_sqrt, grad=grad); ts.append(time.perf_counter() - t0)
    print(f"N={N} D={D} M={M} R={R} grad={grad}: {min(ts)*1e3:.2f} ms  elbo={out if not grad else out[0]:.6f}")

"""Prediction throughput at the headline shape: python tools/dev_predict.py [rows]   (mean and variance of `rows` test points)"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
import bench
from oak import _capi
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
N, D, M, R = 1 << 18, 16, 1024, 2
X, y, Z = bench.synthetic(N, D, M)
spec = bench.make_spec(D, R)
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
d = _capi.KernelDesc(spec)
ctx.sgpr_elbo(d, 0.01)
Xs = np.random.default_rng(1).standard_normal((Ns, D))
ctx.sgpr_predict(d, Xs[:1024])
for _ in range(3):
    ctx.reset_timings()
    t0 = time.perf_counter()
    m, v = ctx.sgpr_predict(d, Xs)
    dt = time.perf_counter() - t0
    tm = ctx.timing("predict")
    print(f"{Ns} rows: wall {dt * 1e3:.1f} ms ({Ns / dt / 1e6:.1f} M rows/s), device phase {tm[0]:.1f} ms", flush=True)

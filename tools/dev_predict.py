"""Timing of oak_sgpr_predict at serving scale (headline model, Ns test rows)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
N, D, M, R = 1 << 20, 16, 1024, 2
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
X, y, Z = bench.synthetic(N, D, M)
spec = bench.make_spec(D, R)
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
d = _capi.KernelDesc(spec)
ctx.sgpr_elbo(d, 0.01, 1e-6)
Xs = np.random.default_rng(5).normal(size=(Ns, D))
ctx.sgpr_predict(d, Xs[:1000])
if len(sys.argv) > 2: ctx.sgpr_set_route(sys.argv[2])
for _ in range(2):
    ctx.reset_timings()
    t0 = time.perf_counter(); mean, var = ctx.sgpr_predict(d, Xs); dt = time.perf_counter() - t0
    print(f"predict {Ns} rows: wall {dt*1e3:.1f} ms ({Ns/dt/1e6:.2f} M rows/s)", {k: round(ctx.timing(k)[0], 2) for k in ("predict", "gram", "trsm", "featurize") if ctx.timing(k)[1]})
print(mean[:3].ravel(), var[:3].ravel())

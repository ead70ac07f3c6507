"""Host-side cost of one evaluation: a problem with M = 1024 but few rows, so the GPU work is the O(M^3) part only; compares
the wall time per step with the GPU-side 'total' timer.  python tools/dev_hosttime.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
ctx = _capi.default_context()
for N in (2048, 32768, 131072):
    X, y, Z = bench.synthetic(max(N, 2048), 16, 1024)
    X, y = X[:N].copy(), y[:N].copy()
    spec = bench.make_spec(16, 2)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    d = _capi.KernelDesc(spec)
    for _ in range(3): ctx.sgpr_elbo(d, 0.01)
    ctx.reset_timings()
    K = 30
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(K): ctx.sgpr_elbo(d, 0.01)
    ctx.sync(); t1 = time.perf_counter()
    ph = {k: round(ctx.timing(k)[0] / K, 3) for k in ("featurize", "gram", "syrk", "reduce", "tail", "total")}
    print(f"N={N}: wall {1e3*(t1-t0)/K:.3f} ms per step; GPU phases {ph}", flush=True)

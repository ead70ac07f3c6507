"""Timing of the device Lloyd loop at the headline inducing-point problem (N = 2^20, D = 16, K = 1024)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "orthogonal-additive-gaussian-processes_amd"))
from oak import _capi

N, D, K = 1 << 20, 16, 1024
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(20240601)
X = rng.normal(size=(N, D))
seeds = X[rng.choice(N, K, replace=False)].copy()
ctx = _capi.default_context()
ctx.kmeans(X, seeds, 1, 0.0)
for it in (1, iters):
    t0 = time.perf_counter()
    C, labels, inertia, n = ctx.kmeans(X, seeds, it, 0.0)
    dt = time.perf_counter() - t0
    print(f"max_iter={it}: n_iter={n} wall={dt*1e3:.1f} ms inertia={inertia:.6e}")

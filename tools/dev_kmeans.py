"""Inducing-point k-means at the headline shape (what oak_model.fit runs before the first evaluation): python tools/dev_kmeans.py [N] [D] [K]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
X = np.random.default_rng(0).standard_normal((N, D))
ctx = _capi.default_context()
rs = np.random.RandomState(0)
t0 = time.perf_counter(); seeds, _ = ctx.kmeans_plusplus(X, K, random_state=rs); t1 = time.perf_counter()
print(f"k-means++ seeding: {t1 - t0:.3f} s", flush=True)
for max_iter in (1, 20, 300):
    t0 = time.perf_counter(); centres, labels, inertia, n_iter = ctx.kmeans(X, seeds, max_iter, 1e-4 * float(np.mean(np.var(X, axis=0)))); dt = time.perf_counter() - t0
    print(f"Lloyd max_iter={max_iter}: {n_iter} iterations in {dt:.3f} s ({dt / max(n_iter, 1) * 1e3:.2f} ms per iteration incl. the upload of X)", flush=True)

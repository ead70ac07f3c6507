"""Time of the many-row triangular solve alone (oak_bench_trsm) on a random well-conditioned factor: python tools/dev_trsm_time.py [M] [rows] [reps]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rng = np.random.default_rng(0)
A = rng.standard_normal((M, M)) / np.sqrt(M)
L = np.linalg.cholesky(A @ A.T + np.eye(M))
B = rng.standard_normal((N, M))
ctx = _capi.default_context()
Xs, ms = ctx.bench_trsm(L, B, trans=False, reps=reps)
err = np.abs(Xs[:512] @ L.T - B[:512]).max()
lib = os.environ.get("OAK_HIP_LIB", "lib/liboak_hip.so")       # A/B of kernel variants: build each into its own .so
print(f"{lib}: M={M} rows={N}: {ms:.3f} ms = {N*M*M/ms/1e9:.1f} TFLOP/s (residual check {err:.1e})", flush=True)

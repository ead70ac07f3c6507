"""Cholesky of the O(M^3) tail alone: GPU time per factorisation and log det against NumPy (oak_bench_potrf).
python tools/dev_potrf.py [n ...]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
ctx = _capi.default_context()
for n in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]:
    ms, ld = ctx.bench_potrf(n, 20)
    i = np.arange(n)
    A = np.exp(-0.02 * np.abs(i[:, None] - i[None, :])) + 1e-3 * np.eye(n)
    ref = 2.0 * np.log(np.diag(np.linalg.cholesky(A))).sum()
    print(f"n={n:5d}  {ms * 1e3:8.1f} us per factorisation ({ms * 1e3 / ((n + 31) // 32):5.2f} us per 32-column step)  logdet rel err {abs(ld - ref) / abs(ref):.2e}", flush=True)

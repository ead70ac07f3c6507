import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
M = int(sys.argv[1]) if len(sys.argv) > 1 else 384
N = 8192
rng = np.random.default_rng(0)
A = rng.standard_normal((M, M)) / np.sqrt(M)
L = np.linalg.cholesky(A @ A.T + np.eye(M))
B = rng.standard_normal((N, M))
ctx = _capi.default_context()
Xs, ms = ctx.bench_trsm(L, B, trans=False, reps=1)
ref = np.linalg.solve(L, B.T).T
err = np.abs(Xs - ref)
np.set_printoptions(linewidth=250, precision=1)
print("max err per 16-col tile (rows = all):")
print(err.reshape(N, M // 16, 16).max(axis=(0, 2)))
print("max err per 16-row group of the first 64 rows, per 128-col block:")
print(err[:64].reshape(4, 16, M // 128, 128).max(axis=(1, 3)))
bad = np.argwhere(err > 1e-9)
print("bad entries:", len(bad), "first:", bad[:5].tolist())
r = bad[0][0] if len(bad) else 0
print("row", r, "errors by column (first 32 of first bad block):")
c0 = (bad[0][1] // 128) * 128 if len(bad) else 0
print(err[r, c0:c0 + 32])

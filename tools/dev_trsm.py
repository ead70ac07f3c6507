"""Accuracy and time of the many-row triangular solve (oak_bench_trsm) on the Cholesky factor of a real OAK Kuu:
residual max|x L^T - b| / (|x| |L|) on sampled rows in extended precision, and the forward error against a longdouble
substitution.  python tools/dev_trsm.py [M ...]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
from oracle import oak_oracle as o

ctx = _capi.default_context()
for M in [int(a) for a in sys.argv[1:]] or [384, 640, 1024]:
    D = 4 if M <= 640 else 16
    N = 16384
    X, y, Z = o.synthetic_problem(N, D, M, seed=M)
    spec = o.make_spec(D, 2, lengthscales=list(np.linspace(0.8, 1.5, D)))
    Kuu = o.oak_K(spec, Z) + 1e-6 * np.eye(M)
    L = np.linalg.cholesky(Kuu)
    B = o.oak_K(spec, X, Z)                                   # rows = K(x_n, Z): the whitened route's right-hand sides
    Xs, ms = ctx.bench_trsm(L, B, trans=False, reps=3)
    rows = np.random.default_rng(0).choice(N, 64, replace=False)
    Ll = L.astype(np.longdouble)
    # longdouble forward substitution for the sampled rows
    ref = np.zeros((len(rows), M), dtype=np.longdouble)
    Bl = B[rows].astype(np.longdouble)
    for j in range(M):
        ref[:, j] = (Bl[:, j] - ref[:, :j] @ Ll[j, :j]) / Ll[j, j]
    xs = Xs[rows].astype(np.longdouble)
    resid = np.abs(xs @ Ll.T - Bl).max(axis=1) / (np.abs(xs) @ np.abs(Ll.T)).max(axis=1)
    fwd = np.abs(xs - ref).max(axis=1) / np.abs(ref).max(axis=1)
    flops = N * M * M
    print(f"M={M} cond(Kuu)={np.linalg.cond(Kuu):.1e}: {ms:.3f} ms ({flops/ms/1e9:.1f} TFLOP/s)  residual max {float(resid.max()):.2e}  "
          f"forward error max {float(fwd.max()):.2e} median {float(np.median(fwd)):.2e}", flush=True)

"""The one-launch many-row triangular solve (csrc/trsm_fused.hip) against the blocked multi-launch form it replaces
(OAK_TRSM_UNFUSED=1): accuracy on the Cholesky factor of a real OAK Kuu (extended-precision substitution on sampled rows)
and time at a row count that fills the GPU.  python tools/dev_trsm_fused.py [M ...]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
from oracle import oak_oracle as o

ctx = _capi.default_context()
NBIG = int(os.environ.get("NBIG", 262144))
for M in [int(a) for a in sys.argv[1:]] or [384, 640, 1024]:
    D = 4 if M <= 640 else 16
    N = 16384
    X, y, Z = o.synthetic_problem(N, D, M, seed=M)
    spec = o.make_spec(D, 2, lengthscales=list(np.linspace(0.8, 1.5, D)))
    Kuu = o.oak_K(spec, Z) + 1e-6 * np.eye(M)
    L = np.linalg.cholesky(Kuu)
    B = o.oak_K(spec, X, Z)
    rows = np.random.default_rng(0).choice(N, 64, replace=False)
    Ll = L.astype(np.longdouble)
    ref = np.zeros((len(rows), M), dtype=np.longdouble)
    Bl = B[rows].astype(np.longdouble)
    for j in range(M):
        ref[:, j] = (Bl[:, j] - ref[:, :j] @ Ll[j, :j]) / Ll[j, j]
    Bbig = np.random.default_rng(1).standard_normal((NBIG, M))
    for label, env in (("fused", None), ("unfused", "1")):
        if env is None: os.environ.pop("OAK_TRSM_UNFUSED", None)
        else: os.environ["OAK_TRSM_UNFUSED"] = env
        Xs, _ = ctx.bench_trsm(L, B, trans=False, reps=1)
        xs = Xs[rows].astype(np.longdouble)
        resid = np.abs(xs @ Ll.T - Bl).max(axis=1) / (np.abs(xs) @ np.abs(Ll.T)).max(axis=1)
        fwd = np.abs(xs - ref).max(axis=1) / np.abs(ref).max(axis=1)
        full = np.abs(Xs - np.linalg.solve(L, B.T).T).max() / np.abs(Xs).max()
        _, ms = ctx.bench_trsm(L, Bbig, trans=False, reps=5)
        print(f"M={M} {label:8s} cond(Kuu)={np.linalg.cond(Kuu):.1e}: residual max {float(resid.max()):.2e}  forward error max "
              f"{float(fwd.max()):.2e}  vs numpy solve (all rows) {full:.2e} | {NBIG} rows: {ms:.3f} ms = {NBIG*M*M/ms/1e9:.1f} TFLOP/s", flush=True)

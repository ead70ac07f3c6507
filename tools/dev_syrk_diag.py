"""Phi from the quad packing of the SYRK's diagonal tiles against the diagonal-pair packing (OAK_SYRK_DIAG=pairs): which
16 x 16 tiles differ.  python tools/dev_syrk_diag.py [M ...]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
from oracle import oak_oracle as o
for M in [int(a) for a in sys.argv[1:]] or [256, 384, 1024]:
    X, y, Z = o.synthetic_problem(20000, 6, M, seed=M)
    spec = o.make_spec(6, 2)
    d = _capi.KernelDesc(spec)
    out = {}
    for mode in ("pairs", "quads"):
        os.environ["OAK_SYRK_DIAG"] = mode
        ctx = _capi.HipContext(0)
        ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
        ctx.sgpr_local_stats(d)
        out[mode] = ctx.sgpr_get_stats()[:M * M].reshape(M, M)
        ctx.close()
    diff = np.abs(out["quads"] - out["pairs"])
    T = M // 16
    bad = np.argwhere(diff.reshape(T, 16, T, 16).max(axis=(1, 3)) > 1e-9 * np.abs(out["pairs"]).max())
    print(f"M={M}: max diff {diff.max():.3e}; differing 16x16 tiles (row, col): {bad[:40].tolist()} ({len(bad)} in all)")

"""Phi-route deviation from the whitened (GPflow-order) route against cond(Kuu), for both accumulations of Phi: does the exact int8 accumulation move
the conditioning threshold of the auto route?  N = 2^18, D = 16, M = 1024; the conditioning is swept through the lengthscale.  python tools/dev_cond.py"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
N, D, M, R = 1 << 18, 16, 1024, 2
X, y, Z = bench.synthetic(N, D, M)
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
TERMS = ("sum_log_diag_LB", "cTc", "tr_AAT")
for ls in (1.0, 1.5, 2.0, 2.5, 3.0, 4.0):
    spec = bench.make_spec(D, R)
    for dim in spec["dims"]:
        dim["lengthscale"] = ls
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_route("whitened"); ctx.sgpr_set_precision("fp64")
    ew = ctx.sgpr_elbo(d, 0.01); tw = ctx.sgpr_last_terms()
    out = [f"ls={ls}: diag-ratio^2 {tw['cond_estimate']:.3g}"]
    ctx.sgpr_set_route("phi")
    for mode in ("fp64", "int8crt"):
        ctx.sgpr_set_precision(mode)
        e = ctx.sgpr_elbo(d, 0.01); t = ctx.sgpr_last_terms()
        out.append(f"{mode}: ELBO {abs(e - ew) / abs(ew):.1e} " + " ".join(f"{k} {abs(t[k] - tw[k]) / abs(tw[k]):.1e}" for k in TERMS))
    print("  |  ".join(out), flush=True)

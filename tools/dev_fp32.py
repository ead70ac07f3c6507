"""fp32 statistics mode against fp64 on a few problems: conditioning estimate, ELBO and term deviations, step times.
python tools/dev_fp32.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
ctx = _capi.default_context()
for name, rows in (("c2", None), ("headline", 131072), ("headline", None), ("c3", None), ("c5", None)):
    cfg = bench.CONFIGS[name]
    N, D, M, R = rows or cfg["N"], cfg["D"], cfg["M"], cfg["R"]
    X, y, Z = bench.synthetic(cfg["N"], D, M, mixed=cfg.get("mixed", False))
    X, y = np.ascontiguousarray(X[:N]), np.ascontiguousarray(y[:N])
    spec = bench.make_spec(D, R, mixed=cfg.get("mixed", False))
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    out = {}
    for mode in ("fp64", "fp32"):
        ctx.sgpr_set_precision(mode)
        for _ in range(2):
            e = ctx.sgpr_elbo(d, 0.01)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(5):
            e = ctx.sgpr_elbo(d, 0.01)
        ctx.sync(); dt = (time.perf_counter() - t0) / 5
        out[mode] = (e, ctx.sgpr_last_terms(), dt, ctx.sgpr_stats_precision())
    ctx.sgpr_set_precision("fp64")
    e64, t64, dt64, _ = out["fp64"]; e32, t32, dt32, used = out["fp32"]
    terms = {k: abs(t32[k] - t64[k]) / max(abs(t64[k]), 1e-300) for k in ("sum_log_diag_LB", "cTc", "tr_AAT")}
    print(f"{name} N={N} M={M} D={D} R={R}: cond_est={t32['cond_estimate']:.3g} used={used} fp64 {dt64*1e3:.2f} ms  fp32 {dt32*1e3:.2f} ms "
          f"({dt64/dt32:.2f}x)  ELBO rel diff {abs(e32-e64)/abs(e64):.2e}  terms {({k: float(f'{v:.1e}') for k, v in terms.items()})}", flush=True)

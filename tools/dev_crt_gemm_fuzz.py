"""Random shapes through the int8 adjoint GEMM of the gradient (csrc/crt_gemm.hip, forced on with OAK_CRT_GEMM=1) against the fp64 GEMM on the
same statistics: gradient w.r.t. the hyperparameters and the inducing inputs, relative to their largest entries, next to the conditioning estimate.
python tools/dev_crt_gemm_fuzz.py [cases] [seed]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oak import _capi
import cases
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _capi.default_context()
KINDS = [("gaussian",), ("gaussian", "binary"), ("gaussian", "categorical", "binary", "uniform"), ("gaussian", "mog"), ("gauss2", "gaussian")]
for it in range(ncase):
    N = int(rng.integers(4096, 120000)); M = int(rng.choice([96, 200, 256, 300, 512, 640, 768, 1000, 1024, 1500, 2048, 3000, 4096])); D = int(rng.integers(2, 24)); R = int(rng.integers(1, min(D, 4) + 1))
    kinds = KINDS[int(rng.integers(len(KINDS)))]
    spec = cases.random_spec(rng, D, R, kinds, share=bool(rng.integers(2)))
    for dim in spec["dims"]:
        if dim["type"] == "rbf": dim["lengthscale"] = float(rng.uniform(0.3, 1.2))
    X = cases.random_inputs(rng, spec, N)
    Z = X[rng.choice(N, M, replace=False)].copy()
    y = (np.sin(X[:, 0]) + 0.1 * rng.standard_normal(N)).reshape(-1, 1)
    nx = int(rng.integers(0, 3))
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi"); ctx.sgpr_set_precision("int8crt")
    ctx.sgpr_set_extra_targets(np.column_stack([np.cos(X[:, 0] * (p + 2)) for p in range(nx)]) if nx else None)
    os.environ["OAK_CRT_GEMM"] = "0"; e0, g0, z0 = ctx.sgpr_elbo_grad_z(d, 0.05, M, D)
    os.environ["OAK_CRT_GEMM"] = "1"; e1, g1, z1 = ctx.sgpr_elbo_grad_z(d, 0.05, M, D)
    info = ctx.bench_crt_info(); est = ctx.sgpr_last_terms()["cond_estimate"]
    dg = float(np.abs(g1 - g0).max() / np.abs(g0).max()); dz = float(np.abs(z1 - z0).max() / (np.abs(z0).max() + 1e-300))
    ok = info["gemm_planes"] > 0 and e1 == e0 and dg <= 1e-12 * max(est, 100.0) and dz <= 1e-10 * max(est, 100.0) and np.isfinite(g1).all()
    print(f"{it:2d} N={N} M={M} D={D} R={R} nx={nx} kinds={kinds}: used={ctx.sgpr_stats_precision()} fused={info['fused']} gemm planes={info['gemm_planes']} bits={info['gemm_bits']} "
          f"estimate {est:.3g}  grad {dg:.1e}  gradZ {dz:.1e}{'' if ok else '   <<<<<< CHECK'}", flush=True)
os.environ.pop("OAK_CRT_GEMM", None); ctx.sgpr_set_extra_targets(None); ctx.sgpr_set_precision("auto")

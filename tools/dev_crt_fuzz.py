"""Random shapes through both accumulations of Phi (int8 CRT forced on): max deviation of Phi relative to sqrt(Phi_aa Phi_bb), psi, ELBO.
python tools/dev_crt_fuzz.py [cases] [seed]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oak import _capi
import cases
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _capi.default_context()
KINDS = [("gaussian",), ("gaussian", "binary"), ("gaussian", "categorical", "binary", "uniform"), ("gaussian", "mog"), ("gauss2", "gaussian"), ("none", "gaussian")]
worst = 0.0
for it in range(ncase):
    N = int(rng.integers(4096, 90000)); M = int(rng.integers(17, 700)); D = int(rng.integers(1, 40)); R = int(rng.integers(0, min(D, 6) + 1))
    kinds = KINDS[int(rng.integers(len(KINDS)))]
    share = bool(rng.integers(2))
    spec = cases.random_spec(rng, D, R, kinds, share=share)
    X = cases.random_inputs(rng, spec, N)
    Z = X[rng.choice(N, M, replace=False)].copy()
    y = (np.sin(X[:, 0]) + 0.1 * rng.standard_normal(N)).reshape(-1, 1)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    if rng.integers(3) == 0: ctx.sgpr_set_panel_rows(int(rng.integers(3000, 20000)))
    else: ctx.sgpr_set_panel_rows(0)
    ctx.sgpr_set_precision("fp64"); ctx.sgpr_local_stats(d); s64 = ctx.sgpr_get_stats()
    ctx.sgpr_set_precision("int8crt"); ctx.sgpr_local_stats(d); sc = ctx.sgpr_get_stats(); used = ctx.sgpr_stats_precision(); info = ctx.bench_crt_info()
    P64, Pc = s64[:M * M].reshape(M, M), sc[:M * M].reshape(M, M)
    dg = np.sqrt(np.abs(np.outer(np.diag(P64), np.diag(P64)))) + 1e-300
    dev = float((np.abs(Pc - P64) / dg).max()); dpsi = float(np.abs(sc[M * M:M * M + M] - s64[M * M:M * M + M]).max() / (np.abs(s64[M * M:M * M + M]).max() + 1e-300))
    worst = max(worst, dev)
    flag = "" if (dev < 5e-13 and dpsi < 1e-12 and np.array_equal(Pc, Pc.T) and used == "int8crt") else "   <<<<<< CHECK"
    print(f"{it:2d} N={N} M={M} D={D} R={R} share={share} kinds={kinds} used={used} fused={info['fused']} planes={info['planes']} bits={info['bits']}: Phi dev {dev:.2e} psi {dpsi:.1e}{flag}", flush=True)
ctx.sgpr_set_panel_rows(0); ctx.sgpr_set_precision("auto")
print(f"worst Phi deviation {worst:.2e}")

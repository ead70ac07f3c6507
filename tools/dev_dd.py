"""Ill-conditioned Kuu (near-duplicate inducing points, long lengthscales): the automatic route -- exact int8 Phi, double-double whitening in the tail
(csrc/ddgemm.hip) -- against the whitened route and the C oracle, term by term, with times.  python tools/dev_dd.py [N] [M]"""
import sys, time, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
from oracle import c_oracle, oak_oracle as o
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
M = int(sys.argv[2]) if len(sys.argv) > 2 else 768
D, R = 8, 2
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D))
y = (np.sin(X[:, 0]) + 0.5 * X[:, 1] * X[:, 2] + 0.1 * rng.standard_normal(N)).reshape(-1, 1); y = (y - y.mean()) / y.std()
Z = X[:M].copy(); Z[M // 2:] = Z[:M - M // 2] + 0.02 * rng.standard_normal((M - M // 2, D))      # near-duplicates
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
TERMS = ("sum_log_diag_LB", "cTc", "tr_AAT", "logdet_Kuu")
for ls in (1.0, 2.0, 3.0):
    spec = o.make_spec(D, R, lengthscales=[ls] * D)
    d = _capi.KernelDesc(spec)
    ref, parts = c_oracle.sgpr_elbo_chunked(spec, X, y, Z, 0.01, chunk=8192, return_parts=True)
    rows = []
    for label, route, prec, env in (("auto (int8 Phi + dd tail)", "auto", "auto", {}), ("phi, int8 Phi, fp64 tail", "phi", "int8crt", {"OAK_TAIL_DD": "0"}),
                                    ("phi, fp64 kernels", "phi", "fp64", {}), ("whitened", "whitened", "fp64", {})):
        for k, v in env.items(): os.environ[k] = v
        ctx.sgpr_set_route(route); ctx.sgpr_set_precision(prec)
        for _ in range(2): e = ctx.sgpr_elbo(d, 0.01)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(5): e = ctx.sgpr_elbo(d, 0.01)
        ctx.sync(); dt = (time.perf_counter() - t0) / 5
        t = ctx.sgpr_last_terms(); info = ctx.bench_crt_info()
        for k in env: del os.environ[k]
        rows.append(f"   {label:28s} {dt*1e3:7.2f} ms  whitened={ctx.sgpr_stats_whitened()} prec={ctx.sgpr_stats_precision()} dd={info['tail_dd']}  ELBO {abs(e-ref)/abs(ref):.1e}  " +
                    " ".join(f"{k} {abs(t[k]-parts['terms'][k])/max(abs(parts['terms'][k]),1.0 if k=='logdet_Kuu' else 1e-300):.1e}" for k in TERMS))
    print(f"N={N} M={M} ls={ls}: diag-ratio^2 {t['cond_estimate']:.3g}")
    print("\n".join(rows), flush=True)
ctx.sgpr_set_route("auto"); ctx.sgpr_set_precision("auto")

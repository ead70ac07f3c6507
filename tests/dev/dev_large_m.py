"""Sanity at a large inducing set (M = 4096): ELBO / gradient step against the multicore C oracle."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
from oracle import c_oracle, oak_oracle as o
N, D, M, R = 65536, 12, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 2
X, y, Z = o.synthetic_problem(N, D, M, seed=3)
spec = o.make_spec(D, R)
d = _capi.KernelDesc(spec)
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
for route in ("phi", "whitened"):
    ctx.sgpr_set_route(route)
    ctx.sgpr_elbo(d, 0.01)
    ctx.reset_timings()
    t0 = time.perf_counter(); e = ctx.sgpr_elbo(d, 0.01); t1 = time.perf_counter()
    eg, g = ctx.sgpr_elbo_grad(d, 0.01); t2 = time.perf_counter()
    print(route, "elbo", e, f"fwd {1e3*(t1-t0):.1f} ms, fwd+grad {1e3*(t2-t1):.1f} ms", {k: round(ctx.timing(k)[0]/max(ctx.timing(k)[1],1),2) for k in ("gram","syrk","trsm","tail","bwd_tail")})
t0 = time.perf_counter(); er = c_oracle.sgpr_elbo_chunked(spec, X, y, Z, 0.01, 1e-6, chunk=16384); print("oracle", er, f"{time.perf_counter()-t0:.1f} s", "rel", abs(e-er)/abs(er))

"""Wall time per evaluation at the sizes the reference's own examples run (UCI regression: N ~ 1e3 .. 5e4, M = 200 inducing points):
python tools/dev_small.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
import bench
from oak import _capi
ctx = _capi.default_context()
for (N, D, M, R) in ((1000, 8, 200, 2), (5000, 8, 200, 2), (20000, 13, 200, 3), (50000, 8, 200, 8), (9000, 32, 200, 4)):
    X, y, Z = bench.synthetic(N, D, M)
    spec = bench.make_spec(D, R)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
    out = []
    for route in ("phi", "whitened"):
        ctx.sgpr_set_route(route)
        for name, fn in (("elbo", lambda: ctx.sgpr_elbo(_capi.KernelDesc(spec), 0.01)), ("elbo+grad", lambda: ctx.sgpr_elbo_grad(_capi.KernelDesc(spec), 0.01))):
            fn(); fn()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            out.append(f"{route} {name} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(200):
        _capi.KernelDesc(spec)
    tk = (time.perf_counter() - t0) / 200 * 1e3
    print(f"N={N} D={D} M={M} R={R}: " + ", ".join(out) + f"; KernelDesc() {tk:.3f} ms", flush=True)
# the full GP (the reference takes it for N <= 1000, model_utils.py:374) at its regression example's setting, depth = number of inputs
for (N, D) in ((455, 13), (1000, 8), (1000, 13), (300, 6)):
    X, y, _ = bench.synthetic(N, D, 8)
    spec = bench.make_spec(D, D)
    ctx.gpr_set_data(X, y)
    fn = lambda: ctx.gpr_log_marginal_grad(_capi.KernelDesc(spec), 0.01)
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print(f"GPR N={N} D={D} depth={D}: log marginal + gradient {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)

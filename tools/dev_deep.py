"""Forward and forward+gradient at depths beyond the fast backward kernels (depth > 4 or > 32 sub-kernels): python tools/dev_deep.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
import bench
from oak import _capi
ctx = _capi.default_context()
N, M = 1 << 18, 1024
for (D, R) in ((16, 2), (16, 4), (13, 5), (13, 13), (16, 8), (32, 4), (32, 8), (32, 32), (40, 3), (24, 12), (32, 16), (64, 2), (48, 6)):
    X, y, Z = bench.synthetic(N, D, M)
    spec = bench.make_spec(D, R)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    d = _capi.KernelDesc(spec)
    def timed(fn, k=3):
        fn(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(k):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / k * 1e3
    f = timed(lambda: ctx.sgpr_elbo(d, 0.01))
    ctx.reset_timings()
    g = timed(lambda: ctx.sgpr_elbo_grad(d, 0.01))
    t = {k: ctx.timing(k)[0] / max(ctx.timing(k)[1], 1) for k in ("gram", "bwd_gram", "bwd_gemm")}
    print(f"D={D} R={R}: forward {f:.2f} ms, forward+gradient {g:.2f} ms  (gram {t['gram']:.2f}, bwd pair kernel {t['bwd_gram']:.2f}, bwd gemm {t['bwd_gemm']:.2f})", flush=True)

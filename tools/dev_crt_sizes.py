"""Where does the int8 route pay?  Forward step on both routes over (N, M) at D = 16, order 2 (the data behind the automatic rule N M^2 >= 2^36, M >= 512).
python tools/dev_crt_sizes.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
ctx = _capi.default_context()
D, R = 16, 2
Xf, yf, _ = bench.synthetic(1 << 20, D, 2048)
spec = bench.make_spec(D, R)
d = _capi.KernelDesc(spec)
for M in (512, 768, 1024, 1536, 2048):
    Z = np.ascontiguousarray(Xf[:M])
    for N in (1 << 16, 1 << 17, 1 << 18, 1 << 20):
        ctx.sgpr_set_data(np.ascontiguousarray(Xf[:N]), np.ascontiguousarray(yf[:N])); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
        out = {}
        for mode in ("fp64", "int8crt"):
            ctx.sgpr_set_precision(mode)
            for _ in range(2): ctx.sgpr_elbo(d, 0.01)
            ctx.sync(); t0 = time.perf_counter()
            K = 6
            for _ in range(K): ctx.sgpr_elbo(d, 0.01)
            ctx.sync(); out[mode] = (time.perf_counter() - t0) / K * 1e3
        print(f"M={M} N={N} log2(N M^2)={np.log2(float(N) * M * M):.1f}: fp64 {out['fp64']:.2f} ms  int8crt {out['int8crt']:.2f} ms  ratio {out['fp64'] / out['int8crt']:.2f}", flush=True)

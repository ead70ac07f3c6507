"""Per-rank forward step time for the row shards of the headline problem (N/g rows, g = 1, 2, 4, 8) on ONE GPU:
an upper bound on strong scaling (the all-reduce is not included).  --grad: also the forward+gradient step."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd"))
sys.path.insert(0, str(ROOT))
from oak import _capi
import bench

N, D, M, R = 1 << 20, 16, 1024, 2
X, y, Z = bench.synthetic(N, D, M)
spec = bench.make_spec(D, R)
ctx = _capi.default_context()
PH = ["featurize", "gram", "syrk", "crt_syrk", "crt_reduce", "reduce", "tail", "total"]
GS = tuple(int(a) for a in sys.argv[1:] if a.isdigit()) or (1, 2, 4, 8)      # e.g. `dev_shard.py 8` under rocprofv3 for one shard size
for g in GS:
    n = N // g
    ctx.sgpr_set_data(np.ascontiguousarray(X[:n]), np.ascontiguousarray(y[:n]))
    ctx.sgpr_set_inducing(Z)
    ctx.sgpr_set_route("phi")
    for _ in range(2):
        ctx.sgpr_elbo(_capi.KernelDesc(spec), 0.01, 1e-6)
    ctx.reset_timings()
    ctx.sync(); t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        ctx.sgpr_elbo(_capi.KernelDesc(spec), 0.01, 1e-6)
    ctx.sync(); dt = (time.perf_counter() - t0) / K
    ph = {k: round(ctx.timing(k)[0] / K, 3) for k in PH}
    if g == GS[0]:
        base = dt * g                      # perfect strong scaling measured from the first (largest) shard of this run
    print(f"g={g} rows={n} wall={dt*1e3:.2f} ms  linear={base*1e3/g:.2f}  precision={ctx.sgpr_stats_precision()}  {ph}")
    if "--grad" in sys.argv:
        for _ in range(2):
            ctx.sgpr_elbo_grad(_capi.KernelDesc(spec), 0.01, 1e-6)
        ctx.reset_timings()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(K):
            ctx.sgpr_elbo_grad(_capi.KernelDesc(spec), 0.01, 1e-6)
        ctx.sync(); dtg = (time.perf_counter() - t0) / K
        phg = {k: round(ctx.timing(k)[0] / K, 3) for k in PH + ["bwd_gemm", "bwd_gram", "bwd_tail"]}
        print(f"      forward+gradient wall={dtg*1e3:.2f} ms  adjoint GEMM planes={ctx.bench_crt_info()['gemm_planes']}  {phg}")

"""Beyond the panel chunk (N = 4 x 2^20 rows > 16 GiB of Kfu): the chunked forward / backward against the loopback
communicator on one copy of the rows (same arithmetic, unchunked).  python tools/dev_bigN.py [reps]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, D, M, R = 1 << 20, 16, 1024, 2
X, y, Z = bench.synthetic(N, D, M)
spec = bench.make_spec(D, R)
d = _capi.KernelDesc(spec)
ref = _capi.HipContext(0)
ref.sgpr_set_data(X, y); ref.sgpr_set_inducing(Z); ref.sgpr_set_route("phi"); ref.comm_init_loopback(reps)
e_ref, g_ref = ref.sgpr_elbo_grad(d, 0.01, 1e-6)
ref.close()
ctx = _capi.HipContext(0)
ctx.sgpr_set_data(np.tile(X, (reps, 1)), np.tile(y, (reps, 1))); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
for it in range(2):
    ctx.reset_timings()
    t0 = time.perf_counter(); e = ctx.sgpr_elbo(d, 0.01, 1e-6); t1 = time.perf_counter()
    e2, g = ctx.sgpr_elbo_grad(d, 0.01, 1e-6); t2 = time.perf_counter()
print(f"N={reps * N}: forward {1e3 * (t1 - t0):.1f} ms, forward+gradient {1e3 * (t2 - t1):.1f} ms")
print(f"elbo rel diff vs loopback {abs(e - e_ref) / abs(e_ref):.2e} / {abs(e2 - e_ref) / abs(e_ref):.2e}; gradient max rel diff {np.abs(g - g_ref).max() / np.abs(g_ref).max():.2e}")
free, total = ctx.mem_info(); print(f"device memory in use {(total - free) / 2**30:.1f} GiB")

"""Forward + gradient at the headline shape (or `N D M R` from the command line): wall per evaluation and the GPU phase times,
the backward pair kernel's among them (bwd_gram).  OAK_BWD_ROWS=0 selects the columns-in-lanes kernel, OAK_BWD_ROWS_COLS the
rows-in-lanes kernel's columns per workgroup."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench

N, D, M, R = [int(a) for a in (sys.argv[1:5] + [1 << 20, 16, 1024, 2][len(sys.argv) - 1:])][:4]
X, y, Z = bench.synthetic(N, D, M)
desc = _capi.KernelDesc(bench.make_spec(D, R))
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
for _ in range(2):
    e, g = ctx.sgpr_elbo_grad(desc, 0.01)
ctx.reset_timings(); ctx.sync(); t0 = time.perf_counter()
K = 5
for _ in range(K):
    e, g = ctx.sgpr_elbo_grad(desc, 0.01)
ctx.sync(); dt = (time.perf_counter() - t0) / K
PH = ["gram", "syrk", "tail", "bwd_gemm", "bwd_gram", "bwd_tail", "bwd_small", "total"]
print(f"N={N} D={D} M={M} R={R}: {dt * 1e3:.2f} ms per forward+gradient", {k: round(ctx.timing(k)[0] / K, 3) for k in PH})
print("elbo %.12e  grad[:4] %s  |grad| %.12e" % (e, np.array2string(g[:4], precision=10), float(np.linalg.norm(g))))

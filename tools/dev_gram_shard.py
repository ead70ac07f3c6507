"""Gram-stage time of one rank's row shard (N/8 rows of the headline problem) for a tuning-knob sweep."""
import sys, time, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
N, D, M, R = (1 << 20) // int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17, 16, 1024, 2
X, y, Z = bench.synthetic(N, D, M); spec = bench.make_spec(D, R)
ctx = _capi.default_context(); ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
d = _capi.KernelDesc(spec)
for _ in range(3): ctx.sgpr_elbo(d, 0.01)
ctx.reset_timings(); K = 20
for _ in range(K): ctx.sgpr_elbo(d, 0.01)
print(os.environ.get("OAK_GRAM_WG_PER_CU", "default"), {k: round(ctx.timing(k)[0] / K, 3) for k in ("gram", "syrk", "tail", "total")})

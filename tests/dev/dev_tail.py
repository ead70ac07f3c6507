"""Developer script: repeat the O(M^3) SGPR tail at M=1024 for kernel-level profiling."""
import sys; from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
import numpy as np
from oak import _capi
from oracle import oak_oracle as o
X, y, Z = o.synthetic_problem(65536, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
spec = o.make_spec(16, 2); d = _capi.KernelDesc(spec)
ctx = _capi.HipContext(0); ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
ctx.sgpr_local_stats(d)
for _ in range(10): ctx.sgpr_tail(d, 0.01)

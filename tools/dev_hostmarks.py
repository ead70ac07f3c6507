"""Host-side timeline of one forward evaluation: when the host STARTED enqueueing each phase (the library's breadcrumbs,
oak_debug_state, microsecond ages), next to the wall time per step.  python tools/dev_hostmarks.py [c2|shard8|headline]"""
import re, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = dict(c2=(65536, 8, 512), shard8=(131072, 16, 1024), headline=(1 << 20, 16, 1024))[which]
N, D, M = cfg
X, y, Z = bench.synthetic(N, D, M)
desc = _capi.KernelDesc(bench.make_spec(D, 2))
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
for _ in range(5):
    ctx.sgpr_elbo(desc, 0.01)
ctx.sync(); t0 = time.perf_counter()
K = 50
for _ in range(K):
    ctx.sgpr_elbo(desc, 0.01)
ctx.sync(); print(f"{which}: wall {(time.perf_counter() - t0) / K * 1e3:.3f} ms per step")
ages = [(m.group(1), float(m.group(2))) for m in re.finditer(r"\] (\S+)\s+([0-9.]+) s ago", _capi.debug_state())]
# the last evaluation: from the last 'elbo_enter'
k = max(i for i, (n, _) in enumerate(ages) if n == "elbo_enter")
t_enter = ages[k][1]
for n, a in ages[k:]:
    print(f"  {n:12s} +{(t_enter - a) * 1e6:8.1f} us")

"""A few forward ELBO steps of a bench configuration (for kernel traces of the tail):  python tools/dev_fwd_only.py [config] [steps] [rows]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "headline"]
N, D, M, R = cfg["N"], cfg["D"], cfg["M"], cfg["R"]
X, y, Z = bench.synthetic(N, D, M, mixed=cfg.get("mixed", False))
spec = bench.make_spec(D, R, mixed=cfg.get("mixed", False))
ctx = _capi.default_context()
if len(sys.argv) > 3:
    X, y = X[:int(sys.argv[3])].copy(), y[:int(sys.argv[3])].copy()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    print(ctx.sgpr_elbo(_capi.KernelDesc(spec), 0.01, 1e-6))

"""Time of the N x M Gram kernel alone on resident data (oak_bench_gram_resident), headline shape: python tools/dev_gram_time.py [reps]
OAK_HIP_LIB selects the library (A/B of kernel variants built into their own .so)."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
cfg = bench.CONFIGS[os.environ.get("CONFIG", "headline")]
N, D, M, R = cfg["N"], int(os.environ.get("D", cfg["D"])), cfg["M"], int(os.environ.get("R", cfg["R"]))     # D= / R= override the config
mixed = cfg.get("mixed", False) and D == cfg["D"]
X, y, Z = bench.synthetic(N, D, M, mixed=mixed)
ctx = _capi.HipContext(0)
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
desc = _capi.KernelDesc(bench.make_spec(D, R, mixed=mixed))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx.bench_gram_resident(desc)
ts = []
for _ in range(reps):
    ctx.reset_timings(); ctx.bench_gram_resident(desc)
    t = ctx.timing("gram"); ts.append(t[0] / max(t[1], 1))
print(f"{os.environ.get('OAK_HIP_LIB', 'lib/liboak_hip.so')}: gram {N} x {M}, D={D}: " + " ".join(f"{t:.3f}" for t in ts) + " ms")

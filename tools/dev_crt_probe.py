"""Phase times of the int8 CRT statistics pass alone (oak_sgpr_local_stats: no tail, so the timing probes of the int8 SYRK -- which
produce wrong numbers on purpose -- can be timed).  OAK_CRT_SYRK selects the kernel variant.  python tools/dev_crt_probe.py [config]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
cfg = bench.CONFIGS[name]
N, D, M, R = cfg["N"], cfg["D"], cfg["M"], cfg["R"]
X, y, Z = bench.synthetic(N, D, M, mixed=cfg.get("mixed", False))
d = _capi.KernelDesc(bench.make_spec(D, R, mixed=cfg.get("mixed", False)))
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi"); ctx.sgpr_set_precision("int8crt")
for _ in range(2): ctx.sgpr_local_stats(d)
ctx.sync(); ctx.reset_timings(); t0 = time.perf_counter()
REPS = int(__import__("os").environ.get("OAK_PROBE_REPS", "5"))
for _ in range(REPS): ctx.sgpr_local_stats(d)
ctx.sync(); dt = (time.perf_counter() - t0) / REPS
ph = {}
for p in ("featurize", "gram", "crt_convert", "crt_syrk", "crt_reduce", "reduce"):
    ms, cnt = ctx.timing(p)
    if cnt: ph[p] = round(ms / cnt, 3)
print(f"{name}: stats pass {dt * 1e3:.2f} ms  {ph}", flush=True)

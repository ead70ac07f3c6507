"""Runs one piece of the path back to back for a few seconds (for tools/power_probe.sh): python tools/dev_loop.py gram|stats_fp64|stats_crt|grad [seconds]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
what = sys.argv[1]; secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
cfg = bench.CONFIGS["headline"]
N, D, M, R = cfg["N"], cfg["D"], cfg["M"], cfg["R"]
X, y, Z = bench.synthetic(N, D, M)
d = _capi.KernelDesc(bench.make_spec(D, R))
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
ctx.sgpr_set_precision("fp64" if what in ("stats_fp64", "gram", "grad") else "int8crt")
fn = {"gram": lambda: ctx.bench_gram_resident(d), "stats_fp64": lambda: ctx.sgpr_local_stats(d), "stats_crt": lambda: ctx.sgpr_local_stats(d),
      "grad": lambda: ctx.sgpr_elbo_grad(d, 0.01)}[what]
fn(); ctx.sync()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    fn(); n += 1
ctx.sync()
print(f"{what}: {n} calls, {(time.perf_counter() - t0) / n * 1e3:.2f} ms each")

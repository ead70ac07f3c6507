"""Sobol pass at the depths the reference's regression example uses (depth = number of inputs): python tools/dev_sobol_deep.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
import numpy as np, itertools
import bench
from oak import _capi
ctx = _capi.default_context()
for (D, R, M) in ((13, 13, 200), (11, 11, 200), (8, 8, 200), (13, 4, 200), (16, 16, 128)):
    spec = bench.make_spec(D, R)
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((M, D)); alpha = rng.standard_normal(M)
    subsets = [list(t) for r in range(1, R + 1) for t in itertools.combinations(range(D), r)]
    packed = ctx.pack_subsets(subsets)
    d = _capi.KernelDesc(spec)
    ctx.sobol(d, Z, alpha, packed)
    t0 = time.perf_counter(); out = ctx.sobol(d, Z, alpha, packed); dt = time.perf_counter() - t0
    print(f"D={D} depth={R} M={M}: {len(subsets)} terms in {dt*1e3:.1f} ms, info {ctx.sobol_last_info()}", flush=True)

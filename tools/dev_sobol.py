"""Time of the Sobol pass (oak_sobol) at BASELINE config 5's shape: python tools/dev_sobol.py [M] [D] [depth] [reps] [paths]
paths: comma list of gram,terms (default both).  Prints wall time per call, the device phases and the path taken."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
import bench
from oak import _capi
from oracle import oak_oracle as o
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
D = int(sys.argv[2]) if len(sys.argv) > 2 else 32
R = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
paths = (sys.argv[5] if len(sys.argv) > 5 else "gram,terms").split(",")
X, y, Z = bench.synthetic(max(M, 4096), D, M, mixed=(D == 32))
spec = bench.make_spec(D, R, mixed=(D == 32))
d = _capi.KernelDesc(spec)
rng = np.random.default_rng(3)
alpha = rng.standard_normal(M) * rng.uniform(0.1, 2.0, M)
subsets = o.list_representation(D, R)[1:]
ctx = _capi.default_context()
res = {}
for path in paths:
    ctx.sobol_set_path(path)
    res[path] = ctx.sobol(d, Z, alpha, subsets)          # warm-up (buffers, descriptor tables)
    ctx.reset_timings()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.sobol(d, Z, alpha, subsets)
    wall = (time.perf_counter() - t0) / reps * 1e3
    info = ctx.sobol_last_info()
    ph = {k: ctx.timing(k) for k in ("sobol", "sobol_panel", "sobol_syrk", "sobol_L")}
    nc = info["columns"]
    Mp = -(-nc // 128) * 128 if nc else 0
    flop = Mp * (Mp + 1) * info["pair_rows"] if nc else 0
    syrk_ms = ph["sobol_syrk"][0] / reps
    print(f"path={info['path']}: {len(subsets)} terms, M={M}, D={D}, depth={R}: wall {wall:.2f} ms/call, device {ph['sobol'][0]/reps:.2f} ms "
          f"(L {ph['sobol_L'][0]/reps:.2f}, panel {ph['sobol_panel'][0]/reps:.2f}, syrk {syrk_ms:.2f}"
          + (f" = {flop/syrk_ms/1e9:.1f} TFLOP/s padded" if syrk_ms > 0 else "") + f"), columns {nc}, pairing disagreement {info['pairing_disagreement']:.2e}",
          flush=True)
ctx.sobol_set_path("auto")
if len(res) == 2:
    a, b = res["gram"], res["terms"]
    print(f"gram vs terms: max |diff| / max = {np.abs(a-b).max()/np.abs(b).max():.2e}; normalised max diff {np.abs(a/a.sum()-b/b.sum()).max():.2e}; "
          f"min term {a.min():.3e}")

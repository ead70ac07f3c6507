"""SYRK row-split sweep (OAK_SYRK_NSPLIT) for one configuration / row count: each setting in a child process (the knob is read
per call, but the partial buffer grows).  python tools/dev_syrk_sweep.py <config> <rows> <nsplit,nsplit,...>"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 4 and sys.argv[4] == "child":
    sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
    import numpy as np
    from oak import _capi
    import bench
    cfg = bench.CONFIGS[sys.argv[1]]
    n = int(sys.argv[2])
    N, D, M, R = cfg["N"], cfg["D"], cfg["M"], cfg["R"]
    X, y, Z = bench.synthetic(max(N, n), D, M)
    spec = bench.make_spec(D, R)
    ctx = _capi.default_context()
    ctx.sgpr_set_data(np.ascontiguousarray(X[:n]), np.ascontiguousarray(y[:n])); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    for _ in range(3):
        ctx.sgpr_elbo(_capi.KernelDesc(spec), 0.01, 1e-6)
    ctx.reset_timings()
    K = 20
    import time
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(K):
        ctx.sgpr_elbo(_capi.KernelDesc(spec), 0.01, 1e-6)
    ctx.sync(); dt = (time.perf_counter() - t0) / K
    ph = {k: round(ctx.timing(k)[0] / K, 4) for k in ("featurize", "gram", "syrk", "reduce", "tail", "total")}
    fl = M * (M + 1) * n
    print(f"nsplit={sys.argv[3]:>4s} wall={dt*1e3:7.3f} ms syrk={ph['syrk']:.4f} ms ({fl / ph['syrk'] / 1e9:.1f} TF/s) {ph}", flush=True)
else:
    for ns in sys.argv[3].split(","):
        env = dict(os.environ)
        if ns != "auto":
            env["OAK_SYRK_NSPLIT"] = ns
        subprocess.call([sys.executable, __file__, sys.argv[1], sys.argv[2], ns, "child"], env=env)

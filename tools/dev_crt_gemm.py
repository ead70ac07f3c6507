"""The gradient's adjoint panel G = Kfu H on the int8 pipe (csrc/crt_gemm.hip, OAK_CRT_GEMM) against the fp64 GEMM: gradient
deviation between the two and from the fp64-kernel route, forward+gradient step times and the backward phases.
python tools/dev_crt_gemm.py [--configs headline,c3] [--bits 44]"""
import argparse, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="headline")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--bits", default="")
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--int8-only", action="store_true", help="only the int8 GEMM mode (profiling runs)")
ap.add_argument("--cond-sweep", action="store_true", help="gradient error of the int8 and the fp64 adjoint GEMM against the whitened route's gradient over a range of conditioning")
args = ap.parse_args()
ctx = _capi.default_context()
if args.cond_sweep:
    from oracle import oak_oracle as o
    N, M, D = 65536, 768, 8
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N, D))
    y = (np.sin(X[:, 0]) + 0.5 * X[:, 1] * X[:, 2] + 0.1 * rng.standard_normal(N)).reshape(-1, 1); y = (y - y.mean()) / y.std()
    for ls, spread in ((0.5, 1.0), (0.7, 1.0), (1.0, 1.0), (1.5, 1.0), (1.0, 0.3), (1.5, 0.1), (1.5, 0.02), (3.0, 0.02)):
        Z = X[:M].copy()
        if spread < 1.0: Z[M // 2:] = Z[:M - M // 2] + spread * rng.standard_normal((M - M // 2, D))
        spec = o.make_spec(D, 2, lengthscales=[ls] * D); d = _capi.KernelDesc(spec)
        ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
        ctx.sgpr_set_route("whitened"); ctx.sgpr_set_precision("fp64"); _, gw, zw = ctx.sgpr_elbo_grad_z(d, 0.01, M, D)
        ctx.sgpr_set_route("phi"); _, g64, z64 = ctx.sgpr_elbo_grad_z(d, 0.01, M, D)
        ctx.sgpr_set_precision("int8crt")
        os.environ["OAK_CRT_GEMM"] = "0"; _, ga, za = ctx.sgpr_elbo_grad_z(d, 0.01, M, D); est = ctx.sgpr_last_terms()["cond_estimate"]
        os.environ["OAK_CRT_GEMM"] = "1"; _, gb, zb = ctx.sgpr_elbo_grad_z(d, 0.01, M, D); info = ctx.bench_crt_info()
        sc, sz = np.abs(gw).max(), np.abs(zw).max()
        print(f"ls {ls} spread {spread}: estimate {est:.3g}  |g - g_whitened| / max|g|: fp64 phi {np.abs(g64 - gw).max() / sc:.1e}, int8 Phi + fp64 GEMM {np.abs(ga - gw).max() / sc:.1e}, "
              f"int8 Phi + int8 GEMM {np.abs(gb - gw).max() / sc:.1e}   (tail_dd {info['tail_dd']})\n"
              f"      gradient w.r.t. Z, same order: {np.abs(z64 - zw).max() / sz:.1e}, {np.abs(za - zw).max() / sz:.1e}, {np.abs(zb - zw).max() / sz:.1e}", flush=True)
    os.environ.pop("OAK_CRT_GEMM", None)
    sys.exit(0)
PHASES = ("gram", "crt_syrk", "syrk", "tail", "bwd_tail", "bwd_gemm", "bwd_gram", "bwd_small", "total")
for name in args.configs.split(","):
    cfg = bench.CONFIGS[name]
    N, D, M, R = cfg["N"], cfg["D"], cfg["M"], cfg["R"]
    if args.rows: N = args.rows
    X, y, Z = bench.synthetic(N, D, M, mixed=cfg.get("mixed", False))
    spec = bench.make_spec(D, R, mixed=cfg.get("mixed", False))
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    res = {}
    modes = (("fp64 kernels", "fp64", "0"), ("int8 Phi, fp64 GEMM", "int8crt", "0"), ("int8 Phi, int8 GEMM", "int8crt", "1"))
    for label, mode, env in (modes[2:] if args.int8_only else modes):
        ctx.sgpr_set_precision(mode)
        os.environ["OAK_CRT_GEMM"] = env
        if args.bits: os.environ["OAK_CRT_GEMM_BITS"] = args.bits
        for _ in range(2):
            e, g = ctx.sgpr_elbo_grad(d, 0.01)
        ctx.sync(); ctx.reset_timings(); t0 = time.perf_counter()
        for _ in range(args.reps):
            e, g = ctx.sgpr_elbo_grad(d, 0.01)
        ctx.sync(); dt = (time.perf_counter() - t0) / args.reps
        ph = {}
        for p in PHASES:
            try:
                ms, cnt = ctx.timing(p)
                if cnt: ph[p] = round(ms / cnt, 3)
            except Exception:
                pass
        res[label] = (e, g, dt, ph)
        print(f"{name} N={N} M={M}: {label}: {dt * 1e3:.2f} ms  {ph}", flush=True)
    if args.int8_only: continue
    g0 = res["fp64 kernels"][1]
    for label in ("int8 Phi, fp64 GEMM", "int8 Phi, int8 GEMM"):
        g = res[label][1]
        print(f"   gradient of '{label}' vs fp64 kernels: max |d| / max |g| = {np.abs(g - g0).max() / np.abs(g0).max():.2e}, "
              f"max rel (entries > 1e-6 max) = {np.max(np.abs(g - g0)[np.abs(g0) > 1e-6 * np.abs(g0).max()] / np.abs(g0)[np.abs(g0) > 1e-6 * np.abs(g0).max()]):.2e}")
    ga, gb = res["int8 Phi, fp64 GEMM"][1], res["int8 Phi, int8 GEMM"][1]
    print(f"   int8 GEMM vs fp64 GEMM (same Phi): max |d| / max |g| = {np.abs(ga - gb).max() / np.abs(ga).max():.2e}")
    os.environ["OAK_CRT_GEMM"] = "0"
    ctx.sgpr_set_precision("auto")

"""int8 CRT statistics mode (oak_sgpr_set_precision("int8crt"), csrc/crt.hip) against the fp64 kernels on the benchmark
problems: deviation of Phi, of the ELBO and of every kernel-dependent term (between the two modes and, with --oracle, of each
against the multicore C oracle on a row sample), step times and the phase breakdown.
python tools/dev_crt.py [--oracle] [--configs c2,headline,c3,c5]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--configs", default="c2,headline,c3,c5")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--check-fused", action="store_true", help="the fused Gram epilogue against the stand-alone conversion pass: Phi must be bit-identical")
args = ap.parse_args()
ctx = _capi.default_context()
TERMS = ("sum_log_diag_LB", "cTc", "tr_AAT", "kappa", "logdet_Kuu")
PHASES = ("featurize", "gram", "syrk", "crt_convert", "crt_syrk", "crt_reduce", "reduce", "tail")
for name in args.configs.split(","):
    cfg = bench.CONFIGS[name]
    N, D, M, R = cfg["N"], cfg["D"], cfg["M"], cfg["R"]
    X, y, Z = bench.synthetic(N, D, M, mixed=cfg.get("mixed", False))
    spec = bench.make_spec(D, R, mixed=cfg.get("mixed", False))
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    out = {}
    for mode in ("fp64", "int8crt"):
        ctx.sgpr_set_precision(mode)
        for _ in range(2):
            e = ctx.sgpr_elbo(d, 0.01)
        ctx.sync(); ctx.reset_timings(); t0 = time.perf_counter()
        for _ in range(args.reps):
            e = ctx.sgpr_elbo(d, 0.01)
        ctx.sync(); dt = (time.perf_counter() - t0) / args.reps
        ph = {}
        for p in PHASES:
            try:
                ms, cnt = ctx.timing(p)
                if cnt: ph[p] = round(ms / cnt, 3)
            except Exception:
                pass
        out[mode] = dict(e=e, terms=ctx.sgpr_last_terms(), dt=dt, used=ctx.sgpr_stats_precision(), stats=ctx.sgpr_get_stats(), phases=ph)
    if args.check_fused:
        import os
        ctx.sgpr_set_precision("int8crt")
        os.environ["OAK_CRT_UNFUSED"] = "1"
        eu = ctx.sgpr_elbo(d, 0.01); su = ctx.sgpr_get_stats()
        del os.environ["OAK_CRT_UNFUSED"]
        ef = ctx.sgpr_elbo(d, 0.01); sf = ctx.sgpr_get_stats()
        print(f"   fused vs stand-alone conversion: Phi identical={np.array_equal(su[:M * M], sf[:M * M])} (max |d| {np.abs(su[:M * M] - sf[:M * M]).max():.3e}), "
              f"psi max rel d {np.abs(su[M * M:M * M + M] - sf[M * M:M * M + M]).max() / np.abs(su[M * M:M * M + M]).max():.2e}, ELBO {eu!r} {ef!r}", flush=True)
        eg, g = ctx.sgpr_elbo_grad(d, 0.01)           # fused pass that also writes the fp64 panel for the backward
        ctx.sgpr_set_precision("fp64")
        eg64, g64 = ctx.sgpr_elbo_grad(d, 0.01)
        print(f"   gradient call: ELBO rel d {abs(eg - eg64) / abs(eg64):.2e}, grad max rel d {np.abs(g - g64).max() / np.abs(g64).max():.2e}", flush=True)
    ctx.sgpr_set_precision("fp64")
    a, b = out["fp64"], out["int8crt"]
    P64, Pc = a["stats"][:M * M].reshape(M, M), b["stats"][:M * M].reshape(M, M)
    dg = np.sqrt(np.outer(np.diag(P64), np.diag(P64)))
    print(f"{name} N={N} M={M} D={D} R={R}: used={b['used']}  fp64 {a['dt']*1e3:.2f} ms  int8crt {b['dt']*1e3:.2f} ms ({a['dt']/b['dt']:.2f}x)", flush=True)
    print(f"   cond_est={b['terms']['cond_estimate']:.3g}  Phi: max |d| / sqrt(Phi_aa Phi_bb) = {np.abs(Pc - P64).max() and (np.abs(Pc - P64) / dg).max():.2e}, "
          f"symmetric={np.array_equal(Pc, Pc.T)}  ELBO rel diff {abs(b['e'] - a['e']) / abs(a['e']):.2e}")
    print("   terms (crt vs fp64): " + ", ".join(f"{k} {abs(b['terms'][k] - a['terms'][k]) / max(abs(a['terms'][k]), 1e-300):.1e}" for k in TERMS))
    print(f"   phases fp64 {a['phases']}\n   phases crt  {b['phases']}", flush=True)
    if args.oracle:
        from oracle import c_oracle
        ns = min(N, 65536)
        ref, parts = c_oracle.sgpr_elbo_chunked(spec, X[:ns], y[:ns], Z, 0.01, chunk=16384, return_parts=True)
        ctx.sgpr_set_data(np.ascontiguousarray(X[:ns]), np.ascontiguousarray(y[:ns]))
        for mode in ("fp64", "int8crt"):
            ctx.sgpr_set_precision(mode)
            e = ctx.sgpr_elbo(d, 0.01)
            t = ctx.sgpr_last_terms()
            print(f"   vs oracle on {ns} rows, {mode} (used {ctx.sgpr_stats_precision()}): ELBO {abs(e - ref) / abs(ref):.1e}  " +
                  ", ".join(f"{k} {abs(t[k] - parts['terms'][k]) / max(abs(parts['terms'][k]), 1e-300):.1e}" for k in TERMS if k in parts["terms"]), flush=True)
        ctx.sgpr_set_precision("fp64")

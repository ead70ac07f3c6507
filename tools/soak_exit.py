"""Many short-lived processes: does one ever hang, and if so before or after its work is done?  Each child evaluates a small
(partitioned) and a chunk-sized problem, prints DONE (flushed) and exits normally -- through the binding's exit hook.  A child that
exceeds --limit seconds is killed (by PID) and reported with what it had printed.  python tools/soak_exit.py [--n 60] [--limit 60]"""
import argparse, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, r"%s/orthogonal-additive-gaussian-processes_amd"); sys.path.insert(0, r"%s")
from oak import _capi
import bench
N, D, M = 32768, 8, 256
X, y, Z = bench.synthetic(N, D, M)
desc = _capi.KernelDesc(bench.make_spec(D, 2))
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
for _ in range(5):
    e = ctx.sgpr_elbo(desc, 0.01)
eg, g = ctx.sgpr_elbo_grad(desc, 0.01)
c2 = _capi.HipContext(0); c2.sgpr_set_data(X[:4000], y[:4000]); c2.sgpr_set_inducing(Z[:64]); c2.sgpr_elbo(desc, 0.02); c2.close()
print("DONE %%.6e" %% e, flush=True)
''' % (ROOT, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=60); ap.add_argument("--limit", type=float, default=60.0)
args = ap.parse_args()
hung_before = hung_after = 0
t0 = time.time()
for i in range(args.n):
    p = subprocess.Popen([sys.executable, "-c", CHILD], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        out, _ = p.communicate(timeout=args.limit)
        if p.returncode != 0 or "DONE" not in out:
            print(f"child {i}: rc={p.returncode}\n{out[-1500:]}", flush=True)
    except subprocess.TimeoutExpired:
        p.kill()
        out, _ = p.communicate()
        after = "DONE" in (out or "")
        hung_after += after; hung_before += (not after)
        print(f"child {i}: HUNG {'AFTER' if after else 'BEFORE'} its work was done; output tail: {(out or '')[-300:]!r}", flush=True)
print(f"{args.n} children in {time.time() - t0:.0f} s: {hung_before} hung before DONE, {hung_after} hung after DONE (at exit)")

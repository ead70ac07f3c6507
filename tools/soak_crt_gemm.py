"""Repeats gradient evaluations that take the int8 adjoint GEMM (csrc/crt_gemm.hip) and compares every result with the first BIT FOR BIT:
integer arithmetic and a fixed fold order make any mistake in the kernel's hand-written waits (counted vmcnt, lgkmcnt by inline asm) visible as a
difference.  A second context keeps the GPU busy with fp64-kernel evaluations from another host thread.
python tools/soak_crt_gemm.py [seconds] [rows]"""
import sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 196608
D, M, R = 16, 1024, 2
X, y, Z = bench.synthetic(rows, D, M)
spec = bench.make_spec(D, R)
d = _capi.KernelDesc(spec)
ctx = _capi.HipContext(0)
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi"); ctx.sgpr_set_precision("int8crt")
e0, g0 = ctx.sgpr_elbo_grad(d, 0.01)
assert ctx.bench_crt_info()["gemm_planes"] >= 13, ctx.bench_crt_info()
stop = False
def noise():
    c2 = _capi.HipContext(0)
    c2.sgpr_set_data(X[:65536], y[:65536]); c2.sgpr_set_inducing(Z[:512]); c2.sgpr_set_route("phi"); c2.sgpr_set_precision("fp64")
    d2 = _capi.KernelDesc(spec)
    while not stop:
        c2.sgpr_elbo_grad(d2, 0.01)
    c2.close()
th = threading.Thread(target=noise); th.start()
t0 = time.perf_counter(); n = bad = 0
while time.perf_counter() - t0 < secs:
    e, g = ctx.sgpr_elbo_grad(d, 0.01)
    n += 1
    if e != e0 or g.tobytes() != g0.tobytes():
        bad += 1
        print(f"iteration {n}: differs (max |dg| / max |g| = {np.abs(g - g0).max() / np.abs(g0).max():.3e})", flush=True)
stop = True; th.join()
print(f"{n} gradient evaluations of {rows} rows with the int8 adjoint GEMM next to a second context's load: {bad} differ from the first")
sys.exit(1 if bad else 0)

"""Cost of the inducing-input gradient at the headline shape, register-resident kernels against the general one, and of a grouped kernel:
python tools/dev_gradz.py        (OAK_BWDZ_GENERAL=1 in the environment selects the general kernel for every shape)"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
import bench
from oak import _capi
N, D, M, R = 1 << 18, 16, 1024, 2
X, y, Z = bench.synthetic(N, D, M)
spec = bench.make_spec(D, R)
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")


def timed(fn, k=3):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(k):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / k * 1e3


g = timed(lambda: ctx.sgpr_elbo_grad(_capi.KernelDesc(spec), 0.01))
gz = timed(lambda: ctx.sgpr_elbo_grad_z(_capi.KernelDesc(spec), 0.01, M, D))
print(f"N={N} D={D} M={M} R={R} general={os.environ.get('OAK_BWDZ_GENERAL', '0')}: forward+gradient {g:.2f} ms, with the inducing inputs {gz:.2f} ms (+{gz - g:.2f})", flush=True)
# the same columns as 8 groups of two (unconstrained RBFs): fused Gram with extra feature rows, general backward kernels
gspec = dict(dims=[dict(type="rbf", lengthscale=1.5, variance=1.0, measure=None, active_dim=2 * k, active_dims=[2 * k, 2 * k + 1]) for k in range(8)],
             order_variances=[1.0, 0.8, 0.5], max_interaction_depth=2, share_var_across_orders=True)
f = timed(lambda: ctx.sgpr_elbo(_capi.KernelDesc(gspec), 0.01))
g = timed(lambda: ctx.sgpr_elbo_grad(_capi.KernelDesc(gspec), 0.01))
gz = timed(lambda: ctx.sgpr_elbo_grad_z(_capi.KernelDesc(gspec), 0.01, M, D))
print(f"8 groups of 2 columns: forward {f:.2f} ms, forward+gradient {g:.2f} ms, with the inducing inputs {gz:.2f} ms", flush=True)

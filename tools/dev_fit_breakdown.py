"""Where a BFGS evaluation of the model API spends its time: GPU phases (library timers) against wall time per evaluation, and the
route each evaluation took.  python tools/dev_fit_breakdown.py [N D M maxiter]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd"))
from oak import gpflow_lite as gpflow
from oak.model_utils import oak_model
from oak import _capi

N, D, M, maxiter = [int(a) for a in (sys.argv[1:5] + [1048576, 16, 1024, 8][len(sys.argv) - 1:])][:4]
rng = np.random.default_rng(1)
X = rng.normal(size=(N, D))
y = (np.sin(X[:, 0]) + 0.5 * X[:, 1] ** 2 + 0.8 * X[:, 2] * X[:, 3] + 0.1 * rng.normal(size=N))[:, None]
oak = oak_model(max_interaction_depth=2, num_inducing=M, sparse=True, use_normalising_flow=False)
oak.fit(X, y, optimise=False)
ctx = oak.m._hip                      # the model owns its device context
closure = oak.m.training_loss_closure()
oak.m.training_loss()
ctx.reset_timings()
t0 = time.perf_counter()
res = gpflow.Scipy().minimize(closure, oak.m.trainable_variables, method="BFGS", on_linalg_error="inf", options={"maxiter": maxiter})
dt = time.perf_counter() - t0
names = ["featurize", "gram", "trsm", "syrk", "reduce", "tail", "total", "bwd_gemm", "bwd_gram", "bwd_tail", "bwd_small"]
ph = {k: ctx.timing(k) for k in names}
n = max(res.nfev, 1)
print(f"{res.nfev} evaluations, {dt / n * 1e3:.1f} ms wall each; GPU 'total' {ph['total'][0] / max(ph['total'][1], 1):.1f} ms x {ph['total'][1]} calls")
print({k: (round(v[0] / n, 2), v[1]) for k, v in ph.items()})
print("last terms:", ctx.sgpr_last_terms())

"""End-to-end use of the drop-in model API on a synthetic regression problem:
    oak_model.fit (normalising flows, device k-means inducing points, BFGS on the HIP ELBO + analytic gradient)
    -> predict -> Sobol indices.
usage: python examples/fit_synthetic.py [N] [D] [M] [maxiter]
"""
import sys, time
from pathlib import Path
import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd"))
from oak import gpflow_lite as gpflow
from oak.model_utils import oak_model
from oak import _capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M = int(sys.argv[3]) if len(sys.argv) > 3 else 512
maxiter = int(sys.argv[4]) if len(sys.argv) > 4 else 30

rng = np.random.default_rng(1)
X = rng.normal(size=(N, D))
f = np.sin(X[:, 0]) + 0.5 * X[:, 1] ** 2 + 0.8 * X[:, 2] * X[:, 3]          # two main effects + one interaction
y = (f + 0.1 * rng.normal(size=N))[:, None]
Xt = rng.normal(size=(20_000, D))
ft = np.sin(Xt[:, 0]) + 0.5 * Xt[:, 1] ** 2 + 0.8 * Xt[:, 2] * Xt[:, 3]

t0 = time.perf_counter()
oak = oak_model(max_interaction_depth=2, num_inducing=M, sparse=True, use_normalising_flow=False)
oak.fit(X, y, optimise=False)
t1 = time.perf_counter()
print(f"fit(optimise=False): {t1 - t0:.2f} s  (scaling, {M}-centre k-means on {N} rows, model build)")
loss0 = oak.m.training_loss()
ctx = _capi.default_context()
ctx.reset_timings()
t2 = time.perf_counter()
res = gpflow.Scipy().minimize(oak.m.training_loss_closure(), oak.m.trainable_variables, method="BFGS",
                              on_linalg_error="inf", options={"maxiter": maxiter})
t3 = time.perf_counter()
print(f"BFGS: {res.nit} iterations, {res.nfev} loss+gradient evaluations in {t3 - t2:.2f} s "
      f"({(t3 - t2) / max(res.nfev, 1) * 1e3:.1f} ms per evaluation); loss {loss0:.1f} -> {res.fun:.1f}")
t4 = time.perf_counter()
pred = oak.predict(Xt)
t5 = time.perf_counter()
rmse = float(np.sqrt(np.mean((pred - ft) ** 2)))
print(f"predict {Xt.shape[0]} rows: {t5 - t4:.3f} s, RMSE vs noise-free target {rmse:.4f} (noise sd 0.1)")
t6 = time.perf_counter()
oak.get_sobol()
t7 = time.perf_counter()
order = np.argsort(oak.normalised_sobols)[::-1][:5]
print(f"Sobol ({len(oak.normalised_sobols)} terms): {t7 - t6:.3f} s; top terms:",
      [(oak.tuple_of_indices[i], round(float(oak.normalised_sobols[i]), 3)) for i in order])

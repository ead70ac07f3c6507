import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import gpflow_lite as gpflow
from oak.model_utils import oak_model
from oak import _capi
from oak.oak_kernel import kernel_to_spec
from oracle import oak_oracle as o
N, D, M = 200000, 8, 512
rng = np.random.default_rng(1)
X = rng.normal(size=(N, D))
f = np.sin(X[:, 0]) + 0.5 * X[:, 1] ** 2 + 0.8 * X[:, 2] * X[:, 3]
y = (f + 0.1 * rng.normal(size=N))[:, None]
oak = oak_model(max_interaction_depth=2, num_inducing=M, sparse=True, use_normalising_flow=False)
oak.fit(X, y, optimise=False)
hist = []
clos = oak.m.training_loss_closure()
class Wrap:
    def __call__(self): return clos()
    def value_and_grad(self, variables):
        try:
            l, g = clos.value_and_grad(variables)
        except Exception as e:
            spec = kernel_to_spec(oak.m.kernel)
            print("FAILED:", e)
            print("lengthscales", [d.get("lengthscale") for d in spec["dims"]])
            print("order var", spec["order_variances"], "noise", oak.m.likelihood.variance.numpy())
            Z = oak.m.inducing_variable.Z.numpy()
            Kuu = o.oak_K(spec, Z, Z) if hasattr(o, "oak_K") else None
            if Kuu is not None:
                ev = np.linalg.eigvalsh(Kuu + 1e-6 * np.eye(M))
                print("eig min/max of Kuu+jitter:", ev[0], ev[-1])
            raise
        hist.append(l)
        print(len(hist), l, flush=True)
        return l, g
try:
    gpflow.Scipy().minimize(Wrap(), oak.m.trainable_variables, method="BFGS", options={"maxiter": 30})
except Exception as e:
    print("stopped:", type(e).__name__)

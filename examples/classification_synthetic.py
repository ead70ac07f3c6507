"""Binary classification with the OAK kernel: the flow of the reference's examples/uci/uci_classification_train.py
(oak_model.fit builds the kernel and the input transform, an SVGP with a Bernoulli likelihood is trained on it by BFGS,
then accuracy / NLL / Sobol indices) on synthetic data, every N-sized step on the MI355X.

    python examples/classification_synthetic.py [N] [M]
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "orthogonal-additive-gaussian-processes_amd"))
from oak import gpflow_lite as gpflow                       # noqa: E402
from oak.gpflow_lite import inv_logit, set_trainable         # noqa: E402
from oak.model_utils import oak_model                        # noqa: E402
from oak.utils import kmeans_centres                         # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = np.random.default_rng(4)
    X = rng.normal(size=(N + 2000, 5))
    f = 2.0 * np.sin(X[:, 0]) + 1.5 * X[:, 1] * X[:, 2]
    y = (rng.uniform(size=len(X)) < 1.0 / (1.0 + np.exp(-3 * f))).astype(float)[:, None]
    Xtr, ytr, Xte, yte = X[:N], y[:N], X[N:], y[N:]
    oak = oak_model(max_interaction_depth=2, num_inducing=M, sparse=True)
    oak.fit(Xtr, ytr, optimise=False, initialise_inducing_points=False)
    data = (np.asarray(oak.m.data[0]), ytr)
    t0 = time.time()
    Z = kmeans_centres(data[0], M)
    oak.m = gpflow.models.SVGP(kernel=oak.m.kernel, likelihood=gpflow.likelihoods.Bernoulli(invlink=inv_logit),
                               inducing_variable=Z, whiten=True, q_diag=True)
    set_trainable(oak.m.inducing_variable, False)
    res = gpflow.optimizers.Scipy().minimize(oak.m.training_loss_closure(data), oak.m.trainable_variables, method="BFGS",
                                             options={"maxiter": 200})
    XT = oak._transform_x(Xte)
    mu, _ = oak.m.predict_f(XT)
    err = np.mean((np.asarray(inv_logit(mu)) > 0.5).astype(float) != yte)
    nll = -np.asarray(oak.m.predict_log_density((XT, yte))).mean()
    print(f"N={N} M={M}: {res.nit} BFGS iterations ({res.nfev} ELBO+gradient evaluations) in {time.time() - t0:.1f} s")
    print(f"test error {err:.4f}, nll {nll:.4f}")
    oak.m.data = data
    oak.get_sobol()
    order = np.argsort(oak.normalised_sobols)[::-1][:4]
    for j in order:
        print(f"  term {tuple(int(i) for i in oak.tuple_of_indices[j])}: normalised Sobol {oak.normalised_sobols[j]:.4f}")


if __name__ == "__main__":
    main()

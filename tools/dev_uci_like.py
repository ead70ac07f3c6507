"""The reference's own workflow at the size of its UCI examples (default options: normalising flows, k-means inducing points, depth = inputs,
BFGS with the default iteration cap): where the wall time goes.  python tools/dev_uci_like.py [N] [D]"""
import cProfile, pstats, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd"))
from oak.model_utils import oak_model
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(3)
X = rng.normal(size=(N, D)) * rng.uniform(0.5, 3, D) + rng.normal(size=D)
y = (np.sin(X[:, 0]) + 0.3 * X[:, 1] * X[:, 2] + 0.1 * rng.normal(size=N))[:, None]
oak_model(max_interaction_depth=2, num_inducing=50).fit(X[:500], y[:500])        # library load, first-call costs
oak = oak_model(max_interaction_depth=D, num_inducing=200)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
oak.fit(X, y)
pr.disable()
t1 = time.perf_counter()
mean = oak.predict(X[:2000])
t2 = time.perf_counter()
oak.get_sobol()
t3 = time.perf_counter()
print(f"N={N} D={D} depth={D} M=200: fit {t1 - t0:.2f} s, predict(2000) {t2 - t1:.3f} s, get_sobol {t3 - t2:.3f} s ({len(oak.normalised_sobols)} terms)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

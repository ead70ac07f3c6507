import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oak import _capi
from oracle import oak_oracle as o
hip = _capi.default_context()
for ls in (1e-3, 3e-2, 50.0, 1e3):
    rng = np.random.default_rng(int(ls * 1000) % 97)
    D, R = 4, 2
    spec = o.make_spec(D, R, lengthscales=[ls, ls * 1.5, 1.0, ls], order_variances=[0.8, 1.1, 0.6])
    X, X2 = rng.standard_normal((150, D)), rng.standard_normal((70, D))
    X2[:5] = X[:5]
    d = _capi.KernelDesc(spec)
    got, ref = hip.gram(d, X, X2), o.oak_K(spec, X, X2)
    err = np.abs(got - ref); i, j = np.unravel_index(err.argmax(), err.shape)
    print(ls, "max err", err.max(), "at", (i, j), "got", got[i, j], "ref", ref[i, j], "max ref", np.abs(ref).max())
    # per-dimension check of the worst pair
    for dd in range(D):
        kd = o.base_K(X[[i]][:, [dd]], X2[[j]][:, [dd]], spec["dims"][dd])[0, 0]
        print("   dim", dd, "x", X[i, dd], "z", X2[j, dd], "k_d(oracle)", kd)

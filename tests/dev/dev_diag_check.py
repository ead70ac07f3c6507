import sys
from pathlib import Path
import numpy as np, mpmath as mp
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
from oracle import oak_oracle as o
mp.mp.dps = 40
rng = np.random.default_rng(5)
variances = (2.0 ** -20, 2.0 ** 20, 0.999); ls = [0.01, 0.05, 0.002]
spec = o.make_spec(3, 2, lengthscales=ls); spec["share_var_across_orders"] = False; spec["order_variances"] = [0.8]
for d_, v in zip(spec["dims"], variances): d_["variance"] = v
X = rng.normal(size=(300, 3)) * 2.0
hip = _capi.default_context()
kd = hip.gram_diag(_capi.KernelDesc(spec), X); kr = o.oak_K_diag(spec, X)
i = int(np.argmax(np.abs(kd - kr)))
def kdiag_mp(x):
    ks = []
    for d in range(3):
        l, bv, xx = mp.mpf(ls[d]), mp.mpf(variances[d]), mp.mpf(float(x[d]))
        cov = bv * l / mp.sqrt(l * l + 1) * mp.e ** (-(xx ** 2) / (2 * (l * l + 1)))
        var = bv * l / mp.sqrt(l * l + 2)
        ks.append(bv - cov * cov / var)
    e1 = sum(ks); e2 = ks[0] * ks[1] + ks[0] * ks[2] + ks[1] * ks[2]
    return mp.mpf("0.8") + e1 + e2
t = kdiag_mp(X[i])
print("row", i, "hip", repr(kd[i]), "oracle", repr(kr[i]), "mp", mp.nstr(t, 20), "hip rel", float(abs(kd[i] - t) / t), "oracle rel", float(abs(kr[i] - t) / t))

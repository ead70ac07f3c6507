"""Full-GP (GPR) path timing at a few sizes (config C1 is plumbing-sized; this is a sanity sweep)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
from oracle import oak_oracle as o
ctx = _capi.default_context()
for N in (1030, 4096, 8192, 16384):
    D = 8
    X, y, _ = o.synthetic_problem(N, D, 8, seed=1)
    spec = o.make_spec(D, 2)
    d = _capi.KernelDesc(spec)
    ctx.gpr_set_data(X, y)
    ctx.gpr_log_marginal(d, 0.1)
    t0 = time.perf_counter(); lml = ctx.gpr_log_marginal(d, 0.1); t1 = time.perf_counter()
    lg, g = ctx.gpr_log_marginal_grad(d, 0.1); t2 = time.perf_counter()
    ref = o.gpr_log_marginal_likelihood(spec, X, y, 0.1) if N <= 4096 else float("nan")
    print(f"N={N}: lml {lml:.6f} (oracle {ref:.6f}) fwd {1e3*(t1-t0):.1f} ms, fwd+grad {1e3*(t2-t1):.1f} ms")

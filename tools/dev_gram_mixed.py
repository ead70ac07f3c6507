"""Resident Gram pass at the C5 shape: all-continuous against 20 continuous + 8 binary + 4 categorical sub-kernels, and the discrete
share alone: python tools/dev_gram_mixed.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
import bench
from oak import _capi
N, D, M, R = 262144, 32, 2048, 4
X, y, Z = bench.synthetic(N, D, M, mixed=True)
ctx = _capi.default_context()
ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)


def timed(spec, k=5):
    d = _capi.KernelDesc(spec)
    ctx.bench_gram_resident(d); ctx.sync(); t0 = time.perf_counter()
    for _ in range(k):
        ctx.bench_gram_resident(d)
    ctx.sync()
    return (time.perf_counter() - t0) / k * 1e3


mixed = bench.make_spec(D, R, mixed=True)
cont = bench.make_spec(D, R)
for name, spec in (("32 continuous", cont), ("20 continuous + 8 binary + 4 categorical", mixed)):
    print(f"{name}: {timed(spec):.2f} ms", flush=True)
for name, sel in (("20 continuous only", range(20)), ("8 binary only", range(20, 28)), ("4 categorical only", range(28, 32)), ("12 discrete only", range(20, 32))):
    s = dict(mixed); s["dims"] = [dict(mixed["dims"][d], active_dim=d) for d in sel]
    print(f"{name}: {timed(s):.2f} ms", flush=True)

"""Stress of independent contexts on several host threads (the scenario of test_independent_contexts_are_thread_safe), repeated;
a watchdog dumps every thread's Python stack if a round does not finish.  python tools/dev_threads.py [rounds] [close_in_thread]"""
import faulthandler, sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
from oak import _capi
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
close_in_thread = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
probs = []
for t in range(4):
    X, y, Z = bench.synthetic(3000 + 517 * t, 3 + t, 64 + 32 * t)
    probs.append((X, y, Z, bench.make_spec(3 + t, 2)))
ctxs = []


def run(p, reps):
    X, y, Z, spec = p
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    for _ in range(reps):
        ctx.sgpr_elbo_grad(d, 0.02)
        ctx.sgpr_predict(d, X[:300])
        ctx.gram(d, X[:200], Z)
    if close_in_thread:
        ctx.close()
    else:
        ctxs.append(ctx)


for r in range(rounds):
    faulthandler.dump_traceback_later(60, exit=True)
    ths = [threading.Thread(target=run, args=(p, 6)) for p in probs]
    t0 = time.time()
    for th in ths: th.start()
    for th in ths: th.join()
    faulthandler.cancel_dump_traceback_later()
    for c in ctxs: c.close()
    ctxs.clear()
    if r % 10 == 0: print(f"round {r} ok ({time.time() - t0:.2f} s)", flush=True)
print("all rounds finished")

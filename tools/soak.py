"""Soak of the three scenarios an intermittent stall was once seen around, many times in ONE process:
  threads   four host threads, each with its own context: objective + gradient + prediction + explicit Gram, contexts closed
            in their threads (tests/test_gpu_sgpr.py::test_independent_contexts_are_thread_safe)
  rccl      a 1-rank RCCL communicator created, used for an objective and a gradient, destroyed
  streams   the side-stream factorisation chain next to the N-sized Gram / SYRK at a size where they overlap, both routes
Every iteration runs under a watchdog on another thread: if it does not finish within --limit seconds, the library's own view
of its contexts (oak_debug_state: busy streams, last phases / collectives) and every Python stack go to gpurun_out/soak_<tag>.txt
and the process exits with status 3 -- it never re-executes itself and never waits forever.

    python tools/soak.py [--iters 200] [--limit 120] [--tag r03] [--only threads,rccl,streams]
"""
import argparse, faulthandler, os, sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "orthogonal-additive-gaussian-processes_amd")); sys.path.insert(0, str(ROOT))
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
from oak import _capi
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--limit", type=float, default=120.0)
ap.add_argument("--tag", default="run")
ap.add_argument("--only", default="threads,rccl,streams")
args = ap.parse_args()
out_dir = ROOT / "gpurun_out"
out_dir.mkdir(exist_ok=True)
log = open(out_dir / f"soak_{args.tag}.txt", "w")


def say(msg):
    print(msg, flush=True)
    log.write(msg + "\n"); log.flush()


def native_backtraces():
    """Every thread's C-level stack through a debugger started as a CHILD process (never an exec of this one): where inside the
    HIP runtime a wedged call sits.  The child may attach because this process names it as its tracer (Yama ptrace_scope = 1)."""
    import ctypes, shutil, subprocess
    gdb = shutil.which("rocgdb") or shutil.which("gdb") or "/opt/rocm/bin/rocgdb"
    try:
        ctypes.CDLL(None).prctl(0x59616d61, ctypes.c_ulong(-1 & 0xffffffffffffffff), 0, 0, 0)      # PR_SET_PTRACER, PR_SET_PTRACER_ANY
        r = subprocess.run([gdb, "-p", str(os.getpid()), "-batch", "-ex", "set pagination off", "-ex", "info sharedlibrary", "-ex", "thread apply all bt 25"],
                           capture_output=True, text=True, timeout=90)
        return r.stdout[-60000:] + "\n" + r.stderr[-3000:]
    except Exception as ex:                       # noqa: BLE001
        return f"(no native backtrace: {ex!r})"


def stalled(what):
    say(f"STALL: {what} exceeded {args.limit:.0f} s")
    try:
        say(_capi.debug_state())
    except Exception as ex:                       # noqa: BLE001
        say(f"(no library state: {ex!r})")
    faulthandler.dump_traceback(file=log, all_threads=True)
    log.flush()
    say(native_backtraces())
    os._exit(3)


probs = []
for t in range(4):
    X, y, Z = bench.synthetic(3000 + 517 * t, 3 + t, 64 + 32 * t)
    probs.append((X, y, Z, bench.make_spec(3 + t, 2)))
Xs, ys, Zs = bench.synthetic(131072, 16, 1024)
spec_s = bench.make_spec(16, 2)


def scenario_threads():
    def run(p):
        X, y, Z, spec = p
        ctx = _capi.HipContext(0)
        d = _capi.KernelDesc(spec)
        ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
        for _ in range(3):
            ctx.sgpr_elbo_grad(d, 0.02)
            ctx.sgpr_predict(d, X[:300])
            ctx.gram(d, X[:200], Z)
        ctx.close()
    ths = [threading.Thread(target=run, args=(p,)) for p in probs]
    [t.start() for t in ths]
    [t.join() for t in ths]


def scenario_rccl():
    X, y, Z, spec = probs[1]
    ctx = _capi.HipContext(0)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z)
    ctx.comm_init(_capi.HipContext.comm_unique_id(), 1, 0)
    ctx.sgpr_elbo(d, 0.02)
    ctx.sgpr_elbo_grad(d, 0.02)
    ctx.comm_destroy()
    ctx.close()


stream_ctx = None


def scenario_streams():
    global stream_ctx
    if stream_ctx is None:
        stream_ctx = _capi.HipContext(0)
        stream_ctx.sgpr_set_data(Xs, ys); stream_ctx.sgpr_set_inducing(Zs)
    d = _capi.KernelDesc(spec_s)
    for route in ("phi", "whitened", "auto"):
        stream_ctx.sgpr_set_route(route)
        stream_ctx.sgpr_elbo(d, 0.01)
    stream_ctx.sgpr_elbo_grad(d, 0.01)


SCEN = {"threads": scenario_threads, "rccl": scenario_rccl, "streams": scenario_streams}
todo = [s for s in args.only.split(",") if s]
t_all = time.time()
worst = {s: 0.0 for s in todo}
for it in range(args.iters):
    for name in todo:
        timer = threading.Timer(args.limit, stalled, args=(f"iteration {it}, scenario {name}",))
        timer.daemon = True
        timer.start()
        t0 = time.time()
        SCEN[name]()
        dt = time.time() - t0
        timer.cancel()
        worst[name] = max(worst[name], dt)
    if it % 20 == 0 or it == args.iters - 1:
        say(f"iteration {it}: ok, {time.time() - t_all:.0f} s so far, slowest " + ", ".join(f"{k} {v:.2f} s" for k, v in worst.items()))
say(f"soak finished: {args.iters} iterations of {todo} in {time.time() - t_all:.0f} s, no stall; slowest " +
    ", ".join(f"{k} {v:.2f} s" for k, v in worst.items()))

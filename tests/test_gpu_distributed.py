"""GPU: the multi-GPU plumbing that can be exercised on a one-GPU box: RCCL is dlopen'ed, a 1-rank communicator is
created, the reduce-scatter + all-gather exchange is the identity, and the host-reducer path of ShardedSGPR (the one the
2-rank gloo CPU test drives with oracle statistics) reproduces the single-context ELBO from HIP statistics."""
import numpy as np
import pytest

from oak import _capi
from oak import distributed as D
from oracle import oak_oracle as o

pytestmark = pytest.mark.gpu


def _rccl_init(ctx, uid, nranks, rank, timeout=120.0):
    """ncclCommInitRank on a helper thread: a box whose RCCL bootstrap stalls fails THIS test after `timeout` seconds instead of
    holding the whole suite until the watchdog ends the process (the call itself cannot be interrupted)."""
    import threading
    box = {}

    def run():
        try:
            ctx.comm_init(uid, nranks, rank)
            box["ok"] = True
        except Exception as e:      # noqa: BLE001
            box["err"] = e

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        pytest.fail(f"RCCL communicator creation did not return within {timeout:.0f} s on this box")
    if "err" in box:
        raise box["err"]


def test_single_rank_rccl_exchange_is_identity():
    ctx = _capi.HipContext(0)
    X, y, Z = o.synthetic_problem(3000, 5, 100)
    spec = o.make_spec(5, 2)
    d = _capi.KernelDesc(spec)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("phi")
    e0 = ctx.sgpr_elbo(d, 0.01)
    uid = _capi.HipContext.comm_unique_id()
    assert len(uid) == 128
    _rccl_init(ctx, uid, 1, 0)
    assert ctx.sgpr_elbo(d, 0.01) == e0
    e1, g1 = ctx.sgpr_elbo_grad(d, 0.01)
    assert e1 == e0 and np.all(np.isfinite(g1))
    v = np.arange(5.0)
    np.testing.assert_array_equal(ctx.comm_allreduce_host(v.copy()), v)
    np.testing.assert_array_equal(ctx.comm_allgatherv(v, [5]), v)
    # which library was loaded: ROCm's own (the one rccl.h under $ROCM_PATH belongs to), same major version as the header
    info = _capi.HipContext.comm_info()
    import os
    assert info["path"].startswith(os.environ.get("ROCM_PATH", "/opt/rocm")), info
    major = lambda v: v // 10000 if v >= 10000 else v // 1000
    assert info["version"] > 0 and major(info["version"]) == major(info["header_version"]), info
    ctx.close()


def test_host_exchange_communicator_runs_every_collective_of_the_library():
    """oak_comm_init_host with a callback that plays the second rank (it adds the OTHER shard's contribution, computed
    beforehand on the same GPU): forward statistics, the gradient record and the all-gather all go through the callback, and the
    sharded result equals the single-context one."""
    X, y, Z = o.synthetic_problem(6001, 5, 96, seed=3)
    spec = o.make_spec(5, 2, lengthscales=[0.9, 1.1, 1.3, 0.8, 1.0])
    d = _capi.KernelDesc(spec)
    ref = _capi.HipContext(0)
    ref.sgpr_set_data(X, y); ref.sgpr_set_inducing(Z); ref.sgpr_set_route("whitened")
    e_ref, g_ref = ref.sgpr_elbo_grad(d, 0.02)
    # rank 1's buffers, recorded from a context that runs as "rank 1" with a callback that only records
    seen = []
    other = _capi.HipContext(0)
    other.sgpr_set_data(X[3000:], y[3000:]); other.sgpr_set_inducing(Z); other.sgpr_set_route("whitened")
    other.sgpr_set_global_rows(len(X))
    mine = _capi.HipContext(0)
    mine.sgpr_set_data(X[:3000], y[:3000]); mine.sgpr_set_inducing(Z); mine.sgpr_set_route("whitened")
    mine.sgpr_set_global_rows(len(X))
    # the two "ranks" run in lock step on two host threads; the callback is a two-party sum through a barrier
    import threading
    bar = threading.Barrier(2)
    slots = [None, None]

    def make_cb(r):
        def cb(a):
            slots[r] = a
            bar.wait()
            out = slots[0] + slots[1]
            bar.wait()
            return out
        return cb
    mine.comm_init_host(2, 0, make_cb(0))
    other.comm_init_host(2, 1, make_cb(1))
    res = {}

    def run(r, c):
        res[r] = c.sgpr_elbo_grad(d, 0.02)
        res[("gather", r)] = c.comm_allgatherv(np.full(r + 2, float(r)), [2, 3])
    th = [threading.Thread(target=run, args=(r, c)) for r, c in enumerate((mine, other))]
    [t.start() for t in th]; [t.join(120) for t in th]
    assert not any(t.is_alive() for t in th)
    for r in (0, 1):
        e, g = res[r]
        assert abs(e - e_ref) <= 1e-12 * abs(e_ref)
        np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-9 * np.abs(g_ref).max())
        np.testing.assert_array_equal(res[("gather", r)], [0, 0, 1, 1, 1])
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])      # bit-identical on both ranks
    for c in (mine, other, ref):
        c.close()


def _model_worker(rank, world, port, out_dir):
    import contextlib, io, traceback
    from pathlib import Path
    try:
        with contextlib.redirect_stdout(io.StringIO()):          # print_summary of the model is not wanted in the test log
            _model_worker_body(rank, world, port, out_dir)
    except BaseException:                                       # noqa: BLE001
        (Path(out_dir) / f"err{world}_{rank}.txt").write_text(traceback.format_exc())
        raise


def _model_worker_body(rank, world, port, out_dir):
    import os, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for p in (str(root / "orthogonal-additive-gaussian-processes_amd"), str(root), str(root / "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OAK_HIP_DEVICE="0")
    from oak import distributed as D2
    from oak import gpflow_lite as gpflow
    from oak.model_utils import oak_model
    D2.init_from_env(exchange="host")               # both ranks share GPU 0: the sums go through the TCP control plane
    rng = np.random.default_rng(5)
    X = rng.normal(size=(30001, 4))
    y = (np.sin(X[:, 0]) + 0.5 * X[:, 1] ** 2 + 0.7 * X[:, 2] * X[:, 3] + 0.1 * rng.normal(size=len(X)))[:, None]
    m = oak_model(max_interaction_depth=2, num_inducing=64, sparse=True)
    m.fit(X, y, optimise=False)                     # flows, k-means inducing points: the reference's default path
    closure = m.m.training_loss_closure()
    variables = m.m.trainable_variables
    loss0, grad0 = closure.value_and_grad(variables)
    # a second, non-trivial point (all hyper-parameters different from each other and from their initial values)
    for k, p in enumerate(m.m.trainable_parameters):
        p.assign(np.asarray(p.numpy()) * (0.6 + 0.25 * k))
    loss1, grad1 = closure.value_and_grad(variables)
    gpflow.Scipy().minimize(closure, variables, method="BFGS", options={"maxiter": 3})
    params3 = np.concatenate([np.ravel(p.numpy()) for p in m.m.trainable_parameters])
    m.m.SHARDED_PREDICT_MIN_ROWS = 16
    sobol3, pred3 = m.get_sobol(), m.predict(X[:1000])
    m.optimise()                                    # ... and on to convergence, as oak_model.fit(optimise=True) does
    params = np.concatenate([np.ravel(p.numpy()) for p in m.m.trainable_parameters])
    np.savez(Path(out_dir) / f"m{world}_{rank}.npz", loss0=loss0, grad0=np.concatenate([np.ravel(g) for g in grad0]), params3=params3,
             loss1=loss1, grad1=np.concatenate([np.ravel(g) for g in grad1]),
             sobol3=sobol3, pred3=pred3, params=params, sobol=m.get_sobol(), pred=m.predict(X[:1000]), loss=m.m.training_loss(),
             whitened=m.m._hip.sgpr_stats_whitened(), rows=len(X))
    D2.shutdown()


def test_oak_model_fit_row_sharded_over_two_ranks_equals_the_single_rank_fit(tmp_path):
    """oak_model.fit + BFGS + get_sobol + predict as two processes under oak.distributed.init_from_env (both on GPU 0, host
    exchange): every rank keeps half of the rows on the device, BFGS runs replicated on all-reduced statistics and gradient
    records (model_utils.py:249-408, 429-443, 499-524).  Objective and gradient equal the single-process ones to rounding --
    1e-11 / 1e-9 at the initial hyper-parameters AND at a second point where they all differ (small problem => the auto
    route whitens, so the order of the sums is not amplified by cond(Kuu)).  A BFGS trajectory is not a continuous function of
    its inputs at that level: measured here, the 1e-9 gradient differences (two shards sum in another order than one) have
    grown to 4e-3 in the hyper-parameters after only three iterations, so the trajectories and the optimum are compared to
    a few percent, not to 1e-9; what IS exact is that the two ranks stay bit-identical throughout."""
    import multiprocessing as mp
    import socket
    ctxm = mp.get_context("spawn")

    def launch(world):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        ps = [ctxm.Process(target=_model_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
        [p.start() for p in ps]
        [p.join(900) for p in ps]
        errs = "\n".join(f.read_text() for f in sorted(tmp_path.glob("err*.txt")))
        assert all(p.exitcode == 0 for p in ps), ([p.exitcode for p in ps], errs[-3000:])
    launch(1)
    launch(2)
    ref = np.load(tmp_path / "m1_0.npz")
    got = [np.load(tmp_path / f"m2_{r}.npz") for r in range(2)]
    live = np.abs(ref["params"]) > 1e-6             # a variance the optimiser drove towards zero (1e-12 .. 1e-40) carries no information
    for g in got:
        assert abs(g["loss0"] - ref["loss0"]) <= 1e-11 * abs(ref["loss0"])
        np.testing.assert_allclose(g["grad0"], ref["grad0"], rtol=1e-9, atol=1e-9 * np.abs(ref["grad0"]).max())
        assert abs(g["loss1"] - ref["loss1"]) <= 1e-11 * abs(ref["loss1"])
        np.testing.assert_allclose(g["grad1"], ref["grad1"], rtol=1e-9, atol=1e-9 * np.abs(ref["grad1"]).max())
        np.testing.assert_allclose(g["params3"], ref["params3"], rtol=3e-2)
        np.testing.assert_allclose(g["sobol3"], ref["sobol3"], atol=3e-2)
        np.testing.assert_allclose(g["pred3"], ref["pred3"], atol=3e-2 * np.abs(ref["pred3"]).max())
        assert abs(g["loss"] - ref["loss"]) <= 1e-3 * abs(ref["loss"])
        np.testing.assert_allclose(g["params"][live], ref["params"][live], rtol=5e-2)
        np.testing.assert_allclose(g["sobol"], ref["sobol"], atol=3e-2)
    for k in ("loss0", "grad0", "loss1", "grad1", "params3", "sobol3", "pred3", "params", "sobol", "pred", "loss"):
        np.testing.assert_array_equal(got[0][k], got[1][k])


def test_sharded_sgpr_host_reducer_matches_single_context():
    """Two row shards evaluated one after the other on the same GPU, reduced on the host: the N>1 code path minus xGMI."""
    X, y, Z = o.synthetic_problem(4001, 6, 120)
    spec = o.make_spec(6, 2)
    d = _capi.KernelDesc(spec)
    ref = o.sgpr_elbo(spec, X, y, Z, 0.01)
    world = 2
    ctxs = [_capi.HipContext(0) for _ in range(world)]
    packed = []
    for r, c in enumerate(ctxs):
        lo, hi = D.shard_bounds(len(X), r, world)
        c.sgpr_set_data(X[lo:hi], y[lo:hi]); c.sgpr_set_inducing(Z); c.sgpr_set_route(D.choose_route(len(X), len(Z)))
        c.sgpr_local_stats(d)
        packed.append(c.sgpr_get_stats())
    total = packed[0] + packed[1]
    for r, c in enumerate(ctxs):
        lo, hi = D.shard_bounds(len(X), r, world)
        m = D.ShardedSGPR(c, X[lo:hi], y[lo:hi], Z, len(X), reducer=lambda p, t=total: t)
        assert abs(m.elbo(d, 0.01) - ref) <= 1e-10 * abs(ref)


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("route", ["phi", "whitened"])
def test_loopback_ranks_equal_stacked_rows(world, route):
    """The N > 1 arithmetic on one GPU: with the loopback communicator (`world` ranks holding the SAME rows, every
    all-reduce multiplies by `world`) ELBO, hyper-parameter gradient and inducing-input gradient must equal a single-rank
    run on the rows stacked `world` times -- this exercises the all-reduce placement, the 1/nranks scaling of the
    replicated <G_uu, dKuu> terms and the replicated tail."""
    X, y, Z = o.synthetic_problem(1500, 4, 40, seed=9)
    spec = o.make_spec(4, 2, lengthscales=[1.1, 0.8, 1.5, 1.0], order_variances=[0.7, 1.2, 0.9])
    d = _capi.KernelDesc(spec)
    ref_ctx = _capi.HipContext(0)
    ref_ctx.sgpr_set_data(np.tile(X, (world, 1)), np.tile(y, (world, 1))); ref_ctx.sgpr_set_inducing(Z); ref_ctx.sgpr_set_route(route)
    e_ref, g_ref, gz_ref = ref_ctx.sgpr_elbo_grad_z(d, 0.05, 40, 4)
    ctx = _capi.HipContext(0)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route(route)
    ctx.comm_init_loopback(world)
    e = ctx.sgpr_elbo(d, 0.05)
    e2, g, gz = ctx.sgpr_elbo_grad_z(d, 0.05, 40, 4)
    assert abs(e - e_ref) <= 1e-11 * abs(e_ref) and abs(e2 - e_ref) <= 1e-11 * abs(e_ref)
    np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-9 * np.abs(g_ref).max())
    np.testing.assert_allclose(gz, gz_ref, rtol=1e-9, atol=1e-9 * np.abs(gz_ref).max())
    ctx.comm_destroy()
    assert abs(ctx.sgpr_elbo(d, 0.05) - e_ref) > 1e-3 * abs(e_ref)       # without the communicator it is a different problem
    ctx.close(); ref_ctx.close()


@pytest.mark.parametrize("world", [2, 5])
def test_term_sharded_sobol_and_row_sharded_predict(world):
    """The collective-free pieces with the HIP kernels: each simulated rank evaluates its block of Sobol terms /
    test rows on its own context; stitched together they equal the single-context answer bit for bit."""
    from oak import distributed as D
    X, y, Z = o.synthetic_problem(900, 5, 32, seed=3)
    spec = o.make_spec(5, 3, lengthscales=[1.1, 0.8, 1.5, 1.0, 0.7], order_variances=[0.7, 1.2, 0.9, 0.4])
    d = _capi.KernelDesc(spec)
    ref = _capi.HipContext(0)
    ref.sgpr_set_data(X, y); ref.sgpr_set_inducing(Z)
    ref.sgpr_elbo(d, 0.05)
    alpha = ref.sgpr_alpha(32)
    subsets = [list(s) for r in range(1, 4) for s in __import__("itertools").combinations(range(5), r)]
    sob_ref = ref.sobol(d, Z, alpha, subsets)
    Xs = np.random.default_rng(0).standard_normal((101, 5))
    m_ref, v_ref = ref.sgpr_predict(d, Xs)
    blocks_s, blocks_p = {}, {}
    ctxs = []
    for rank in range(world):            # pass 1: every rank computes its block; the "gather" just records it
        ctx = _capi.HipContext(0)
        ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_elbo(d, 0.05)
        ctxs.append(ctx)
        lo, hi = D.shard_bounds(len(subsets), rank, world)
        blocks_s[rank] = ctx.sobol(d, Z, alpha, subsets[lo:hi]) if hi > lo else np.empty(0)
        lo, hi = D.shard_bounds(len(Xs), rank, world)
        blocks_p[rank] = np.stack(ctx.sgpr_predict(d, Xs[lo:hi]), axis=1) if hi > lo else np.empty((0, 2))
    for rank in range(world):            # pass 2: the library helpers with a gather that returns all recorded blocks
        sob = D.sharded_sobol(ctxs[rank], d, Z, alpha, subsets, rank, world, gather=lambda local: [blocks_s[r] for r in range(world)])
        mean, var = D.sharded_predict(ctxs[rank], d, Xs, rank, world, gather=lambda local: [blocks_p[r] for r in range(world)])
        assert np.array_equal(sob, sob_ref) and np.array_equal(mean, m_ref) and np.array_equal(var, v_ref)
    for c in ctxs + [ref]:
        c.close()


def test_auto_route_is_decided_on_global_rows_under_a_communicator():
    """Shard size and global size on opposite sides of the auto rule's threshold (N*M = 2^24): the local count alone would
    whiten (and so would one rank of a pair whose shards differ by a row, while its peer would not); the library must
    decide on the communicator-wide row count, here 2 x 100 000 rows x 128 inducing points, where the well-conditioned Kuu
    keeps the phi route -- and the result must equal the single-rank run on the stacked rows."""
    n_local, M, world = 100_000, 128, 2
    assert n_local * M <= (1 << 24) < world * n_local * M
    X, y, Z = o.synthetic_problem(n_local, 12, M, seed=4)
    spec = o.make_spec(12, 2)
    d = _capi.KernelDesc(spec)
    ref = _capi.HipContext(0)
    ref.sgpr_set_data(np.tile(X, (world, 1)), np.tile(y, (world, 1))); ref.sgpr_set_inducing(Z); ref.sgpr_set_route("auto")
    e_ref = ref.sgpr_elbo(d, 0.05)
    assert not ref.sgpr_stats_whitened()
    ctx = _capi.HipContext(0)
    ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route("auto")
    ctx.sgpr_elbo(d, 0.05)
    assert ctx.sgpr_stats_whitened()                 # alone, this shard is a small problem: whitened
    ctx.comm_init_loopback(world)
    e = ctx.sgpr_elbo(d, 0.05)
    assert not ctx.sgpr_stats_whitened()             # under the communicator the global size decides
    assert abs(e - e_ref) <= 1e-11 * abs(e_ref)
    eg, g = ctx.sgpr_elbo_grad(d, 0.05)
    assert not ctx.sgpr_stats_whitened() and abs(eg - e_ref) <= 1e-11 * abs(e_ref)
    ctx.comm_destroy()
    # host-exchange users declare the global size instead
    ctx.sgpr_set_global_rows(world * n_local)
    ctx.sgpr_local_stats(d)
    assert not ctx.sgpr_stats_whitened()
    ctx.sgpr_set_global_rows(0)
    ctx.sgpr_local_stats(d)
    assert ctx.sgpr_stats_whitened()
    ctx.close(); ref.close()


def test_mixed_route_statistics_are_rejected():
    """A sum of one whitened and one raw shard (what two ranks deciding differently would all-reduce) must fail loudly, both
    when it is handed back through set_stats and when the flag passed with it contradicts the vector."""
    X, y, Z = o.synthetic_problem(3000, 5, 64, seed=2)
    spec = o.make_spec(5, 2)
    d = _capi.KernelDesc(spec)
    ctx = _capi.HipContext(0)
    ctx.sgpr_set_data(X[:1500], y[:1500]); ctx.sgpr_set_inducing(Z)
    ctx.sgpr_set_route("phi"); ctx.sgpr_local_stats(d); raw = ctx.sgpr_get_stats()
    ctx.sgpr_set_data(X[1500:], y[1500:])
    ctx.sgpr_set_route("whitened"); ctx.sgpr_local_stats(d); white = ctx.sgpr_get_stats()
    assert raw[-2:].tolist() == [0.0, 1.0] and white[-2:].tolist() == [1.0, 1.0]
    for flag in (False, True):
        with pytest.raises(_capi.OakHipError, match="mix"):
            ctx.sgpr_set_stats(raw + white, flag)
    with pytest.raises(_capi.OakHipError, match="flagged"):
        ctx.sgpr_set_stats(raw + raw, True)
    with pytest.raises(ValueError):
        D.unpack_stats(raw + white, 64)
    ctx.sgpr_set_stats(white + white, True)          # consistent sums pass
    ctx.sgpr_set_stats(raw + raw, False)
    ctx.close()


@pytest.mark.parametrize("n", [1, 5, 8, 13, 1027])
def test_allreduce_slices_and_padding_for_lengths_not_divisible_by_the_world(n):
    """comm_allreduce_dev splits the vector into equal slices padded in a staging buffer (reduce-scatter + all-gather): for
    every length, divisible by the world size or not, a 1-rank RCCL communicator must return the vector unchanged and the
    loopback communicator (world 8) 8 x the vector, with nothing written past its end."""
    ctx = _capi.HipContext(0)
    v = np.arange(1.0, n + 1)
    if n == 13:                       # one real (1-rank) RCCL communicator is enough: its slicing is the identity case
        _rccl_init(ctx, _capi.HipContext.comm_unique_id(), 1, 0)
        np.testing.assert_array_equal(ctx.comm_allreduce_host(v.copy()), v)
    for world in (2, 3, 8):
        ctx.comm_init_loopback(world)
        np.testing.assert_array_equal(ctx.comm_allreduce_host(v.copy()), world * v)
    ctx.close()


def _run_bench(extra, env_extra=None):
    import json, os, socket, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1", OAK_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    cmd = [sys.executable, str(root / "bench.py"), "--config", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


@pytest.mark.parametrize("route", ["phi", "auto"])
def test_bench_two_ranks_on_one_gpu_through_the_host_exchange(route):
    """bench.py --gpus 2 end to end on ONE GPU (both ranks on device 0, statistics all-reduced over gloo): the launcher, the
    row sharding, the route carried by the summed statistics and the aggregation of the JSON line.  The loss of the two-rank
    job must equal the one-rank loss (1e-10), rows_per_gpu must be half, and the line must not be flagged degraded (the host
    exchange was ASKED for here; a fallback to it is what `degraded` marks)."""
    rc1, one, err1 = _run_bench(["--gpus", "1", "--route", route])
    assert rc1 == 0 and one is not None, err1[-2000:]
    rc2, two, err2 = _run_bench(["--gpus", "2", "--exchange", "host", "--route", route])
    assert rc2 == 0 and two is not None, err2[-2000:]
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["degraded"] is False
    assert two["config"]["rows_per_gpu"] * 2 == one["config"]["rows_per_gpu"] == one["config"]["N"]
    assert two["config"]["exchange"] == "host" and two["steps"] == 3 and two["warmup"] == 1
    # two shards sum Phi in a different order; the phi route amplifies that by cond(Kuu) (measured 1.4e-10 here), auto whitens
    assert abs(two["loss"] - one["loss"]) <= (2e-9 if route == "phi" else 1e-10) * abs(one["loss"]), (two["loss"], one["loss"])
    assert two["value"] > 0 and two["unit"] == "steps/s" and two["scaling"] == "strong"


def test_bench_fails_loudly_when_rccl_cannot_be_used():
    """Two ranks on ONE device with the default RCCL exchange: the communicator cannot be created (duplicate device).  Without
    --allow-host-exchange the run must exit non-zero and print no result line; with it, a line flagged `degraded` and still a
    non-zero status."""
    rc, line, err = _run_bench(["--gpus", "2"], {"OAK_BENCH_RCCL_TIMEOUT": "60"})
    assert rc != 0 and line is None, (rc, line, err[-1500:])
    rc, line, err = _run_bench(["--gpus", "2", "--allow-host-exchange"], {"OAK_BENCH_RCCL_TIMEOUT": "60"})
    assert rc != 0 and line is not None and line["degraded"] is True and "host" in line["config"]["exchange"], (rc, line, err[-1500:])


def test_bench_stdout_is_exactly_one_json_line():
    """The driver parses the bench's stdout: ONE line, the JSON -- with the bounded model fit and the CPU baseline enabled (the
    model classes print like the reference's, and the oracle build may chat), and under the two-rank launcher."""
    import json, os, socket, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for extra in ([], ["--gpus", "2", "--exchange", "host"]):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1", OAK_BENCH_DEVICE="0")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, str(root / "bench.py"), "--config", "tiny", "--steps", "2", "--warmup", "1", "--fit-maxiter", "2",
               "--cpu-sample-rows", "4096"] + extra
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = p.stdout.splitlines()
        assert len(lines) == 1, lines[:6]
        d = json.loads(lines[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline"):
            assert key in d, key
        if not extra:
            assert "cpu_baseline" in d and "fit" in d and "error" not in d["fit"], d.get("fit")


def _two_party_callbacks():
    """Host all-reduce callbacks for two contexts driven in lock step from two host threads (a two-party sum)."""
    import threading
    slots, bar = [None, None], threading.Barrier(2)

    def make_cb(r):
        def cb(a):
            slots[r] = np.array(a, copy=True)
            bar.wait()
            out = slots[0] + slots[1]
            bar.wait()
            return out
        return cb
    return make_cb(0), make_cb(1)


@pytest.mark.parametrize("path", ["gram", "terms"])
def test_collective_sobol_shards_pair_rows_or_terms_over_the_ranks(path):
    """oak_sobol_collective under a 2-rank host-exchange communicator: each rank builds the Gram of products over ITS half of
    the index-pair rows (or evaluates its block of terms), the partial results are summed through the communicator, and both
    ranks end with every term -- equal to the one-context answer, identical on both ranks."""
    import threading
    import cases
    rng = np.random.default_rng(21)
    spec = cases.random_spec(rng, 8, 4, kinds=("gaussian", "binary", "categorical", "gauss2"))
    d = _capi.KernelDesc(spec)
    n = 130
    Z = cases.random_inputs(rng, spec, n)
    alpha = rng.standard_normal(n)
    subsets = o.list_representation(8, 4)[1:]
    ref_ctx = _capi.HipContext(0)
    ref_ctx.sobol_set_path(path)
    ref = ref_ctx.sobol(d, Z, alpha, subsets)
    oracle = np.array(o.compute_sobol_oak(spec, Z, alpha.reshape(-1, 1))[1])
    np.testing.assert_allclose(ref, oracle, rtol=1e-9, atol=1e-11 * np.abs(oracle).max())
    cb0, cb1 = _two_party_callbacks()
    ranks = [_capi.HipContext(0), _capi.HipContext(0)]
    ranks[0].comm_init_host(2, 0, cb0); ranks[1].comm_init_host(2, 1, cb1)
    res, info = {}, {}

    def run(r):
        ranks[r].sobol_set_path(path)
        res[r] = ranks[r].sobol(d, Z, alpha, subsets, collective=True)
        info[r] = ranks[r].sobol_last_info()
    th = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
    [t.start() for t in th]; [t.join(120) for t in th]
    assert not any(t.is_alive() for t in th) and set(res) == {0, 1}
    assert np.array_equal(res[0], res[1])
    np.testing.assert_allclose(res[0], ref, rtol=1e-11, atol=1e-13 * np.abs(ref).max())
    assert info[0]["path"] == path
    # a non-collective call on an attached context still evaluates everything on its own
    np.testing.assert_allclose(ranks[0].sobol(d, Z, alpha, subsets), ref, rtol=1e-12, atol=1e-14 * np.abs(ref).max())
    for c in ranks + [ref_ctx]:
        c.close()


def test_allgatherv_refuses_blocks_for_ranks_it_cannot_reach():
    """A context with no communicator (or a smaller one) asked to gather blocks of several ranks must fail, not hand back a
    buffer with zeros where the other ranks' blocks belong (round-3 advisor finding)."""
    ctx = _capi.HipContext(0)
    np.testing.assert_array_equal(ctx.comm_allgatherv(np.arange(3.0), [3]), np.arange(3.0))
    with pytest.raises(_capi.OakHipError) as ei:
        ctx.comm_allgatherv(np.arange(3.0), [3, 4])
    assert ei.value.status == _capi.OAK_E_STATE
    ctx.comm_init_loopback(2)
    np.testing.assert_array_equal(ctx.comm_allgatherv(np.arange(3.0), [3, 3]), np.tile(np.arange(3.0), 2))
    with pytest.raises(ValueError):
        ctx.comm_allgatherv(np.arange(3.0), [3, 4])           # loopback ranks hold the same block by construction
    with pytest.raises(_capi.OakHipError):
        ctx.comm_allgatherv(np.arange(3.0), [3, 3, 3])
    ctx.close()


def test_several_output_columns_under_a_two_rank_communicator():
    """Row shards + extra target columns: each rank forms [Kuf y_p | y_p^T y_p] of its rows for every output, the library sums
    them next to the packed statistics, and bound / gradient of the P-column model equal the one-context evaluation on both
    ranks (bit-identical between the ranks)."""
    import threading
    X, y, Z = o.synthetic_problem(5001, 5, 96, seed=5)
    rng = np.random.default_rng(2)
    Y = np.concatenate([y, rng.standard_normal((len(X), 3)) + y], axis=1)
    spec = o.make_spec(5, 2, lengthscales=[0.9, 1.1, 1.3, 0.8, 1.0])
    d = _capi.KernelDesc(spec)
    ref = _capi.HipContext(0)
    ref.sgpr_set_data(X, Y[:, 0]); ref.sgpr_set_extra_targets(Y[:, 1:]); ref.sgpr_set_inducing(Z); ref.sgpr_set_route("phi")
    e_ref, g_ref = ref.sgpr_elbo_grad(d, 0.05)
    assert abs(e_ref - o.sgpr_elbo(spec, X, Y, Z, 0.05)) <= 1e-10 * abs(e_ref)
    cb0, cb1 = _two_party_callbacks()
    cuts = [(0, 2000), (2000, len(X))]
    ranks = [_capi.HipContext(0), _capi.HipContext(0)]
    for r, (lo, hi) in enumerate(cuts):
        ranks[r].sgpr_set_data(X[lo:hi], Y[lo:hi, 0]); ranks[r].sgpr_set_extra_targets(Y[lo:hi, 1:])
        ranks[r].sgpr_set_inducing(Z); ranks[r].sgpr_set_route("phi"); ranks[r].sgpr_set_global_rows(len(X))
    ranks[0].comm_init_host(2, 0, cb0); ranks[1].comm_init_host(2, 1, cb1)
    res = {}

    def run(r):
        res[r] = ranks[r].sgpr_elbo_grad(d, 0.05)
        ranks[r].sgpr_select_output(3)
        res[("alpha", r)] = ranks[r].sgpr_alpha(96)
    th = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
    [t.start() for t in th]; [t.join(120) for t in th]
    assert not any(t.is_alive() for t in th) and 0 in res and 1 in res
    for r in (0, 1):
        assert abs(res[r][0] - e_ref) <= 1e-12 * abs(e_ref)
        np.testing.assert_allclose(res[r][1], g_ref, rtol=1e-9, atol=1e-9 * np.abs(g_ref).max())
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[("alpha", 0)], res[("alpha", 1)])
    ref.sgpr_select_output(3)
    np.testing.assert_allclose(res[("alpha", 0)], ref.sgpr_alpha(96), rtol=1e-8, atol=1e-10)
    for c in ranks + [ref]:
        c.close()


@pytest.mark.parametrize("route", ["phi", "whitened"])
def test_three_call_sequence_with_extra_outputs_under_a_loopback_communicator(route):
    """oak_hip.h documents oak_sgpr_elbo as local_stats -> allreduce_stats -> tail.  With extra target columns the three calls must
    exchange what the fused entry point does: [Kuf y_p | y_p^T y_p] of the extra outputs is summed next to the packed statistics
    (it used to be summed inside oak_sgpr_elbo only, so the hand-written sequence paired summed Phi / psi with per-shard psi_p)."""
    X, y, Z = o.synthetic_problem(3000, 4, 64, seed=13)
    rng = np.random.default_rng(4)
    Y = np.concatenate([y, rng.standard_normal((len(X), 2)) + 0.5 * y], axis=1)
    spec = o.make_spec(4, 2, lengthscales=[0.9, 1.1, 1.3, 0.8])
    d = _capi.KernelDesc(spec)
    world = 2
    ctx = _capi.HipContext(0)
    try:
        ctx.sgpr_set_data(X, Y[:, 0]); ctx.sgpr_set_extra_targets(Y[:, 1:]); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route(route)
        ctx.comm_init_loopback(world)
        e_fused = ctx.sgpr_elbo(d, 0.07)
        ctx.sgpr_local_stats(d)
        ctx.comm_allreduce_stats()
        e_seq, _ = ctx.sgpr_tail(d, 0.07)
        Xs, Ys = np.tile(X, (world, 1)), np.tile(Y, (world, 1))
        e_ref = o.sgpr_elbo(spec, Xs, Ys, Z, 0.07)
        assert abs(e_fused - e_ref) <= 1e-10 * abs(e_ref)
        assert abs(e_seq - e_ref) <= 1e-10 * abs(e_ref)
        assert abs(e_seq - e_fused) <= 1e-13 * abs(e_ref)
    finally:
        ctx.close()


@pytest.mark.parametrize("route", ["phi", "auto", "whitened"])
def test_partitioned_forward_pass_under_a_loopback_communicator(route, monkeypatch):
    """The spatially partitioned forward pass (CU-masked streams, sgpr.hip) with a communicator attached: the packed statistics are
    all-reduced after the partition has been left, the auto route's conditioning decision is a scalar collective on the (masked)
    side stream.  Two loopback ranks holding the same rows must equal one rank on the stacked rows, with the partition forced on."""
    monkeypatch.setenv("OAK_PARTITION", "1")
    X, y, Z = o.synthetic_problem(24000, 6, 256, seed=17)
    spec = o.make_spec(6, 2, lengthscales=[1.1, 0.8, 1.5, 1.0, 0.9, 1.2], order_variances=[0.7, 1.2, 0.9])
    d = _capi.KernelDesc(spec)
    world = 2
    ref_ctx = _capi.HipContext(0)
    ctx = _capi.HipContext(0)
    try:
        ref_ctx.sgpr_set_data(np.tile(X, (world, 1)), np.tile(y, (world, 1))); ref_ctx.sgpr_set_inducing(Z); ref_ctx.sgpr_set_route(route)
        e_ref, g_ref = ref_ctx.sgpr_elbo_grad(d, 0.05)
        ctx.sgpr_set_data(X, y); ctx.sgpr_set_inducing(Z); ctx.sgpr_set_route(route)
        ctx.comm_init_loopback(world)
        e = ctx.sgpr_elbo(d, 0.05)
        e2, g = ctx.sgpr_elbo_grad(d, 0.05)
        assert abs(e - e_ref) <= 1e-11 * abs(e_ref) and e2 == e
        np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-9 * np.abs(g_ref).max())
        assert ref_ctx.sgpr_stats_whitened() == ctx.sgpr_stats_whitened()
    finally:
        ctx.close(); ref_ctx.close()

"""Host-side logic of the mirror package that needs no device: transforms, parameter containers, kernel description
packing, constructor semantics and error conventions of the reference interface."""
import itertools

import os

import numpy as np
import pytest

import cases
from oak import _capi
from oak import gpflow_lite as gpflow
from oak.input_measures import EmpiricalMeasure, GaussianMeasure, MOGMeasure, UniformMeasure
from oak.model_utils import _calculate_features, estimate_one_dim_gmm
from oak.normalising_flow import Normalizer
from oak.oak_kernel import OAKKernel, KernelComponenent, _categorical_chain, bounded_param, get_list_representation, kernel_to_spec
from oak.ortho_binary_kernel import OrthogonalBinary
from oak.ortho_categorical_kernel import OrthogonalCategorical
from oak.ortho_rbf_kernel import OrthogonalRBFKernel
from oracle import oak_oracle as o


@pytest.mark.parametrize("tr", [gpflow.Softplus(), gpflow.Softplus(1e-6), gpflow.Sigmoid(1e-3, 1e3), gpflow.Sigmoid(1e-6, 2)])
def test_transform_roundtrip_and_derivative(tr):
    u = np.linspace(-6, 6, 25)
    x = tr.forward(u)
    np.testing.assert_allclose(tr.inverse(x), u, rtol=1e-9, atol=1e-9)
    h = 1e-6
    np.testing.assert_allclose(tr.dforward(u), (tr.forward(u + h) - tr.forward(u - h)) / (2 * h), rtol=1e-6, atol=1e-9)


def test_transforms_match_oracle_restatement():
    u = np.linspace(-4, 4, 9)
    np.testing.assert_allclose(gpflow.Softplus().forward(u), o.softplus(u))
    np.testing.assert_allclose(gpflow.Sigmoid(1e-3, 1e3).forward(u), o.sigmoid_bounded(u, 1e-3, 1e3))


def test_parameter_assign_numpy_and_prior():
    p = gpflow.Parameter(1.0, transform=gpflow.positive())
    p.assign(0.3)
    assert abs(float(p.numpy()) - 0.3) < 1e-15
    p.prior = gpflow.Gamma(1.0, 0.2)
    np.testing.assert_allclose(p.log_prior_density(), o.gamma_log_prob(0.3), rtol=1e-14)
    b = bounded_param(1e-3, 1e3, 1)
    assert abs(float(b.numpy()) - 1.0) < 1e-12
    with pytest.raises(ValueError):
        gpflow.Parameter(-1.0, transform=gpflow.positive())


def test_oak_kernel_constructor_semantics():
    k = OAKKernel([gpflow.RBF] * 3, num_dims=3, max_interaction_depth=2, constrain_orthogonal=True, lengthscale_bounds=[1e-3, 1e3])
    assert len(k.kernels) == 3 and len(k.variances) == 3
    assert all(isinstance(s, OrthogonalRBFKernel) and isinstance(s.measure, GaussianMeasure) for s in k.kernels)
    # shared variances: base variances are constants, not trainable parameters (oak_kernel.py:163-166)
    assert not isinstance(k.kernels[0].base_kernel.variance, gpflow.Parameter)
    assert isinstance(k.kernels[0].base_kernel.lengthscales.transform, gpflow.Sigmoid)
    names = [n for n, _ in k.named_parameters()]
    assert sum("lengthscales" in n for n in names) == 3 and sum(n.startswith("variances") for n in names) == 3
    k2 = OAKKernel([gpflow.RBF] * 2, num_dims=2, max_interaction_depth=2, constrain_orthogonal=True, share_var_across_orders=False)
    assert len(k2.variances) == 1 and isinstance(k2.kernels[0].base_kernel.variance, gpflow.Parameter)
    spec = kernel_to_spec(k)
    assert spec["max_interaction_depth"] == 2 and [d["measure"][0] for d in spec["dims"]] == ["gaussian"] * 3
    desc = _capi.KernelDesc(spec)
    assert desc.D == 3 and desc.R == 2 and list(desc.active_col) == [0, 1, 2]


def test_oak_kernel_mixed_types_and_errors():
    p = np.array([0.2, 0.3, 0.5]).reshape(-1, 1)
    k = OAKKernel([gpflow.RBF, None, None], num_dims=3, max_interaction_depth=2, constrain_orthogonal=True,
                  p0=[None, 0.4, None], p=[None, None, p])
    assert isinstance(k.kernels[1], OrthogonalBinary) and isinstance(k.kernels[2], OrthogonalCategorical)
    spec = kernel_to_spec(k)
    assert [d["type"] for d in spec["dims"]] == ["rbf", "binary", "categorical"]
    with pytest.raises(ValueError):   # both empirical and GMM measure on one input (oak_kernel.py:132-138)
        OAKKernel([gpflow.RBF], num_dims=1, max_interaction_depth=1, constrain_orthogonal=True,
                  empirical_locations=[np.zeros((2, 1))], empirical_weights=[np.full((2, 1), .5)],
                  gmm_measures=[MOGMeasure(np.zeros(1), np.ones(1), np.ones(1))])
    with pytest.raises(AssertionError):   # duplicate active dims (oak_kernel.py:80-82)
        OAKKernel([gpflow.RBF] * 2, num_dims=2, max_interaction_depth=1, active_dims=[[0], [0]])
    with pytest.raises(AssertionError):   # empirical locations without the orthogonal constraint (:192-197)
        OAKKernel([gpflow.RBF], num_dims=1, max_interaction_depth=1, empirical_locations=[np.zeros((2, 1))])
    with pytest.raises(NotImplementedError):
        OrthogonalRBFKernel(OrthogonalBinary(), GaussianMeasure(0, 1))
    with pytest.raises(NotImplementedError):
        OrthogonalRBFKernel(gpflow.RBF(), object())


def test_measures_validate():
    with pytest.raises(AssertionError):
        EmpiricalMeasure(np.zeros((3, 1)), np.ones((3, 1)))
    with pytest.raises(AssertionError):
        MOGMeasure(np.zeros(2), np.ones(2), np.array([0.5, 0.6]))
    with pytest.raises(ValueError):
        MOGMeasure(np.zeros((2, 1)), np.ones(2), np.array([0.5, 0.5]))
    m = MOGMeasure(np.array([3, 2], dtype=int), np.array([3, 10], dtype=int), np.array([0.6, 0.4]))
    assert m.means.dtype == float and m.variances.dtype == float
    assert EmpiricalMeasure(np.zeros((4, 1))).weights.shape == (4, 1)
    assert UniformMeasure(0, 1).as_tuple() == ("uniform", 0.0, 1.0)


def test_list_representation_matches_reference_order():
    k = OAKKernel([gpflow.RBF] * 4, num_dims=4, max_interaction_depth=3, constrain_orthogonal=True)
    sel, comps = get_list_representation(k, num_dims=4)
    expect = [[]] + [list(c) for r in (1, 2, 3) for c in itertools.combinations(range(4), r)]
    assert sel == expect == o.list_representation(4, 3)
    assert all(isinstance(c, KernelComponenent) for c in comps) and len(comps) == len(sel)
    k2 = OAKKernel([gpflow.RBF] * 2, num_dims=2, max_interaction_depth=2, constrain_orthogonal=True)
    assert get_list_representation(k2, num_dims=2)[0] == [[], [0], [1], [0, 1]]


def test_kernel_desc_packs_every_measure():
    spec, *_ = cases.case_B()
    d = _capi.KernelDesc(spec)
    assert list(d.dim_type) == [0, 0, 0, 1, 2, 0]
    assert list(d.measure[[0, 1, 2, 5]]) == [_capi.MEAS_GAUSSIAN, _capi.MEAS_UNIFORM, _capi.MEAS_MOG, _capi.MEAS_EMPIRICAL]
    off, C = d.cat_blocks[4]
    B = d.meas_data[off:off + C * C].reshape(C, C)
    np.testing.assert_allclose(B, o.categorical_table({**spec["dims"][4], "variance": 1.0}), rtol=1e-14)
    with pytest.raises(ValueError):
        _capi.KernelDesc(dict(dims=spec["dims"], order_variances=[1.0], max_interaction_depth=3, share_var_across_orders=True))
    with pytest.raises(ValueError):
        _capi.KernelDesc(dict(dims=spec["dims"][:1], order_variances=[1.0] * 66, max_interaction_depth=65, share_var_across_orders=True))
    # any depth up to 64 can be described (e_r vanishes beyond the number of sub-kernels, so the kernels run at min(R, D);
    # the fused paths take an EFFECTIVE depth up to 32 -- the reference's examples go to 32 (pumadyn32nm) -- the explicit Gram goes deeper)
    assert _capi.KernelDesc(dict(dims=spec["dims"][:1], order_variances=[1.0] * 18, max_interaction_depth=17,
                                 share_var_across_orders=True)).R == 17
    assert _capi.KernelDesc(dict(dims=spec["dims"][:1], order_variances=[1.0] * 14, max_interaction_depth=13,
                                 share_var_across_orders=True)).R == 13


def test_categorical_chain_rule_matches_finite_differences():
    rng = np.random.default_rng(0)
    C = 4
    W, kappa = rng.uniform(size=(C, 2)), rng.uniform(0.5, 1.5, C)
    p = rng.uniform(0.5, 1.5, C); p = (p / p.sum()).reshape(-1, 1)
    G = rng.standard_normal((C, C))
    f = lambda W_, k_: float(np.sum(G * _capi.categorical_table_unit(W_, k_, p)[0]))
    gW, gk = _categorical_chain(W, kappa, p, G)
    h = 1e-6
    for idx in np.ndindex(W.shape):
        Wp, Wm = W.copy(), W.copy(); Wp[idx] += h; Wm[idx] -= h
        np.testing.assert_allclose(gW[idx], (f(Wp, kappa) - f(Wm, kappa)) / (2 * h), rtol=1e-5, atol=1e-7)
    for i in range(C):
        kp, km = kappa.copy(), kappa.copy(); kp[i] += h; km[i] -= h
        np.testing.assert_allclose(gk[i], (f(W, kp) - f(W, km)) / (2 * h), rtol=1e-5, atol=1e-7)


def test_calculate_features_and_errors():
    rng = np.random.default_rng(1)
    X = np.stack([rng.integers(0, 2, 30), rng.integers(0, 3, 30), rng.standard_normal(30)], axis=1).astype(float)
    cont, binary, cat, p0, p = _calculate_features(X, categorical_feature=[1], binary_feature=[0])
    assert (cont, binary, cat) == ([2], [0], [1])
    np.testing.assert_allclose(p0[0], 1 - X[:, 0].mean())
    np.testing.assert_allclose(p[1].sum(), 1.0)
    assert _calculate_features(X, None, None)[3] is None
    with pytest.raises(ValueError):
        _calculate_features(X, categorical_feature=[0], binary_feature=[0])


def test_gmm_fit():
    """tests/test_orthogonality.py:168-171."""
    measure = estimate_one_dim_gmm(K=2, X=np.array([1.0, 1, 1, 10, 10, 10]))
    np.testing.assert_almost_equal(np.sort(measure.means), np.array([1.0, 10.0]))


def test_normalising_flow_transforms():
    """The elementwise flow transforms (NumPy): inverse o forward = id, log-det against a central difference, and the
    objective oracle against its definition through them (tests/test_normalising_flow.py fits the flow: GPU test)."""
    from oracle import flow_oracle
    rng = np.random.default_rng(44)
    x = rng.normal(2, 0.5, size=(100, 1))
    n = Normalizer(x, log=False)
    n.skewness.assign(0.3); n.tailweight.assign(1.4)
    y = np.asarray(n.bijector(x))
    np.testing.assert_allclose(n.bijector.inverse(y), x, rtol=1e-9)
    nl = Normalizer(np.exp(x[:, 0]), log=True)
    nl.skewness.assign(-0.2); nl.tailweight.assign(0.8)
    h = 1e-6
    xs = np.exp(x[:5, 0])
    np.testing.assert_allclose(nl.bijector.forward_log_det_jacobian(xs),
                               np.log((np.asarray(nl.bijector(xs + h)) - np.asarray(nl.bijector(xs - h))) / (2 * h)), rtol=1e-6)
    for m in (n, nl):      # oracle objective == 1/2 E[y^2] - E[log det] assembled from the product's transforms
        xv = m.x
        direct = 0.5 * np.mean(np.square(np.asarray(m.bijector(xv)))) - np.mean(m.bijector.forward_log_det_jacobian(xv))
        orc = flow_oracle.kl_objective(m._g, m.bijector.log, float(m.scale.numpy()), float(m.shift.numpy()),
                                       float(m.skewness.numpy()), float(m.tailweight.numpy()))
        np.testing.assert_allclose(orc, direct, rtol=1e-12)


def test_grouped_active_dims_are_described_as_groups_not_truncated():
    """OAKKernel(active_dims=[[0, 1], [2]]) builds as in the reference (oak_kernel.py:74-82); an unconstrained RBF over two columns
    is described with BOTH columns (evaluated by the explicit Gram entry points, tests/test_gpu_gram.py) -- never as a 1-D kernel
    on the first one -- while a constrained kernel over two columns has no meaning (the reference asserts [N, 1] inputs,
    ortho_rbf_kernel.py:50,83) and raises."""
    from oak import gpflow_lite as gpflow
    from oak.oak_kernel import OAKKernel, kernel_to_spec
    k = OAKKernel([gpflow.kernels.RBF, gpflow.kernels.RBF], num_dims=3, max_interaction_depth=2, active_dims=[[0, 1], [2]],
                  constrain_orthogonal=False)
    spec = kernel_to_spec(k)
    assert spec["dims"][0]["active_dims"] == [0, 1] and spec["dims"][0]["measure"] is None and spec["dims"][1].get("active_dims") is None
    d = _capi.KernelDesc(spec)
    assert d.grouped and d.min_cols == 3 and d.extra_col_off.tolist() == [0, 1, 1] and d.extra_cols.tolist() == [1]
    kc = OAKKernel([gpflow.kernels.RBF, gpflow.kernels.RBF], num_dims=3, max_interaction_depth=2, active_dims=[[0, 1], [2]],
                   constrain_orthogonal=True)
    with pytest.raises(NotImplementedError, match="active columns"):
        kernel_to_spec(kc)
    ok = OAKKernel([gpflow.kernels.RBF, gpflow.kernels.RBF], num_dims=3, max_interaction_depth=2, active_dims=[[1], [2]],
                   constrain_orthogonal=True)
    spec = kernel_to_spec(ok)
    assert [d["active_dim"] for d in spec["dims"]] == [1, 2] and spec["base_var_grad"] is False
    # empirical-measure dims keep a trainable base variance under share_var_across_orders=True: the description says so
    emp = OAKKernel([gpflow.kernels.RBF, gpflow.kernels.RBF], num_dims=2, max_interaction_depth=2, constrain_orthogonal=True,
                    empirical_locations=[None, np.linspace(-1, 1, 5).reshape(-1, 1)],
                    empirical_weights=[None, np.full((5, 1), 0.2)])
    assert kernel_to_spec(emp)["base_var_grad"] is True


_MULTI_OUTPUT = r"""
import sys
import numpy as np
import fake_hip
fake_hip.install()                                # the oracle answers behind the binding's interface (fresh process)
from oracle import oak_oracle as o
from oak import gpflow_lite as gpflow
from oak.oak_kernel import OAKKernel, kernel_to_spec
rng = np.random.default_rng(2)
X, Z = rng.standard_normal((60, 2)), rng.standard_normal((6, 2))
Y = rng.standard_normal((60, 3))
k = OAKKernel([gpflow.kernels.RBF] * 2, num_dims=2, max_interaction_depth=2, constrain_orthogonal=True)
m = gpflow.models.SGPR((X, Y), k, Z, noise_variance=0.3)
spec = kernel_to_spec(k)
np.testing.assert_allclose(m.elbo(), o.sgpr_elbo(spec, X, Y, Z, 0.3), rtol=1e-12)
singles = [gpflow.models.SGPR((X, Y[:, p:p + 1]), k, Z, noise_variance=0.3) for p in range(3)]
np.testing.assert_allclose(m.elbo(), sum(s.elbo() for s in singles), rtol=1e-12)
obj, g, _ = m._objective_and_constrained_grad()
gs = sum(s._objective_and_constrained_grad()[1] for s in singles)
np.testing.assert_allclose(g, gs, rtol=1e-9, atol=1e-9)
mean, var = m.predict_f(X[:7])
mo, vo = o.sgpr_predict_f(spec, X, Y, Z, 0.3, X[:7])
assert mean.numpy().shape == (7, 3) and var.numpy().shape == (7, 3)
np.testing.assert_allclose(mean.numpy(), mo, rtol=1e-9, atol=1e-12)
np.testing.assert_allclose(var.numpy(), vo, rtol=1e-9, atol=1e-12)
assert m.alpha().numpy().shape == (6, 3)
np.testing.assert_allclose(m.elbo(), o.sgpr_elbo(spec, X, Y, Z, 0.3), rtol=1e-12)      # the column walk restarts cleanly
print("multi-output ok")
"""


def test_multi_output_sgpr_is_the_sum_over_columns():
    """Y with P columns (GPflow's independent outputs, shared kernel and noise): the bound is GPflow's N x P formula
    (oracle.sgpr_elbo), the gradient the sum of the single-output gradients, the mean one column per output.  Host logic only:
    the oracle-backed context stands in for the device (a fresh process, it replaces the binding module-wide)."""
    import subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(root / "orthogonal-additive-gaussian-processes_amd"), str(root), str(root / "tests")]))
    r = subprocess.run([sys.executable, "-c", _MULTI_OUTPUT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "multi-output ok" in r.stdout, r.stdout + r.stderr


def test_calculate_features_against_the_reference_executed_fixture():
    """tests/golden/reference_host_logic.npz holds what the reference's OWN ``_calculate_features`` (oak/model_utils.py:703-750,
    executed from /root/reference in the build container by tests/golden/make_reference_golden.py) returned for a 240 x 6 matrix
    with two binary and one categorical column: the mirror must return the same index sets, p0 and p -- exactly."""
    import contextlib, io
    from pathlib import Path
    from oak.model_utils import _calculate_features
    fx = np.load(Path(__file__).resolve().parent / "golden" / "reference_host_logic.npz")
    assert str(fx["label"]).startswith("reference-executed")
    X = fx["X"]
    with contextlib.redirect_stdout(io.StringIO()):
        cont, binary, cat, p0, p = _calculate_features(X, categorical_feature=[3], binary_feature=[1, 4])
        cont_all, b0, c0, p0_none, p_none = _calculate_features(X[:, [0, 2, 5]], None, None)
    assert list(cont) == fx["cf_continuous"].tolist() and list(binary) == fx["cf_binary"].tolist() and list(cat) == fx["cf_categorical"].tolist()
    assert list(cont_all) == fx["cf_all_continuous"].tolist() and b0 == [] and c0 == [] and p0_none is None and p_none is None
    ref_p0 = fx["cf_p0"]
    assert len(p0) == len(ref_p0) == 6
    for j in range(6):
        if np.isnan(ref_p0[j]):
            assert p0[j] is None
        else:
            assert float(p0[j]) == float(ref_p0[j])
    assert all(v is None for j, v in enumerate(p) if j != 3)
    np.testing.assert_array_equal(np.asarray(p[3], dtype=np.float64), fx["cf_p_col3"])


def test_public_signatures_match_the_introspected_reference():
    """tests/golden/reference_signatures.json = ``inspect.signature`` of the reference's callables on the path, read from the
    imported reference modules in the build container (tests/golden/make_reference_golden.py).  Every mirror must take the same
    parameters, in the same order, with the same plain defaults; it may add keyword parameters with defaults AFTER the reference's
    (extensions), nothing else."""
    import inspect, json
    from pathlib import Path
    from oak import input_measures as im, model_utils as mu, oak_kernel as ok, utils as ut
    from oak.ortho_binary_kernel import OrthogonalBinary
    from oak.ortho_categorical_kernel import OrthogonalCategorical
    from oak.ortho_rbf_kernel import OrthogonalRBFKernel
    ref = json.loads((Path(__file__).resolve().parent / "golden" / "reference_signatures.json").read_text())["signatures"]
    mine = {
        "oak_model.__init__": mu.oak_model.__init__, "oak_model.fit": mu.oak_model.fit, "oak_model.optimise": mu.oak_model.optimise,
        "oak_model.predict": mu.oak_model.predict, "oak_model.get_loglik": mu.oak_model.get_loglik, "oak_model.get_sobol": mu.oak_model.get_sobol,
        "oak_model.plot": mu.oak_model.plot, "create_model_oak": mu.create_model_oak, "get_kmeans_centers": mu.get_kmeans_centers,
        "save_model": mu.save_model, "load_model": mu.load_model, "_calculate_features": mu._calculate_features,
        "OAKKernel.__init__": ok.OAKKernel.__init__, "OAKKernel.compute_additive_terms": ok.OAKKernel.compute_additive_terms,
        "OAKKernel.K": ok.OAKKernel.K, "OAKKernel.K_diag": ok.OAKKernel.K_diag, "KernelComponenent.__init__": ok.KernelComponenent.__init__,
        "get_list_representation": ok.get_list_representation, "bounded_param": ok.bounded_param,
        "OrthogonalRBFKernel.__init__": OrthogonalRBFKernel.__init__, "OrthogonalBinary.__init__": OrthogonalBinary.__init__,
        "OrthogonalCategorical.__init__": OrthogonalCategorical.__init__, "UniformMeasure.__init__": im.UniformMeasure.__init__,
        "GaussianMeasure.__init__": im.GaussianMeasure.__init__, "EmpiricalMeasure.__init__": im.EmpiricalMeasure.__init__,
        "MOGMeasure.__init__": im.MOGMeasure.__init__, "compute_sobol_oak": ut.compute_sobol_oak,
        "get_prediction_component": ut.get_prediction_component, "get_model_sufficient_statistics": ut.get_model_sufficient_statistics,
        "compute_L": ut.compute_L, "compute_L_binary_kernel": ut.compute_L_binary_kernel,
        "compute_L_categorical_kernel": ut.compute_L_categorical_kernel, "f1": ut.f1,
        "initialize_kmeans_with_binary": ut.initialize_kmeans_with_binary,
        "initialize_kmeans_with_categorical": ut.initialize_kmeans_with_categorical,
    }
    checked, problems = 0, []
    for name, rsig in ref.items():
        if isinstance(rsig, str) or name not in mine:          # a tf.function wrapper that became a placeholder: nothing to compare
            continue
        params = list(inspect.signature(mine[name]).parameters.values())
        for k, (pname, kind, default) in enumerate(rsig):
            if k >= len(params):
                problems.append(f"{name}: missing parameter {pname}")
                break
            p = params[k]
            if p.name != pname:
                problems.append(f"{name}: parameter {k} is {p.name!r}, the reference's is {pname!r}")
                break
            if default == "<required>":
                if p.default is not inspect.Parameter.empty:
                    problems.append(f"{name}.{pname}: has default {p.default!r}, required in the reference")
            elif not (isinstance(default, str) and default.startswith("<")):
                mine_default = list(p.default) if isinstance(p.default, (list, tuple)) else p.default
                if mine_default != default or (p.default is inspect.Parameter.empty):
                    problems.append(f"{name}.{pname}: default {p.default!r}, the reference's is {default!r}")
        for p in params[len(rsig):]:
            if p.default is inspect.Parameter.empty and p.kind not in (p.VAR_KEYWORD, p.VAR_POSITIONAL):
                problems.append(f"{name}: extra parameter {p.name} without a default")
        checked += 1
    assert checked >= 30 and not problems, "\\n".join(problems)


def test_members_of_the_introspected_reference_exist_in_the_mirror():
    """Same fixture: every method the reference's classes define and every function / class its modules define exists under the same
    name in the mirror -- except the names listed here, each out of scope for a stated reason."""
    import importlib, json
    from pathlib import Path
    fx = json.loads((Path(__file__).resolve().parent / "golden" / "reference_signatures.json").read_text())
    out_of_scope = {
        "oak.utils": {"compute_sobol", "extract_active_dims", "grammer_to_kernel", "model_to_kernel_list"},   # the legacy (pre-OAK) Sobol path, SURVEY section 2
        "oak.normalising_flow": {"make_sinharcsinh", "make_standardizer"},     # builders of TFP bijector objects: the flow is restated in closed form
    }
    missing = []
    for modname, names in fx["module_members"].items():
        mod = importlib.import_module(modname)
        missing += [f"{modname}.{n}" for n in names if n not in out_of_scope.get(modname, ()) and not hasattr(mod, n)]
    owners = {"oak_model": "oak.model_utils", "OAKKernel": "oak.oak_kernel", "KernelComponenent": "oak.oak_kernel",
              "OrthogonalRBFKernel": "oak.ortho_rbf_kernel", "OrthogonalBinary": "oak.ortho_binary_kernel",
              "OrthogonalCategorical": "oak.ortho_categorical_kernel", "UniformMeasure": "oak.input_measures", "GaussianMeasure": "oak.input_measures",
              "EmpiricalMeasure": "oak.input_measures", "MOGMeasure": "oak.input_measures", "Normalizer": "oak.normalising_flow"}
    for cname, members in fx["class_members"].items():
        cls = getattr(importlib.import_module(owners[cname]), cname)
        if cname == "Normalizer":       # KL_objective is an instance attribute there (a callable object with value_and_grad)
            members = [m for m in members if m != "KL_objective"]
        missing += [f"{cname}.{m}" for m in members if not hasattr(cls, m)]
    assert not missing, missing


def test_model_defaults_and_gmm_estimate_against_the_reference_executed_fixture():
    """Same fixture: the attributes a freshly constructed reference ``oak_model`` carries (defaults and a fully specified call) and
    the mixture the reference's ``estimate_one_dim_gmm`` fits (scikit-learn GaussianMixture, spherical, random_state=0)."""
    import json
    from pathlib import Path
    import sklearn
    from oak.model_utils import estimate_one_dim_gmm, oak_model
    fx = np.load(Path(__file__).resolve().parent / "golden" / "reference_host_logic.npz")
    custom = dict(max_interaction_depth=3, num_inducing=50, lengthscale_bounds=[0.01, 10.0], binary_feature=[1], categorical_feature=[2],
                  empirical_measure=[0], use_sparsity_prior=False, gmm_measure=[0, 2, 0], sparse=True, use_normalising_flow=False,
                  share_var_across_orders=False)
    for key, model in (("oak_model_default_attrs", oak_model()), ("oak_model_custom_attrs", oak_model(**custom))):
        ref = json.loads(str(fx[key]))
        mine = vars(model)
        for name, value in ref.items():
            assert name in mine, f"{key}: attribute {name} missing"
            got = list(mine[name]) if isinstance(mine[name], tuple) else mine[name]
            assert got == value, f"{key}: {name} = {got!r}, the reference's {value!r}"
    if str(fx["sklearn_version"]) == sklearn.__version__:
        mog = estimate_one_dim_gmm(3, fx["gmm_x"])
        for name in ("means", "variances", "weights"):
            np.testing.assert_array_equal(np.asarray(getattr(mog, name), dtype=np.float64).reshape(-1), fx[f"gmm_{name}"])


def test_component_list_is_a_list_that_fills_itself():
    """get_list_representation's kernel_list is built lazily (41 449 components at D = 32, depth 4) but must behave as the list the
    reference returns (oak/oak_kernel.py:338-364): isinstance, identity of repeated indexing, concatenation from both sides,
    append, slices, iteration, equality."""
    k = OAKKernel([gpflow.RBF] * 3, num_dims=3, max_interaction_depth=2, constrain_orthogonal=True)
    sel, comps = get_list_representation(k, num_dims=3)
    assert isinstance(comps, list) and len(comps) == len(sel) == 7
    assert comps[2] is comps[2] and comps[-1] is comps[6]
    assert comps[2].iComponent_list == sel[2]
    with pytest.raises(IndexError):
        comps[7]
    both = comps + ["x"]
    assert type(both) is list and len(both) == 8 and both[2] is comps[2]
    assert (["y"] + comps)[1] is comps[0]
    assert [c.iComponent_list for c in comps] == sel
    assert [c.iComponent_list for c in comps[1:3]] == sel[1:3]
    comps.append("z")
    assert len(comps) == 8 and comps[-1] == "z" and comps[2].iComponent_list == sel[2]
    _, again = get_list_representation(k, num_dims=3)
    assert comps[:7] != again[:]          # components compare by identity, as objects do


def test_packed_subsets_are_told_apart_from_a_pair_of_arrays():
    """HipContext.pack_subsets returns a PackedSubsets; only that type is taken as already packed -- two subsets passed as a tuple of
    arrays must not be misread as (indices, offsets)."""
    packed = _capi.HipContext.pack_subsets([[0], [1, 2], [0, 1, 2]])
    assert isinstance(packed, _capi.PackedSubsets)
    flat, off = packed
    assert flat.tolist() == [0, 1, 2, 0, 1, 2] and off.tolist() == [0, 1, 3, 6]
    assert _capi.HipContext.pack_subsets(packed) is packed
    two = _capi.HipContext.pack_subsets((np.array([0]), np.array([1, 2])))
    assert two[0].tolist() == [0, 1, 2] and two[1].tolist() == [0, 1, 3]
    with pytest.raises(ValueError):
        _capi.PackedSubsets(np.array([0, 1]), np.array([1, 2]))

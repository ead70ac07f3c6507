"""Generates tests/golden/reference_sobol_L.npz and tests/golden/reference_host_logic.npz by EXECUTING THE REFERENCE'S OWN CODE (build container only).

    python tests/golden/make_reference_golden.py          # needs /root/reference; nothing of it travels with the repo

What is executed: ``oak.utils.f1 / f2 / f3 / f4`` (oak/utils.py:116-165), ``compute_L`` (:221-240) and
``compute_L_binary_kernel`` (:243-272) -- pure NumPy functions.  The module that holds them imports TensorFlow, GPflow
and TensorFlow-Probability at its top (oak/utils.py:6-24), none of which exists in this image (and none can be
installed: no network).  Those imports are satisfied with INERT placeholders: objects that implement nothing -- every
attribute of a placeholder is another placeholder, calling one returns a placeholder.  They only let the module body
run to the end; a function that touched one would return a placeholder instead of numbers, so the generator
 * calls ONLY the six functions above, whose bodies use nothing but NumPy (read them: no ``tf.``, no ``gpflow.``), and
 * refuses to write anything that is not a plain float64 ndarray of the expected shape.
Second fixture (r04, ``reference_host_logic.npz``): the data-preparation logic that is plain NumPy / scikit-learn in the reference --
``oak.model_utils._calculate_features`` (oak/model_utils.py:703-750: feature-type index sets, p0 of binary and p of categorical columns),
``oak.model_utils.get_kmeans_centers`` (:31-41) and ``oak.utils.initialize_kmeans_with_binary / _with_categorical`` (oak/utils.py:533-574), the
last three running the installed scikit-learn's KMeans exactly as the reference calls it.  ``oak.model_utils`` needs a few more inert
placeholders to load (tikzplotlib, gpflow.models.training_mixins, tensorflow_probability.distributions); the only placeholder call on these
paths is ``tf.random.set_seed(44)`` in get_kmeans_centers, whose result is discarded.
Third file (``reference_signatures.json``): ``inspect.signature`` of the reference's public callables on the path -- parameter names, kinds and
plain defaults, no body executed -- against which tests/test_host_logic.py holds the mirrors' signatures.
Nothing else of the reference is pinned by these files: the ELBO scalar, predictive variance and the transforms stay
"parity unpinned" (DESIGN.md section 3).

The fixture holds inputs and outputs only (a few KB of numbers).  It is labelled inside the file:
``label = "reference-executed: f1-f4 / compute_L / compute_L_binary_kernel only"``.
"""
import sys
import types
from pathlib import Path

sys.dont_write_bytecode = True          # /root/reference is read-only by contract: importing from it must not leave __pycache__ there

import numpy as np

REFERENCE = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "reference_sobol_L.npz"
LABEL = "reference-executed: f1-f4 / compute_L / compute_L_binary_kernel only"
PLACEHOLDERS = ("tensorflow", "tensorflow_probability", "gpflow", "gpflow.config", "gpflow.covariances", "gpflow.covariances.dispatch",
                "gpflow.models", "gpflow.base", "gpflow.utilities", "gpflow.kernels", "gpflow.inducing_variables", "tensorflow_probability.python",
                "tensorflow_probability.python.bijectors", "gpflow.models.training_mixins", "tensorflow_probability.distributions", "tikzplotlib")
OUT_HOST = Path(__file__).resolve().parent / "reference_host_logic.npz"
LABEL_HOST = "reference-executed: _calculate_features / get_kmeans_centers / initialize_kmeans_with_binary / _with_categorical / estimate_one_dim_gmm / oak_model.__init__ only"


class Inert(types.ModuleType):
    """Implements nothing.  Attribute -> Inert, call -> Inert, base class -> dropped."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        child = Inert(f"{self.__name__}.{name}")
        setattr(self, name, child)
        return child

    def __call__(self, *args, **kwargs):
        return Inert(f"{self.__name__}()")

    def __mro_entries__(self, bases):
        return ()

    def __iter__(self):
        return iter(())


def load_reference_utils():
    if not REFERENCE.is_dir():
        raise SystemExit("the reference tree is not here: this generator only runs in the build container")
    for name in PLACEHOLDERS:
        sys.modules.setdefault(name, Inert(name))
    sys.path.insert(0, str(REFERENCE))
    import oak.utils as ref_utils       # the reference's module, from /root/reference
    assert Path(ref_utils.__file__).resolve().is_relative_to(REFERENCE), ref_utils.__file__
    return ref_utils


def load_reference_model_utils():
    load_reference_utils()
    import matplotlib
    matplotlib.use("Agg")
    import oak.model_utils as ref_mu
    assert Path(ref_mu.__file__).resolve().is_relative_to(REFERENCE), ref_mu.__file__
    return ref_mu


def plain(a, shape):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or a.shape != shape or not np.isfinite(a).all():
        raise SystemExit(f"not a plain finite float64 array of shape {shape}: {type(a)} -- a placeholder was touched?")
    return a


def main():
    ref = load_reference_utils()
    rng = np.random.default_rng(20240601)
    out = {"label": np.array(LABEL)}
    # f1..f4 on scattered arguments (the closed forms of eq. 44-47)
    n = 64
    x, y = rng.normal(size=n) * 1.5, rng.normal(size=n) * 1.5
    params = np.array([[1.0, 1.0, 1.0, 0.0], [0.7, 0.35, 1.0, 0.0], [1.3, 2.4, 1.6, 0.4], [2.1, 0.9, 0.5, -0.8], [0.4, 5.0, 2.2, 1.1]])
    out["f_x"], out["f_y"], out["f_params"] = x, y, params              # columns: sigma, lengthscale, delta, mu
    for k, fn in enumerate((ref.f1, ref.f2, ref.f3, ref.f4), start=1):
        out[f"f{k}"] = np.stack([plain(fn(x, y, *p), (n,)) for p in params])
    # compute_L: Gaussian-measure RBF sub-kernel; column `dim` of X
    X = rng.normal(size=(23, 3))
    X[:, 2] *= 2.5
    L_params = np.array([[1.0, 1.0, 0, 1.0, 0.0], [0.6, 1.7, 1, 1.0, 0.0], [2.2, 0.8, 2, 1.4, 0.3], [0.25, 2.5, 1, 0.7, -0.5]])
    out["L_X"], out["L_params"] = X, L_params                            # columns: lengthscale, variance, dim, delta, mu
    out["L"] = np.stack([plain(ref.compute_L(X, p[0], p[1], int(p[2]), p[3], p[4]), (23, 23)) for p in L_params])
    # compute_L_binary_kernel
    Xb = rng.integers(0, 2, size=(19, 2)).astype(np.float64)
    b_params = np.array([[0.5, 1.0, 0], [0.77, 1.0, 1], [0.12, 2.3, 0], [1.0, 0.6, 1], [0.0, 1.0, 0]])
    out["Lb_X"], out["Lb_params"] = Xb, b_params                         # columns: p0, variance, dim
    out["Lb"] = np.stack([plain(ref.compute_L_binary_kernel(Xb, p[0], p[1], int(p[2])), (19, 19)) for p in b_params])
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes): {LABEL}")
    host_logic(ref)


def host_logic(ref_utils):
    import contextlib, io, sklearn
    mu = load_reference_model_utils()
    rng = np.random.default_rng(20240602)
    n = 240
    X = rng.normal(size=(n, 6)) * np.array([1.0, 2.0, 0.5, 1.0, 1.0, 3.0]) + np.array([0.0, 1.0, -2.0, 0.0, 0.0, 0.5])
    X[:, 1] = (rng.random(n) < 0.35).astype(float)                     # binary
    X[:, 3] = rng.choice(4, size=n, p=[0.1, 0.2, 0.3, 0.4]).astype(float)   # categorical, 4 classes
    X[:, 4] = (rng.random(n) < 0.6).astype(float)                      # binary
    out = {"label": np.array(LABEL_HOST), "sklearn_version": np.array(sklearn.__version__), "X": X}
    with contextlib.redirect_stdout(io.StringIO()):                    # the function prints its index lists
        cont, binary, cat, p0, p = mu._calculate_features(X, categorical_feature=[3], binary_feature=[1, 4])
        cont_all, bin_none, cat_none, p0_none, p_none = mu._calculate_features(X[:, [0, 2, 5]], None, None)
    assert p0_none is None and p_none is None and bin_none == [] and cat_none == []
    out["cf_continuous"], out["cf_binary"], out["cf_categorical"] = np.array(cont), np.array(binary), np.array(cat)
    out["cf_all_continuous"] = np.array(cont_all)
    out["cf_p0"] = np.array([np.nan if v is None else float(v) for v in p0])            # NaN stands for the reference's None
    out["cf_p_col3"] = plain(np.asarray(p[3], dtype=np.float64), (4, 1))
    assert all(v is None for i, v in enumerate(p) if i != 3)
    K = 9
    out["K"] = np.array(K)
    out["kmeans_centers"] = plain(np.asarray(mu.get_kmeans_centers(X[:, [0, 2, 5]], K), dtype=np.float64), (K, 3))
    out["init_binary"] = plain(np.asarray(ref_utils.initialize_kmeans_with_binary(X[:, [0, 1, 2, 4, 5]], binary_index=[1, 3], continuous_index=[0, 2, 4],
                                                                                  n_clusters=2), dtype=np.float64), (2, 5))
    out["init_categorical"] = plain(np.asarray(ref_utils.initialize_kmeans_with_categorical(X[:, [0, 3, 2, 5]], binary_index=[], categorical_index=[1],
                                                                                            continuous_index=[0, 2, 3], n_clusters=4),
                                               dtype=np.float64), (4, 4))
    # estimate_one_dim_gmm (oak/model_utils.py:753-770): scikit-learn's GaussianMixture as the reference configures it, wrapped in its MOGMeasure
    xg = np.concatenate([rng.normal(-2.0, 0.5, 150), rng.normal(1.5, 1.0, 250)])
    mog = mu.estimate_one_dim_gmm(3, xg)
    out["gmm_x"] = xg
    out["gmm_means"], out["gmm_variances"], out["gmm_weights"] = (plain(np.asarray(getattr(mog, k), dtype=np.float64), (3,)) for k in ("means", "variances", "weights"))
    # the attributes an oak_model instance starts with (oak/model_utils.py:195-247): plain Python, names and default values
    import json
    def attrs(obj):
        return {k: (v if v is None or isinstance(v, (bool, int, float, str, list)) else f"<{type(v).__name__}>") for k, v in vars(obj).items()}
    out["oak_model_default_attrs"] = np.array(json.dumps(attrs(mu.oak_model()), sort_keys=True))
    out["oak_model_custom_attrs"] = np.array(json.dumps(attrs(mu.oak_model(max_interaction_depth=3, num_inducing=50, lengthscale_bounds=[0.01, 10.0],
                                                                             binary_feature=[1], categorical_feature=[2], empirical_measure=[0],
                                                                             use_sparsity_prior=False, gmm_measure=[0, 2, 0], sparse=True,
                                                                             use_normalising_flow=False, share_var_across_orders=False)), sort_keys=True))
    np.savez_compressed(OUT_HOST, **out)
    print(f"wrote {OUT_HOST} ({OUT_HOST.stat().st_size} bytes): {LABEL_HOST}")
    signatures(ref_utils, mu)


def signatures(ref_utils, ref_mu):
    """Parameter names, kinds and defaults of the reference's public callables on the path, read with ``inspect.signature`` from the
    imported reference modules (an interface description: no body is executed).  Defaults that are placeholders or other objects are
    recorded by type name only."""
    import inspect, json
    import oak.oak_kernel as ok, oak.ortho_rbf_kernel as orb, oak.ortho_binary_kernel as ob, oak.ortho_categorical_kernel as oc
    import oak.input_measures as im

    def default(v):
        if v is inspect.Parameter.empty:
            return "<required>"
        if v is None or isinstance(v, (bool, int, float, str)):
            return v
        if isinstance(v, (list, tuple)) and all(x is None or isinstance(x, (bool, int, float, str)) for x in v):
            return list(v)
        return f"<{type(v).__name__}>"

    def sig(fn):
        return [[p.name, p.kind.name, default(p.default)] for p in inspect.signature(fn).parameters.values()]

    targets = {
        "oak_model.__init__": ref_mu.oak_model.__init__, "oak_model.fit": ref_mu.oak_model.fit, "oak_model.optimise": ref_mu.oak_model.optimise,
        "oak_model.predict": ref_mu.oak_model.predict, "oak_model.get_loglik": ref_mu.oak_model.get_loglik,
        "oak_model.get_sobol": ref_mu.oak_model.get_sobol, "oak_model.plot": ref_mu.oak_model.plot,
        "create_model_oak": ref_mu.create_model_oak, "get_kmeans_centers": ref_mu.get_kmeans_centers, "save_model": ref_mu.save_model,
        "load_model": ref_mu.load_model, "_calculate_features": ref_mu._calculate_features,
        "OAKKernel.__init__": ok.OAKKernel.__init__, "OAKKernel.compute_additive_terms": ok.OAKKernel.compute_additive_terms,
        "OAKKernel.K": ok.OAKKernel.K, "OAKKernel.K_diag": ok.OAKKernel.K_diag, "KernelComponenent.__init__": ok.KernelComponenent.__init__,
        "get_list_representation": ok.get_list_representation, "bounded_param": ok.bounded_param,
        "OrthogonalRBFKernel.__init__": orb.OrthogonalRBFKernel.__init__, "OrthogonalBinary.__init__": ob.OrthogonalBinary.__init__,
        "OrthogonalCategorical.__init__": oc.OrthogonalCategorical.__init__,
        "UniformMeasure.__init__": im.UniformMeasure.__init__, "GaussianMeasure.__init__": im.GaussianMeasure.__init__,
        "EmpiricalMeasure.__init__": im.EmpiricalMeasure.__init__, "MOGMeasure.__init__": im.MOGMeasure.__init__,
        "compute_sobol_oak": ref_utils.compute_sobol_oak, "get_prediction_component": ref_utils.get_prediction_component,
        "get_model_sufficient_statistics": ref_utils.get_model_sufficient_statistics, "compute_L": ref_utils.compute_L,
        "compute_L_binary_kernel": ref_utils.compute_L_binary_kernel, "compute_L_categorical_kernel": ref_utils.compute_L_categorical_kernel,
        "compute_L_empirical_measure": ref_utils.compute_L_empirical_measure, "f1": ref_utils.f1,
        "initialize_kmeans_with_binary": ref_utils.initialize_kmeans_with_binary,
        "initialize_kmeans_with_categorical": ref_utils.initialize_kmeans_with_categorical,
    }
    out = {"label": "reference-introspected: inspect.signature of the imported reference callables (names, kinds, plain defaults)",
           "signatures": {}}
    for name, fn in targets.items():
        try:
            out["signatures"][name] = sig(fn)
        except (TypeError, ValueError) as e:          # e.g. a tf.function-wrapped callable that became a placeholder
            out["signatures"][name] = f"<not introspectable: {type(e).__name__}>"
    # members the reference's classes define themselves, and the functions / classes its modules define (not what they import)
    import oak.normalising_flow as nf
    classes = {"oak_model": ref_mu.oak_model, "OAKKernel": ok.OAKKernel, "KernelComponenent": ok.KernelComponenent,
               "OrthogonalRBFKernel": orb.OrthogonalRBFKernel, "OrthogonalBinary": ob.OrthogonalBinary, "OrthogonalCategorical": oc.OrthogonalCategorical,
               "UniformMeasure": im.UniformMeasure, "GaussianMeasure": im.GaussianMeasure, "EmpiricalMeasure": im.EmpiricalMeasure,
               "MOGMeasure": im.MOGMeasure, "Normalizer": nf.Normalizer}
    out["class_members"] = {name: sorted(k for k, v in vars(cls).items() if callable(v) or isinstance(v, property)) for name, cls in classes.items()}
    modules = {"oak.model_utils": ref_mu, "oak.utils": ref_utils, "oak.oak_kernel": ok, "oak.input_measures": im, "oak.normalising_flow": nf,
               "oak.ortho_rbf_kernel": orb, "oak.ortho_binary_kernel": ob, "oak.ortho_categorical_kernel": oc}
    out["module_members"] = {name: sorted(k for k, v in vars(mod).items() if (inspect.isfunction(v) or inspect.isclass(v)) and getattr(v, "__module__", None) == mod.__name__)
                             for name, mod in modules.items()}
    dst = Path(__file__).resolve().parent / "reference_signatures.json"
    dst.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print(f"wrote {dst} ({dst.stat().st_size} bytes)")


if __name__ == "__main__":
    main()

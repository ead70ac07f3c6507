"""The C-ABI library loads and exports every symbol include/oak_hip.h declares (no compute: runs without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

from oak import _capi

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "oak_hip.h"
BENCH_HEADER = ROOT / "include" / "oak_hip_bench.h"          # measurement hooks: exported and bound, not part of the boundary


def declared_functions(headers=(HEADER, BENCH_HEADER)):
    names = set()
    for h in headers:
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        names.update(re.findall(r"\b(oak_[A-Za-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_measurement_hooks_live_in_their_own_header():
    assert not [n for n in declared_functions((HEADER,)) if n.startswith("oak_bench_")]
    assert all(n.startswith("oak_bench_") for n in declared_functions((BENCH_HEADER,)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("oak_gram", "oak_gram_diag", "oak_sgpr_elbo", "oak_sgpr_predict", "oak_sgpr_elbo_grad", "oak_sobol",
                 "oak_comm_allreduce_stats", "oak_gpr_log_marginal"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert _capi.LIB_PATH.exists(), f"{_capi.LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
    lib = ctypes.CDLL(str(_capi.LIB_PATH))
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"symbols declared in oak_hip.h but not exported: {missing}"


def test_ctypes_signatures_cover_the_header():
    assert sorted(_capi.SIGNATURES) == declared_functions()
    lib = _capi.load_library()
    assert lib.oak_version().decode().startswith("oak_hip")


def test_product_path_fails_loudly_without_a_device():
    """No CPU fallback: without a HIP device context creation raises instead of computing anything."""
    if _capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_capi.OakHipError):
        _capi.HipContext(0)
    from oak.oak_kernel import OAKKernel
    from oak import gpflow_lite as gpflow
    import numpy as np
    k = OAKKernel([gpflow.RBF], num_dims=1, max_interaction_depth=1, constrain_orthogonal=True)
    with pytest.raises(_capi.OakHipError):
        k.K(np.zeros((3, 1)))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_capi.OakHipError):
        _capi.load_library()

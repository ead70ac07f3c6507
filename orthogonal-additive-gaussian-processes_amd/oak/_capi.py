"""ctypes binding of ``liboak_hip.so`` (C ABI declared in ``include/oak_hip.h``).

This is the only module that touches the native library.  There is no CPU
fallback anywhere in the package: if the shared object is missing, or no HIP
device is usable, the first compute call raises :class:`OakHipError`.
"""
from __future__ import annotations

import atexit
import ctypes as C
import weakref
import itertools
import os
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

OAK_OK, OAK_E_ARG, OAK_E_HIP, OAK_E_NOTPD, OAK_E_NCCL, OAK_E_STATE = 0, -1, -2, -3, -4, -5
DIM_RBF, DIM_BINARY, DIM_CATEGORICAL = 0, 1, 2
MEAS_NONE, MEAS_GAUSSIAN, MEAS_UNIFORM, MEAS_EMPIRICAL, MEAS_MOG = 0, 1, 2, 3, 4
MAX_DIMS, MAX_DEPTH = 64, 32        # MAX_DEPTH: EFFECTIVE depth min(max_interaction_depth, D) of the fused kernels
MAX_DEPTH_DESC = 64                  # what a description may carry (deeper than 32: explicit Gram entry points only)

_PKG_ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = Path(os.environ.get("OAK_HIP_LIB", _PKG_ROOT / "lib" / "liboak_hip.so"))


class OakHipError(RuntimeError):
    """Raised when the native HIP library is missing or reports a runtime failure."""

    def __init__(self, message: str, status: int = OAK_E_HIP):
        super().__init__(message)
        self.status = status


class NotPositiveDefiniteError(OakHipError, ArithmeticError):
    """Cholesky met a non-positive pivot (the reference surfaces tf.errors.InvalidArgumentError)."""


class KernelDescStruct(C.Structure):
    _fields_ = [
        ("num_dims", C.c_int32), ("max_depth", C.c_int32), ("share_var", C.c_int32), ("n_order_var", C.c_int32),
        ("order_var", C.POINTER(C.c_double)), ("dim_type", C.POINTER(C.c_int32)),
        ("active_col", C.POINTER(C.c_int32)), ("lengthscale", C.POINTER(C.c_double)),
        ("base_var", C.POINTER(C.c_double)), ("measure", C.POINTER(C.c_int32)),
        ("meas_p0", C.POINTER(C.c_double)), ("meas_p1", C.POINTER(C.c_double)),
        ("meas_k", C.POINTER(C.c_int32)), ("meas_off", C.POINTER(C.c_int32)),
        ("meas_data", C.POINTER(C.c_double)), ("meas_data_len", C.c_int32), ("grad_base_var", C.c_int32),
        ("extra_col_off", C.POINTER(C.c_int32)), ("extra_cols", C.POINTER(C.c_int32)),
    ]


_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int32)
_CTX = C.c_void_p
_HOST_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_void_p)   # oak_host_allreduce_fn
_DESC = C.POINTER(KernelDescStruct)

# name -> (restype, argtypes); every symbol include/oak_hip.h declares
SIGNATURES = {
    "oak_last_error": (C.c_char_p, []),
    "oak_version": (C.c_char_p, []),
    "oak_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "oak_ctx_create": (C.c_int, [C.c_int, C.POINTER(_CTX)]),
    "oak_ctx_destroy": (C.c_int, [_CTX]),
    "oak_sync": (C.c_int, [_CTX]),
    "oak_debug_state": (C.c_int, [C.c_char_p, C.c_int64]),
    "oak_last_timing": (C.c_int, [_CTX, C.c_char_p, _D, _I]),
    "oak_reset_timings": (C.c_int, [_CTX]),
    "oak_device_mem_info": (C.c_int, [_CTX, _D, _D]),
    "oak_gram": (C.c_int, [_CTX, _DESC, _D, C.c_int64, _D, C.c_int64, C.c_int32, _D]),
    "oak_gram_diag": (C.c_int, [_CTX, _DESC, _D, C.c_int64, C.c_int32, _D]),
    "oak_set_gram_form": (C.c_int, [_CTX, C.c_int32]),
    "oak_gram_f32": (C.c_int, [_CTX, _DESC, _D, C.c_int64, _D, C.c_int64, C.c_int32, C.POINTER(C.c_float)]),
    "oak_gram_component": (C.c_int, [_CTX, _DESC, _I, C.c_int32, C.c_int32, _D, C.c_int64, _D, C.c_int64, C.c_int32, _D]),
    "oak_gram_component_diag": (C.c_int, [_CTX, _DESC, _I, C.c_int32, C.c_int32, _D, C.c_int64, C.c_int32, _D]),
    "oak_sgpr_set_data": (C.c_int, [_CTX, _D, _D, C.c_int64, C.c_int32]),
    "oak_sgpr_set_targets": (C.c_int, [_CTX, _D, C.c_int64]),
    "oak_sgpr_set_extra_targets": (C.c_int, [_CTX, _D, C.c_int64, C.c_int32]),
    "oak_sgpr_select_output": (C.c_int, [_CTX, C.c_int32]),
    "oak_sgpr_set_inducing": (C.c_int, [_CTX, _D, C.c_int64, C.c_int32]),
    "oak_sgpr_set_panel_rows": (C.c_int, [_CTX, C.c_int64]),
    "oak_sgpr_local_stats": (C.c_int, [_CTX, _DESC, C.c_double]),
    "oak_sgpr_set_route": (C.c_int, [_CTX, C.c_int32]),
    "oak_sgpr_set_global_rows": (C.c_int, [_CTX, C.c_int64]),
    "oak_sgpr_set_precision": (C.c_int, [_CTX, C.c_int32]),
    "oak_sgpr_stats_precision": (C.c_int, [_CTX, _I]),
    "oak_sgpr_stats_whitened": (C.c_int, [_CTX, _I]),
    "oak_sgpr_stats_len": (C.c_int64, [_CTX]),
    "oak_sgpr_get_stats": (C.c_int, [_CTX, _D]),
    "oak_sgpr_set_stats": (C.c_int, [_CTX, _D, C.c_int32]),
    "oak_sgpr_tail": (C.c_int, [_CTX, _DESC, C.c_double, C.c_double, _D, _D]),
    "oak_sgpr_elbo": (C.c_int, [_CTX, _DESC, C.c_double, C.c_double, _D]),
    "oak_sgpr_alpha": (C.c_int, [_CTX, _D]),
    "oak_sgpr_last_terms": (C.c_int, [_CTX, _D]),
    "oak_sgpr_predict": (C.c_int, [_CTX, _DESC, _D, C.c_int64, C.c_int32, _D, _D]),
    "oak_grad_len": (C.c_int64, [_DESC]),
    "oak_sgpr_elbo_grad": (C.c_int, [_CTX, _DESC, C.c_double, C.c_double, _D, _D]),
    "oak_sgpr_effective_L": (C.c_int, [_CTX, _D]),
    "oak_gpr_chol": (C.c_int, [_CTX, _D]),
    "oak_sgpr_elbo_grad_z": (C.c_int, [_CTX, _DESC, C.c_double, C.c_double, _D, _D, _D]),
    "oak_svgp_elbo_grad": (C.c_int, [_CTX, _DESC, _D, _D, C.c_double, _D, _D, C.c_int32, C.c_int32, C.c_double, _D, _D, _D, _D]),
    "oak_svgp_predict": (C.c_int, [_CTX, _DESC, _D, _D, C.c_double, _D, C.c_int64, C.c_int32, _D, _D, _D, _D, _D, _D, C.c_int32,
                                   C.c_int32, C.c_double]),
    "oak_svgp_posterior": (C.c_int, [_CTX, _DESC, _D, _D, C.c_double, _D, _D]),
    "oak_gpr_set_data": (C.c_int, [_CTX, _D, _D, C.c_int64, C.c_int32]),
    "oak_gpr_set_targets": (C.c_int, [_CTX, _D, C.c_int64]),
    "oak_gpr_log_marginal": (C.c_int, [_CTX, _DESC, C.c_double, _D]),
    "oak_gpr_alpha": (C.c_int, [_CTX, _D]),
    "oak_gpr_predict": (C.c_int, [_CTX, _DESC, _D, C.c_int64, C.c_int32, _D, _D]),
    "oak_gpr_log_marginal_grad": (C.c_int, [_CTX, _DESC, C.c_double, _D, _D]),
    "oak_sobol": (C.c_int, [_CTX, _DESC, _D, C.c_int64, C.c_int32, _D, _I, _I, C.c_int32, C.c_int32, C.c_double, C.c_double, _D]),
    "oak_sobol_collective": (C.c_int, [_CTX, _DESC, _D, C.c_int64, C.c_int32, _D, _I, _I, C.c_int32, C.c_int32, C.c_double, C.c_double, _D]),
    "oak_sobol_set_path": (C.c_int, [_CTX, C.c_int32]),
    "oak_sobol_last_info": (C.c_int, [_CTX, _D]),
    "oak_sobol_L": (C.c_int, [_CTX, _DESC, C.c_int32, C.c_double, C.c_double, C.c_double, _D, C.c_int64, C.c_int32, _D]),
    "oak_cov_x_s": (C.c_int, [_CTX, _DESC, C.c_int32, _D, C.c_int64, C.c_int32, _D, _D]),
    "oak_additive_terms": (C.c_int, [_CTX, _D, C.c_int32, C.c_int64, C.c_int32, _D]),
    "oak_component_predict": (C.c_int, [_CTX, _DESC, _D, C.c_int64, _D, C.c_int64, C.c_int32, _D, _I, _I, C.c_int32, C.c_int32, _D]),
    "oak_comm_unique_id": (C.c_int, [C.c_char_p]),
    "oak_comm_init": (C.c_int, [_CTX, C.c_char_p, C.c_int32, C.c_int32]),
    "oak_comm_destroy": (C.c_int, [_CTX]),
    "oak_comm_init_loopback": (C.c_int, [_CTX, C.c_int32]),
    "oak_comm_allreduce_stats": (C.c_int, [_CTX]),
    "oak_comm_allreduce_host": (C.c_int, [_CTX, _D, C.c_int64]),
    "oak_comm_init_host": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "oak_comm_info": (C.c_int, [C.c_char_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "oak_comm_allgatherv": (C.c_int, [_CTX, _D, C.POINTER(C.c_int64), C.c_int32]),
    "oak_runtime_shutdown": (C.c_int, []),
    "oak_bench_gram_resident": (C.c_int, [_CTX, _DESC, _D]),
    "oak_bench_potrf": (C.c_int, [_CTX, C.c_int64, C.c_int32, _D, _D]),
    "oak_bench_trsm": (C.c_int, [_CTX, _D, C.c_int64, _D, C.c_int64, C.c_int32, C.c_int32, _D]),
    "oak_bench_crt_info": (C.c_int, [_CTX, C.POINTER(C.c_int64)]),
    "oak_flow_objective": (C.c_int, [_CTX, _D, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, _D, _D]),
    "oak_flow_forward": (C.c_int, [_CTX, _D, C.c_int64, C.c_int32, C.c_int32, _I, _D, _D]),
    "oak_kmeans_plusplus": (C.c_int, [_CTX, _D, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64, _D, C.c_int32, _D,
                                      C.POINTER(C.c_int64)]),
    "oak_kmeans": (C.c_int, [_CTX, _D, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _D, C.c_int32, C.c_double, _D, _I,
                             _D, _I]),
}

_lib = None


def load_library():
    """dlopen the in-tree shared object; raise OakHipError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise OakHipError(
            f"native library {LIB_PATH} not found: build it with "
            f"`make -C {_PKG_ROOT / 'csrc'}` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "The OAK HIP path has no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    atexit.register(_shutdown_runtime)
    return lib


_live_contexts = weakref.WeakSet()


def _profiler_attached() -> bool:
    """rocprofv3 (or another rocprofiler-sdk tool) is loaded into this process."""
    env = os.environ
    return any("rocprofiler" in env.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) or "ROCPROFILER_LIBRARY_CTOR" in env


def _shutdown_runtime():
    """Interpreter exit.  Only when a profiler is attached (its finaliser crashes on CU-masked queues left alive) or OAK_SHUTDOWN_AT_EXIT=1
    asks for it: close the contexts still alive, then let the library destroy its pooled streams (oak_runtime_shutdown) -- while the process
    is still whole (a C-level exit handler would run after the profiler's per-thread state is gone).  Otherwise nothing is torn down at
    exit, as before r05: this handler makes no HIP call (hipStreamDestroy is the call that can deadlock against the runtime's event thread,
    DESIGN.md section 6a) and the driver reclaims streams and memory -- a HipContext finalised by the garbage collector still runs
    oak_ctx_destroy (stream synchronisation, hipFree / hipHostFree; its streams go back to the pool, none is destroyed)."""
    want = os.environ.get("OAK_SHUTDOWN_AT_EXIT")
    if want == "0" or (want is None and not _profiler_attached()):
        return
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:                       # noqa: BLE001
            pass
    global _default_ctx
    _default_ctx = None
    if _lib is not None:
        try:
            _lib.oak_runtime_shutdown()
        except Exception:                       # noqa: BLE001
            pass


def _check(status: int):
    if status == OAK_OK:
        return
    msg = load_library().oak_last_error().decode("utf-8", "replace")
    if status == OAK_E_NOTPD:
        raise NotPositiveDefiniteError(msg, status)
    if status == OAK_E_ARG:
        raise ValueError(msg)
    raise OakHipError(msg, status)


def _f64(a, ndim=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise ValueError(f"expected a {ndim}-D array, got shape {a.shape}")
    return a


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_D)


def _ip(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_I)


class KernelDesc:
    """Owns the numpy arrays behind an ``oak_kernel_desc`` POD.

    ``spec`` schema (plain data, also used by the test oracle):
        {"dims": [...], "order_variances": [...], "max_interaction_depth": R, "share_var_across_orders": bool}
    dim entries: {"type": "rbf", "lengthscale", "variance", "measure": None | ("gaussian", mu, var) |
    ("uniform", a, b) | ("empirical", loc, w) | ("mog", means, vars, w), "active_dim": col (optional)},
    {"type": "binary", "p0", "variance"}, {"type": "categorical", "p", "W", "kappa", "variance"}.
    An unconstrained rbf dim may carry "active_dims": [c0, c1, ...] instead of "active_dim": one RBF over those columns
    (OAKKernel(active_dims=[[0, 1], ...]), oak_kernel.py:74-82); every model path takes it except Sobol (undefined for unconstrained kernels).
    """

    def __init__(self, spec: dict):
        dims = spec["dims"]
        D = len(dims)
        R = int(spec["max_interaction_depth"])
        if not (1 <= D <= MAX_DIMS):
            raise ValueError(f"number of sub-kernels {D} outside [1, {MAX_DIMS}]")
        if not (0 <= R <= MAX_DEPTH_DESC):
            raise ValueError(f"max_interaction_depth {R} outside [0, {MAX_DEPTH_DESC}]")
        share = bool(spec.get("share_var_across_orders", True))
        ov = _f64(np.asarray(spec["order_variances"], dtype=np.float64).reshape(-1))
        if ov.size != (R + 1 if share else 1):
            raise ValueError("order_variances has the wrong length")
        self.order_var = ov
        self.dim_type = np.zeros(D, np.int32)
        self.active_col = np.zeros(D, np.int32)
        self.lengthscale = np.ones(D, np.float64)
        self.base_var = np.ones(D, np.float64)
        self.measure = np.zeros(D, np.int32)
        self.meas_p0 = np.zeros(D, np.float64)
        self.meas_p1 = np.zeros(D, np.float64)
        self.meas_k = np.zeros(D, np.int32)
        self.meas_off = np.zeros(D, np.int32)
        data = []
        off = 0
        self.cat_blocks = {}   # dim -> (offset, C) of the categorical table inside meas_data
        extra_off, extra_cols = [0], []
        for d, dim in enumerate(dims):
            group = [int(c) for c in dim["active_dims"]] if dim.get("active_dims") is not None else [int(dim.get("active_dim", d))]
            if len(group) > 1 and not (dim["type"] == "rbf" and dim.get("measure") is None):
                raise NotImplementedError("only an unconstrained RBF sub-kernel reads several columns")
            self.active_col[d] = group[0]
            extra_cols += group[1:]
            extra_off.append(len(extra_cols))
            self.base_var[d] = float(np.asarray(dim.get("variance", 1.0)).reshape(-1)[0])
            t = dim["type"]
            if t == "rbf":
                self.dim_type[d] = DIM_RBF
                self.lengthscale[d] = float(np.asarray(dim["lengthscale"]).reshape(-1)[0])
                m = dim.get("measure")
                if m is None:
                    self.measure[d] = MEAS_NONE
                elif m[0] == "gaussian":
                    self.measure[d] = MEAS_GAUSSIAN
                    self.meas_p0[d], self.meas_p1[d] = float(m[1]), float(m[2])
                elif m[0] == "uniform":
                    self.measure[d] = MEAS_UNIFORM
                    self.meas_p0[d], self.meas_p1[d] = float(m[1]), float(m[2])
                elif m[0] == "empirical":
                    loc = _f64(np.asarray(m[1], dtype=np.float64).reshape(-1))
                    w = _f64(np.asarray(m[2], dtype=np.float64).reshape(-1))
                    if loc.size != w.size:
                        raise ValueError("empirical measure: locations and weights differ in length")
                    self.measure[d] = MEAS_EMPIRICAL
                    self.meas_k[d], self.meas_off[d] = loc.size, off
                    data += [loc, w]
                    off += 2 * loc.size
                elif m[0] == "mog":
                    mu, var, w = (_f64(np.asarray(x, dtype=np.float64).reshape(-1)) for x in m[1:4])
                    if not (mu.size == var.size == w.size):
                        raise ValueError("MOG measure: means, variances and weights differ in length")
                    self.measure[d] = MEAS_MOG
                    self.meas_k[d], self.meas_off[d] = mu.size, off
                    data += [mu, var, w]
                    off += 3 * mu.size
                else:
                    raise NotImplementedError(f"unknown measure {m[0]!r}")
            elif t == "binary":
                self.dim_type[d] = DIM_BINARY
                self.meas_p0[d] = float(dim["p0"])
                self.meas_k[d] = 2
            elif t == "categorical":
                self.dim_type[d] = DIM_CATEGORICAL
                B, p = categorical_table_unit(dim["W"], dim["kappa"], dim["p"])
                Cn = B.shape[0]
                self.meas_k[d], self.meas_off[d] = Cn, off
                self.cat_blocks[d] = (off, Cn)
                data += [B.reshape(-1), p.reshape(-1)]
                off += Cn * Cn + Cn
            else:
                raise NotImplementedError(f"unknown sub-kernel type {t!r}")
        self.meas_data = _f64(np.concatenate(data)) if data else np.zeros(1, np.float64)
        self.meas_len = off
        s = KernelDescStruct()
        s.num_dims, s.max_depth, s.share_var, s.n_order_var = D, R, int(share), int(ov.size)
        s.order_var = _dp(self.order_var)
        s.dim_type = _ip(self.dim_type)
        s.active_col = _ip(self.active_col)
        s.lengthscale = _dp(self.lengthscale)
        s.base_var = _dp(self.base_var)
        s.measure = _ip(self.measure)
        s.meas_p0 = _dp(self.meas_p0)
        s.meas_p1 = _dp(self.meas_p1)
        s.meas_k = _ip(self.meas_k)
        s.meas_off = _ip(self.meas_off)
        s.meas_data = _dp(self.meas_data)
        s.meas_data_len = int(off)
        # default: base variances are differentiated unless every one of them is the constant 1 of a shared-variance kernel
        s.grad_base_var = int(bool(spec.get("base_var_grad", (not share) or bool(np.any(self.base_var != 1.0)))))
        self.extra_col_off = np.asarray(extra_off, dtype=np.int32)
        self.extra_cols = np.asarray(extra_cols if extra_cols else [0], dtype=np.int32)
        self.grouped = bool(extra_cols)
        if self.grouped:
            s.extra_col_off, s.extra_cols = _ip(self.extra_col_off), _ip(self.extra_cols)
        self.struct = s
        self.D, self.R, self.share = D, R, share
        self.min_cols = int(max(self.active_col.max(), max(extra_cols) if extra_cols else 0)) + 1

    @property
    def ref(self):
        return C.byref(self.struct)


def categorical_table_unit(W, kappa, p):
    """Unit-variance coregion table B = A - (Ap)(Ap)^T / (p^T A p), A = W W^T + diag(kappa)
    (oak/ortho_categorical_kernel.py:34-42) -- O(C^2) parameter preparation done on the host."""
    W = np.asarray(W, dtype=np.float64)
    kappa = np.asarray(kappa, dtype=np.float64).reshape(-1)
    p = np.asarray(p, dtype=np.float64).reshape(-1, 1)
    A = W @ W.T + np.diag(kappa)
    Ap = A @ p
    B = A - (Ap @ Ap.T) / float((p.T @ Ap)[0, 0])
    return np.ascontiguousarray(B), np.ascontiguousarray(p)


class PackedSubsets(tuple):
    """(flat int32 dim indices, int32 offsets[n + 1]) as ``HipContext.pack_subsets`` returns them; the type is the marker."""
    __slots__ = ()

    def __new__(cls, flat, off):
        flat, off = np.ascontiguousarray(flat, np.int32), np.ascontiguousarray(off, np.int32)
        if off.ndim != 1 or off.size < 1 or off[0] != 0 or np.any(np.diff(off) < 0) or (off.size > 1 and off[-1] > max(flat.size, 0)):
            raise ValueError("PackedSubsets: offsets must start at 0, be non-decreasing and end within the index array")
        return super().__new__(cls, (flat, off))


class HipContext:
    """One device context (HIP stream + device scratch); mirrors ``oak_ctx``."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = _CTX()
        _check(self._lib.oak_ctx_create(int(device), C.byref(h)))
        self._h = h
        _live_contexts.add(self)
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.oak_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------------------------
    @staticmethod
    def _check_cols(desc: KernelDesc, X: np.ndarray):
        if X.ndim != 2 or X.shape[1] < desc.min_cols:
            raise ValueError(f"input has shape {X.shape}; the kernel reads column {desc.min_cols - 1}")

    def sync(self):
        _check(self._lib.oak_sync(self._h))

    def timing(self, name: str):
        ms, cnt = C.c_double(), C.c_int32()
        _check(self._lib.oak_last_timing(self._h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def reset_timings(self):
        _check(self._lib.oak_reset_timings(self._h))

    def mem_info(self):
        f, t = C.c_double(), C.c_double()
        _check(self._lib.oak_device_mem_info(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    # -- Gram ---------------------------------------------------------------------------------
    def set_gram_form(self, form: str):
        """A/B switch of ``gram`` / ``gram_diag``: "native" (default) or "reference" -- the reference's arithmetic (GPflow's
        expanded squared distance, power sums + Newton-Girard) reproduced on the device, entry by entry."""
        code = {"native": 0, "reference": 1}.get(form)
        if code is None:
            raise ValueError("gram form must be 'native' or 'reference'")
        _check(self._lib.oak_set_gram_form(self._h, code))

    def gram(self, desc: KernelDesc, X, X2=None) -> np.ndarray:
        X = _f64(X, 2)
        self._check_cols(desc, X)
        if X2 is None:
            out = np.empty((X.shape[0], X.shape[0]))
            if out.size:
                _check(self._lib.oak_gram(self._h, desc.ref, _dp(X), X.shape[0], None, 0, X.shape[1], _dp(out)))
            return out
        X2 = _f64(X2, 2)
        if X2.shape[1] != X.shape[1]:
            raise ValueError("X and X2 differ in their number of columns")
        out = np.empty((X.shape[0], X2.shape[0]))
        if out.size:
            _check(self._lib.oak_gram(self._h, desc.ref, _dp(X), X.shape[0], _dp(X2), X2.shape[0], X.shape[1], _dp(out)))
        return out

    def gram_diag(self, desc: KernelDesc, X) -> np.ndarray:
        X = _f64(X, 2)
        self._check_cols(desc, X)
        out = np.empty(X.shape[0])
        if out.size:
            _check(self._lib.oak_gram_diag(self._h, desc.ref, _dp(X), X.shape[0], X.shape[1], _dp(out)))
        return out

    def gram_component(self, desc: KernelDesc, subset: Sequence[int], apply_order_var: bool, X, X2=None) -> np.ndarray:
        X = _f64(X, 2)
        self._check_cols(desc, X)
        sub = np.ascontiguousarray(subset, dtype=np.int32)
        n2 = X.shape[0]
        X2p = None
        if X2 is not None:
            X2 = _f64(X2, 2)
            n2, X2p = X2.shape[0], _dp(X2)
        out = np.empty((X.shape[0], n2))
        if out.size:
            _check(self._lib.oak_gram_component(self._h, desc.ref, _ip(sub), sub.size, int(apply_order_var), _dp(X),
                                                X.shape[0], X2p, 0 if X2 is None else n2, X.shape[1], _dp(out)))
        return out

    def gram_component_diag(self, desc: KernelDesc, subset: Sequence[int], apply_order_var: bool, X) -> np.ndarray:
        X = _f64(X, 2)
        self._check_cols(desc, X)
        sub = np.ascontiguousarray(subset, dtype=np.int32)
        out = np.empty(X.shape[0])
        if out.size:
            _check(self._lib.oak_gram_component_diag(self._h, desc.ref, _ip(sub), sub.size, int(apply_order_var), _dp(X),
                                                     X.shape[0], X.shape[1], _dp(out)))
        return out

    # -- SGPR ---------------------------------------------------------------------------------
    def sgpr_set_data(self, X, Y):
        X, Y = _f64(X, 2), _f64(np.asarray(Y).reshape(-1))
        if Y.shape[0] != X.shape[0]:
            raise ValueError("X and Y differ in their number of rows")
        _check(self._lib.oak_sgpr_set_data(self._h, _dp(X), _dp(Y), X.shape[0], X.shape[1]))

    def sgpr_set_targets(self, y):
        """Another target column for the rows already on the device (one output of an N x P ``Y``)."""
        y = _f64(np.asarray(y).reshape(-1))
        _check(self._lib.oak_sgpr_set_targets(self._h, _dp(y), y.shape[0]))

    def sgpr_set_extra_targets(self, Y_extra):
        """Columns 1 .. of a P-column target matrix (``Y_extra`` [N x (P - 1)], or None / zero columns to forget them): bound and
        gradient become sums over all outputs with the y-independent work shared (oak_sgpr_set_extra_targets)."""
        if Y_extra is None or np.asarray(Y_extra).size == 0:
            _check(self._lib.oak_sgpr_set_extra_targets(self._h, None, 0, 0))
            return
        Yt = np.ascontiguousarray(np.asarray(Y_extra, dtype=np.float64).T)      # one column per row
        _check(self._lib.oak_sgpr_set_extra_targets(self._h, _dp(Yt), Yt.shape[1], Yt.shape[0]))

    def sgpr_select_output(self, p: int):
        _check(self._lib.oak_sgpr_select_output(self._h, int(p)))

    def sgpr_set_inducing(self, Z):
        Z = _f64(Z, 2)
        _check(self._lib.oak_sgpr_set_inducing(self._h, _dp(Z), Z.shape[0], Z.shape[1]))

    def sgpr_set_panel_rows(self, rows: int):
        _check(self._lib.oak_sgpr_set_panel_rows(self._h, int(rows)))

    ROUTES = {"auto": 0, "phi": 1, "whitened": 2}

    def sgpr_set_route(self, route):
        _check(self._lib.oak_sgpr_set_route(self._h, self.ROUTES.get(route, route)))

    PRECISIONS = {"auto": -1, "fp64": 0, "fp32": 1, "int8crt": 2}

    def sgpr_set_precision(self, mode):
        """'auto' (default: 'int8crt' on large phi-route problems, the fp64 kernels elsewhere), 'fp64' (fp64 kernels throughout), 'fp32' = fp32 Kfu panel + fp32-MFMA Phi partials (forward, phi route), or
        'int8crt' = Phi accumulated EXACTLY on the int8 matrix pipe from 48-bit scaled integers (residue planes + Chinese
        remainder reconstruction, csrc/crt.hip; phi route): at least as accurate as the fp64 accumulation, not a lower precision."""
        _check(self._lib.oak_sgpr_set_precision(self._h, self.PRECISIONS.get(mode, mode)))

    def sgpr_stats_precision(self) -> str:
        """'fp32' when the last statistics really came from the fp32 panel path (the mode falls back to fp64 on an
        ill-conditioned Kuu, on the whitened route and for gradient calls)."""
        f = C.c_int32()
        _check(self._lib.oak_sgpr_stats_precision(self._h, C.byref(f)))
        return {0: "fp64", 1: "fp32", 2: "int8crt"}[f.value]

    def sgpr_set_global_rows(self, n_total: int):
        """Rows over all shards (0 = unknown): what the auto route's size rule looks at when this context holds one shard."""
        _check(self._lib.oak_sgpr_set_global_rows(self._h, int(n_total)))

    def sgpr_stats_whitened(self) -> bool:
        f = C.c_int32()
        _check(self._lib.oak_sgpr_stats_whitened(self._h, C.byref(f)))
        return bool(f.value)

    def sgpr_local_stats(self, desc: KernelDesc, jitter: float = 1e-6):
        _check(self._lib.oak_sgpr_local_stats(self._h, desc.ref, float(jitter)))

    def sgpr_get_stats(self) -> np.ndarray:
        out = np.empty(self._lib.oak_sgpr_stats_len(self._h))
        _check(self._lib.oak_sgpr_get_stats(self._h, _dp(out)))
        return out

    def sgpr_set_stats(self, packed, whitened: bool = False):
        packed = _f64(packed, 1)
        if packed.size != self._lib.oak_sgpr_stats_len(self._h):
            raise ValueError("packed statistics have the wrong length")
        _check(self._lib.oak_sgpr_set_stats(self._h, _dp(packed), int(whitened)))

    def sgpr_tail(self, desc: KernelDesc, noise_var: float, jitter: float = 1e-6):
        e = C.c_double()
        terms = np.zeros(8)
        _check(self._lib.oak_sgpr_tail(self._h, desc.ref, float(noise_var), float(jitter), C.byref(e), _dp(terms)))
        return e.value, terms

    def sgpr_elbo(self, desc: KernelDesc, noise_var: float, jitter: float = 1e-6) -> float:
        e = C.c_double()
        _check(self._lib.oak_sgpr_elbo(self._h, desc.ref, float(noise_var), float(jitter), C.byref(e)))
        return e.value

    TERM_NAMES = ("sum_log_diag_LB", "cTc", "tr_AAT", "kappa", "yy", "n_rows", "logdet_Kuu", "cond_estimate")

    def sgpr_last_terms(self) -> dict:
        """Named pieces of the bound from the most recent tail (any entry point)."""
        t = np.zeros(8)
        _check(self._lib.oak_sgpr_last_terms(self._h, _dp(t)))
        return dict(zip(self.TERM_NAMES, t[:8].tolist()))

    def sgpr_alpha(self, M: int) -> np.ndarray:
        out = np.empty(M)
        _check(self._lib.oak_sgpr_alpha(self._h, _dp(out)))
        return out

    def sgpr_predict(self, desc: KernelDesc, Xs):
        Xs = _f64(Xs, 2)
        mean, var = np.empty(Xs.shape[0]), np.empty(Xs.shape[0])
        _check(self._lib.oak_sgpr_predict(self._h, desc.ref, _dp(Xs), Xs.shape[0], Xs.shape[1], _dp(mean), _dp(var)))
        return mean, var

    def grad_len(self, desc: KernelDesc) -> int:
        return int(self._lib.oak_grad_len(desc.ref))

    def sgpr_effective_L(self, M: int) -> np.ndarray:
        out = np.empty((int(M), int(M)))
        _check(self._lib.oak_sgpr_effective_L(self._h, _dp(out)))
        return out

    def gpr_chol(self, N: int) -> np.ndarray:
        out = np.empty((int(N), int(N)))
        _check(self._lib.oak_gpr_chol(self._h, _dp(out)))
        return out

    def sgpr_elbo_grad_z(self, desc: KernelDesc, noise_var: float, M: int, ldx: int, jitter: float = 1e-6):
        """(elbo, grad, gradZ [M, ldx]): as sgpr_elbo_grad plus the gradient w.r.t. the inducing inputs."""
        e = C.c_double()
        g = np.empty(self.grad_len(desc))
        gz = np.empty((int(M), int(ldx)))
        _check(self._lib.oak_sgpr_elbo_grad_z(self._h, desc.ref, float(noise_var), float(jitter), C.byref(e), _dp(g), _dp(gz)))
        return e.value, g, gz

    def sgpr_elbo_grad(self, desc: KernelDesc, noise_var: float, jitter: float = 1e-6):
        e = C.c_double()
        g = np.zeros(self.grad_len(desc))
        _check(self._lib.oak_sgpr_elbo_grad(self._h, desc.ref, float(noise_var), float(jitter), C.byref(e), _dp(g)))
        return e.value, g

    # -- GPR ----------------------------------------------------------------------------------
    def gpr_set_data(self, X, Y):
        X, Y = _f64(X, 2), _f64(np.asarray(Y).reshape(-1))
        if Y.shape[0] != X.shape[0]:
            raise ValueError("X and Y differ in their number of rows")
        _check(self._lib.oak_gpr_set_data(self._h, _dp(X), _dp(Y), X.shape[0], X.shape[1]))

    def gpr_set_targets(self, y):
        y = _f64(np.asarray(y).reshape(-1))
        _check(self._lib.oak_gpr_set_targets(self._h, _dp(y), y.shape[0]))

    def gpr_log_marginal(self, desc: KernelDesc, noise_var: float) -> float:
        e = C.c_double()
        _check(self._lib.oak_gpr_log_marginal(self._h, desc.ref, float(noise_var), C.byref(e)))
        return e.value

    def gpr_log_marginal_grad(self, desc: KernelDesc, noise_var: float):
        e = C.c_double()
        g = np.zeros(self.grad_len(desc))
        _check(self._lib.oak_gpr_log_marginal_grad(self._h, desc.ref, float(noise_var), C.byref(e), _dp(g)))
        return e.value, g

    def gpr_alpha(self, N: int) -> np.ndarray:
        out = np.empty(N)
        _check(self._lib.oak_gpr_alpha(self._h, _dp(out)))
        return out

    def gpr_predict(self, desc: KernelDesc, Xs):
        Xs = _f64(Xs, 2)
        mean, var = np.empty(Xs.shape[0]), np.empty(Xs.shape[0])
        _check(self._lib.oak_gpr_predict(self._h, desc.ref, _dp(Xs), Xs.shape[0], Xs.shape[1], _dp(mean), _dp(var)))
        return mean, var

    # -- SVGP (whitened, diagonal q, Bernoulli) -----------------------------------------------------
    LINKS = {"logit": 0, "probit": 1}

    @staticmethod
    def _gh(n_gh: int):
        x, w = np.polynomial.hermite.hermgauss(int(n_gh))
        return np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(w, dtype=np.float64)

    def svgp_elbo(self, desc: KernelDesc, q_mu, q_sqrt, link: str = "logit", link_eps: float = 1e-3, jitter: float = 1e-6,
                  n_gh: int = 20, grad: bool = False):
        """elbo, or with ``grad`` (elbo, grad [grad_len], d/d q_mu [M], d/d q_sqrt [M])."""
        q_mu, q_sqrt = _f64(q_mu, 1), _f64(q_sqrt, 1)
        x, w = self._gh(n_gh)
        e = C.c_double()
        if not grad:
            _check(self._lib.oak_svgp_elbo_grad(self._h, desc.ref, _dp(q_mu), _dp(q_sqrt), float(jitter), _dp(x), _dp(w), len(x),
                                                self.LINKS[link], float(link_eps), C.byref(e), None, None, None))
            return e.value
        g, gm, gs = np.zeros(self.grad_len(desc)), np.empty(q_mu.size), np.empty(q_mu.size)
        _check(self._lib.oak_svgp_elbo_grad(self._h, desc.ref, _dp(q_mu), _dp(q_sqrt), float(jitter), _dp(x), _dp(w), len(x),
                                            self.LINKS[link], float(link_eps), C.byref(e), _dp(g), _dp(gm), _dp(gs)))
        return e.value, g, gm, gs

    def svgp_predict(self, desc: KernelDesc, q_mu, q_sqrt, Xs, Ys=None, link: str = "logit", link_eps: float = 1e-3,
                     jitter: float = 1e-6, n_gh: int = 20):
        """(mean, var), or with ``Ys`` (mean, var, log predictive density)."""
        q_mu, q_sqrt, Xs = _f64(q_mu, 1), _f64(q_sqrt, 1), _f64(Xs, 2)
        mean, var = np.empty(Xs.shape[0]), np.empty(Xs.shape[0])
        x, w = self._gh(n_gh)
        if Ys is None:
            _check(self._lib.oak_svgp_predict(self._h, desc.ref, _dp(q_mu), _dp(q_sqrt), float(jitter), _dp(Xs), Xs.shape[0],
                                              Xs.shape[1], _dp(mean), _dp(var), None, None, _dp(x), _dp(w), len(x),
                                              self.LINKS[link], float(link_eps)))
            return mean, var
        Ys = _f64(Ys, 1)
        if Ys.size != Xs.shape[0]:
            raise ValueError("Ys must hold one label per row of Xs")
        ld = np.empty(Xs.shape[0])
        _check(self._lib.oak_svgp_predict(self._h, desc.ref, _dp(q_mu), _dp(q_sqrt), float(jitter), _dp(Xs), Xs.shape[0],
                                          Xs.shape[1], _dp(mean), _dp(var), _dp(Ys), _dp(ld), _dp(x), _dp(w), len(x),
                                          self.LINKS[link], float(link_eps)))
        return mean, var, ld

    def svgp_posterior(self, desc: KernelDesc, q_mu, q_sqrt, jitter: float = 1e-6, get_L: bool = True):
        q_mu, q_sqrt = _f64(q_mu, 1), _f64(q_sqrt, 1)
        alpha = np.empty(q_mu.size)
        L = np.empty((q_mu.size, q_mu.size)) if get_L else None
        _check(self._lib.oak_svgp_posterior(self._h, desc.ref, _dp(q_mu), _dp(q_sqrt), float(jitter), _dp(alpha),
                                            _dp(L) if get_L else None))
        return (alpha, L) if get_L else alpha

    # -- Sobol / components -------------------------------------------------------------------
    @staticmethod
    def _pack_subsets(subsets):
        """(flat dim indices, offsets[len + 1]) of a list of subsets.  A caller that evaluates the same term list repeatedly
        can pack once (``pack_subsets`` returns a ``PackedSubsets``) and pass that: walking 41 448 Python lists costs more than
        the device pass.  Only a ``PackedSubsets`` is taken as already packed -- a plain tuple of two arrays is two subsets."""
        if isinstance(subsets, PackedSubsets):
            return subsets
        n = len(subsets)
        lens = np.fromiter(map(len, subsets), dtype=np.int64, count=n)
        off = np.zeros(n + 1, np.int32)
        np.cumsum(lens, out=off[1:])
        total = int(off[-1])
        flat = np.fromiter(itertools.chain.from_iterable(subsets), dtype=np.int32, count=total) if total else np.zeros(1, np.int32)
        return PackedSubsets(np.ascontiguousarray(flat), off)

    pack_subsets = _pack_subsets

    SOBOL_PATHS = {"auto": 0, "terms": 1, "gram": 2}

    def sobol(self, desc: KernelDesc, Xc, alpha, subsets, use_order_var=True, delta=1.0, mu=0.0, collective=False) -> np.ndarray:
        """alpha^T (prod_{d in S} L_d) alpha for every subset S.  ``collective``: every rank of the attached communicator makes
        this call with the same arguments (work sharded on the device side, every rank gets every term)."""
        Xc, alpha = _f64(Xc, 2), _f64(np.asarray(alpha).reshape(-1))
        flat, off = self._pack_subsets(subsets)
        out = np.zeros(len(off) - 1)
        fn = self._lib.oak_sobol_collective if collective else self._lib.oak_sobol
        _check(fn(self._h, desc.ref, _dp(Xc), Xc.shape[0], Xc.shape[1], _dp(alpha), _ip(flat), _ip(off),
                  len(out), int(use_order_var), float(delta), float(mu), _dp(out)))
        return out

    def sobol_set_path(self, path="auto") -> None:
        """'auto' (cost model), 'terms' (one workgroup per term) or 'gram' (Gram of products on the matrix pipe)."""
        _check(self._lib.oak_sobol_set_path(self._h, self.SOBOL_PATHS[path] if isinstance(path, str) else int(path)))

    def sobol_last_info(self) -> dict:
        info = np.zeros(4)
        _check(self._lib.oak_sobol_last_info(self._h, _dp(info)))
        return dict(path={1: "terms", 2: "gram"}.get(int(info[0]), "none"), columns=int(info[1]), pairing_disagreement=float(info[2]),
                    pair_rows=int(info[3]))

    def gram_f32(self, desc: KernelDesc, X1, X2) -> np.ndarray:
        """The fp32 Kuf panel of the fp32 statistics mode (parity checks of that mode)."""
        X1, X2 = _f64(X1, 2), _f64(X2, 2)
        self._check_cols(desc, X1); self._check_cols(desc, X2)
        out = np.empty((X1.shape[0], X2.shape[0]), dtype=np.float32)
        _check(self._lib.oak_gram_f32(self._h, desc.ref, _dp(X1), X1.shape[0], _dp(X2), X2.shape[0], X1.shape[1],
                                      out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def sobol_L(self, desc: KernelDesc, dim: int, v: float, delta: float, mu: float, Xc) -> np.ndarray:
        Xc = _f64(Xc, 2)
        out = np.empty((Xc.shape[0], Xc.shape[0]))
        _check(self._lib.oak_sobol_L(self._h, desc.ref, int(dim), float(v), float(delta), float(mu), _dp(Xc), Xc.shape[0],
                                     Xc.shape[1], _dp(out)))
        return out

    def measure_cov(self, desc: KernelDesc, dim: int, X):
        X = _f64(X, 2)
        c = np.empty(X.shape[0])
        v = C.c_double()
        _check(self._lib.oak_cov_x_s(self._h, desc.ref, int(dim), _dp(X), X.shape[0], X.shape[1], _dp(c), C.byref(v)))
        return c, v.value

    def additive_terms(self, mats, R: int) -> np.ndarray:
        mats = _f64(mats, 2)
        out = np.empty((R + 1, mats.shape[1]))
        _check(self._lib.oak_additive_terms(self._h, _dp(mats), mats.shape[0], mats.shape[1], int(R), _dp(out)))
        return out

    def component_predict(self, desc: KernelDesc, Xs, Xc, alpha, subsets, use_order_var=True) -> np.ndarray:
        Xs, Xc, alpha = _f64(Xs, 2), _f64(Xc, 2), _f64(np.asarray(alpha).reshape(-1))
        flat, off = self._pack_subsets(subsets)
        out = np.zeros((len(subsets), Xs.shape[0]))
        _check(self._lib.oak_component_predict(self._h, desc.ref, _dp(Xs), Xs.shape[0], _dp(Xc), Xc.shape[0], Xs.shape[1],
                                               _dp(alpha), _ip(flat), _ip(off), len(subsets), int(use_order_var), _dp(out)))
        return out

    # -- multi-GPU ----------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(load_library().oak_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        _check(self._lib.oak_comm_init(self._h, unique_id, int(nranks), int(rank)))
        self._comm_rank = int(rank)

    def comm_init_loopback(self, nranks: int):
        """Test communicator: `nranks` identical ranks (every all-reduce multiplies by nranks)."""
        _check(self._lib.oak_comm_init_loopback(self._h, int(nranks)))

    def comm_init_host(self, nranks: int, rank: int, allreduce):
        """Host-exchange communicator: ``allreduce(a)`` must return the sum over all ranks of the 1-D float64 array ``a``
        (same result on every rank).  Every collective of the library then goes through it on a host copy of the buffer."""
        def _cb(buf, n, _user):
            try:
                a = np.ctypeslib.as_array(buf, shape=(int(n),))
                a[:] = np.asarray(allreduce(a.copy()), dtype=np.float64).reshape(-1)
                return 0
            except Exception as ex:                                     # noqa: BLE001  (must not propagate into C)
                self._host_comm_error = ex
                return 1
        self._host_cb = _HOST_ALLREDUCE(_cb)                            # keep the trampoline alive as long as the context
        _check(self._lib.oak_comm_init_host(self._h, int(nranks), int(rank), C.cast(self._host_cb, C.c_void_p), None))
        self._comm_rank = int(rank)

    @staticmethod
    def comm_info() -> dict:
        """Path and version of the librccl.so the library loaded, and the version of the header it was compiled against."""
        path = C.create_string_buffer(1024)
        v, hv = C.c_int32(), C.c_int32()
        _check(load_library().oak_comm_info(path, 1024, C.byref(v), C.byref(hv)))
        return {"path": path.value.decode(), "version": int(v.value), "header_version": int(hv.value)}

    def comm_allgatherv(self, local: np.ndarray, counts) -> np.ndarray:
        """Concatenation, in rank order, of every rank's 1-D block (``counts[r]`` doubles from rank r).  ``len(counts)`` must
        be the size of the context's communicator (the library refuses anything else: a block nobody delivers would come
        back as zeros)."""
        local = _f64(np.asarray(local).reshape(-1), 1)
        counts = np.ascontiguousarray([int(c) for c in counts], dtype=np.int64)
        rank = self.comm_rank()
        if rank >= len(counts) or local.size != counts[rank]:
            raise ValueError("comm_allgatherv: the local block does not have the length this rank announced")
        total, off = int(counts.sum()), int(counts[:rank].sum())
        buf = np.zeros(max(total, 1))
        buf[off:off + local.size] = local
        _check(self._lib.oak_comm_allgatherv(self._h, _dp(buf), counts.ctypes.data_as(C.POINTER(C.c_int64)), len(counts)))
        return buf[:total]

    def comm_rank(self) -> int:
        return getattr(self, "_comm_rank", 0)

    def comm_destroy(self):
        _check(self._lib.oak_comm_destroy(self._h))
        self._host_cb = None

    def comm_allreduce_stats(self):
        _check(self._lib.oak_comm_allreduce_stats(self._h))

    def comm_allreduce_host(self, buf: np.ndarray):
        buf = _f64(buf, 1)
        _check(self._lib.oak_comm_allreduce_host(self._h, _dp(buf), buf.size))
        return buf

    # -- input preprocessing ---------------------------------------------------------------------
    def flow_objective(self, g: Optional[np.ndarray], n: int, use_log: bool, scale: float, shift: float, skewness: float,
                       tailweight: float):
        """KL objective of the normalising flow and its gradient w.r.t. (scale, shift, skewness, tailweight).
        ``g`` (the sample after the optional log) is uploaded when given; ``None`` evaluates on the resident copy."""
        obj = C.c_double()
        grad = np.empty(4)
        gp = None if g is None else _dp(_f64(g, 1))
        _check(self._lib.oak_flow_objective(self._h, gp, int(n), 1 if use_log else 0, float(scale), float(shift), float(skewness),
                                            float(tailweight), C.byref(obj), _dp(grad)))
        return obj.value, grad

    def flow_forward(self, X: np.ndarray, kind, params) -> np.ndarray:
        """Column-wise transform of X [N, D]: kind[d] in {0 copy, 1 flow, 2 flow after log(x - offset), 3 affine};
        params [D, 5] = (offset, scale, shift, skewness, tailweight) or (mean, std, 0, 0, 0)."""
        X = _f64(X, 2)
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        params = _f64(params, 2)
        if kind.shape != (X.shape[1],) or params.shape != (X.shape[1], 5):
            raise ValueError("kind must have D entries and params shape [D, 5]")
        out = np.empty_like(X)
        _check(self._lib.oak_flow_forward(self._h, _dp(X), X.shape[0], X.shape[1], X.shape[1], _ip(kind), _dp(params), _dp(out)))
        return out

    # -- inducing-point initialisation ----------------------------------------------------------
    def kmeans(self, X: np.ndarray, init_centres: np.ndarray, max_iter: int = 300, tol: float = 0.0):
        """Lloyd iterations from ``init_centres`` (scikit-learn's single-run loop; ``tol`` is absolute).
        Returns (centres [K, D], labels [N] int32, inertia, n_iter)."""
        X = _f64(X, 2)
        C0 = _f64(init_centres, 2)
        if C0.shape[1] != X.shape[1]:
            raise ValueError("init_centres must have the same number of columns as X")
        K = C0.shape[0]
        centres = np.empty_like(C0)
        labels = np.empty(X.shape[0], dtype=np.int32)
        inertia = C.c_double()
        n_iter = C.c_int32()
        _check(self._lib.oak_kmeans(self._h, _dp(X), X.shape[0], X.shape[1], X.shape[1], K, _dp(C0), int(max_iter), float(tol),
                                    _dp(centres), _ip(labels), C.byref(inertia), C.byref(n_iter)))
        return centres, labels, inertia.value, n_iter.value

    def kmeans_plusplus(self, X: np.ndarray, n_clusters: int, random_state=None):
        """scikit-learn's greedy k-means++ (``sklearn.cluster.kmeans_plusplus``) on the device.  ``random_state`` is an int
        seed or a ``numpy.random.RandomState``; the draws are made in scikit-learn's order, so the same seed gives the same
        picks.  Returns (centres [K, D], indices [K])."""
        X = _f64(X, 2)
        N, K = X.shape[0], int(n_clusters)
        rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
        n_trials = 2 + int(np.log(K))
        first = int(rs.choice(N, p=np.full(N, 1.0 / N)))
        U = np.ascontiguousarray(rs.uniform(size=(max(K - 1, 0), n_trials))) if K > 1 else np.zeros((1, n_trials))
        centres = np.empty((K, X.shape[1]))
        idx = np.empty(K, dtype=np.int64)
        _check(self._lib.oak_kmeans_plusplus(self._h, _dp(X), N, X.shape[1], X.shape[1], K, first, _dp(U), n_trials, _dp(centres),
                                             idx.ctypes.data_as(C.POINTER(C.c_int64))))
        return centres, idx

    # -- benchmarking -------------------------------------------------------------------------
    def bench_gram_resident(self, desc: KernelDesc) -> float:
        b = C.c_double()
        _check(self._lib.oak_bench_gram_resident(self._h, desc.ref, C.byref(b)))
        return b.value


    def bench_crt_info(self) -> dict:
        """Shape of the most recent int8 CRT accumulation of Phi (all zero when the fp64 / fp32 kernels formed the statistics)."""
        a = (C.c_int64 * 6)()
        _check(self._lib.oak_bench_crt_info(self._h, a))
        d = dict(zip(("planes", "bits", "row_splits", "rows_per_split", "fused", "plane_columns"), [int(v) for v in a]))
        d["tail_dd"] = (d["fused"] >> 1) & 1          # the most recent tail whitened Phi in double-double arithmetic (csrc/ddgemm.hip)
        d["gemm_planes"] = (d["fused"] >> 8) & 0xff   # the most recent gradient call formed its adjoint panel on the int8 pipe (csrc/crt_gemm.hip) from
        d["gemm_bits"] = (d["fused"] >> 16) & 0xff    # this many residue planes, H scaled to this many bits (0 / 0: the fp64 GEMM ran)
        d["fused"] &= 1
        return d

    def bench_potrf(self, n: int, reps: int = 10):
        """(mean ms per factorisation, log det) of the library's Cholesky on an n x n exponential-kernel test matrix."""
        ms, ld = C.c_double(), C.c_double()
        _check(self._lib.oak_bench_potrf(self._h, int(n), int(reps), C.byref(ms), C.byref(ld)))
        return ms.value, ld.value

    def bench_trsm(self, L: np.ndarray, B: np.ndarray, trans: bool = False, reps: int = 3):
        """(X, mean ms): rows of X solve L x = b (or L^T x = b) for the rows b of B -- the library's many-row triangular solve."""
        L = np.ascontiguousarray(L, dtype=np.float64); X = np.array(B, dtype=np.float64, order="C", copy=True)
        n = L.shape[0]
        if L.shape != (n, n) or X.ndim != 2 or X.shape[1] != n:
            raise ValueError("L must be n x n and B nrhs x n")
        ms = C.c_double()
        _check(self._lib.oak_bench_trsm(self._h, _dp(L), n, _dp(X), X.shape[0], int(bool(trans)), int(reps), C.byref(ms)))
        return X, ms.value


_default_ctx: Optional[HipContext] = None


def default_context() -> HipContext:
    """Process-wide context on device ``OAK_HIP_DEVICE`` (default: LOCAL_RANK or 0)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("OAK_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = HipContext(dev)
    return _default_ctx


def debug_state() -> str:
    """What every live context last enqueued and whether its streams have drained (for watchdogs on another thread)."""
    buf = C.create_string_buffer(1 << 16)
    load_library().oak_debug_state(buf, len(buf))
    return buf.value.decode(errors="replace")


def device_count() -> int:
    n = C.c_int()
    st = load_library().oak_device_count(C.byref(n))
    return n.value if st == OAK_OK else 0

"""ctypes binding of include/adaprox_hip.h.

The shared library is the product: if it is missing this module raises — there is no
CPU fallback anywhere in this package (oracle/ is test infrastructure and is never
imported from here).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libadaprox_hip.so")

# ---- enums (include/adaprox_hip.h) ---------------------------------------------------
APX_OK, APX_ERR_INVALID, APX_ERR_HIP, APX_ERR_UNSUPPORTED, APX_ERR_DOMAIN, APX_ERR_CALLBACK, APX_ERR_COMM = range(7)
APX_ABI_VERSION = 5
APX_COMM_ID_BYTES = 128
APX_FLAG_TINY_GAMMA, APX_FLAG_NAN_GAMMA, APX_FLAG_DOMAIN, APX_FLAG_TEAM_FALLBACK = 1, 2, 4, 8
APX_F64, APX_F32 = 0, 1
APX_LAYOUT_COLS, APX_LAYOUT_ROWS = 0, 1
APX_F_ZERO, APX_F_LEAST_SQUARES, APX_F_LOGISTIC, APX_F_CALLBACK, APX_F_QUADRATIC = range(5)
APX_G_ZERO, APX_G_L1, APX_G_BOX, APX_G_BOXV, APX_G_INDZERO, APX_G_L2 = range(6)
APX_RULE_FIXED, APX_RULE_MM, APX_RULE_OURS, APX_RULE_OURS_PLUS = range(4)
APX_KSTAT_GEMV_N, APX_KSTAT_GEMV_T, APX_KSTAT_ONE_PASS, APX_KSTAT_COUNT = 0, 1, 2, 3
(APX_OPT_MULTI_ROWS_MASK, APX_OPT_SLAB_GENERATION, APX_OPT_TEAM_WAIT_MS, APX_OPT_TEAM_COOPERATIVE, APX_OPT_TEAM_FAULT_GRID,
 APX_OPT_GEMV_N_VARIANT, APX_OPT_GEMV_T_VARIANT, APX_OPT_GEMV_N_BLOCKS, APX_OPT_GEMV_T_BLOCKS, APX_OPT_FUSED_TAIL,
 APX_OPT_MULTI_WAVES, APX_OPT_TEAM_MEMBERS, APX_OPT_TEAM_SLOTS, APX_OPT_ROW_DEAL, APX_OPT_TEAM_INTERLEAVE, APX_OPT_TEAM_PD_DEFER,
 APX_OPT_FAULT_SETUP, APX_OPT_SPARSE_WINDOWS, APX_OPT_ONE_PASS_MIN_ROWS) = range(1, 20)

c_void_p, c_int32, c_int64, c_double = C.c_void_p, C.c_int32, C.c_int64, C.c_double


class ProxDesc(C.Structure):
    _fields_ = [("kind", c_int32), ("_pad", c_int32), ("lam", c_double), ("lb", c_double), ("ub", c_double),
                ("lb_vec", c_void_p), ("ub_vec", c_void_p), ("shift", c_void_p)]


EVAL_FN = C.CFUNCTYPE(c_int32, c_void_p, c_void_p, c_int64, C.POINTER(c_double))
PULLBACK_FN = C.CFUNCTYPE(c_int32, c_void_p, c_void_p, c_int64)
ALLREDUCE_FN = C.CFUNCTYPE(c_int32, c_void_p, c_void_p, c_int64, c_int32, c_void_p)


class ProblemDesc(C.Structure):
    _fields_ = [("f_kind", c_int32), ("dtype", c_int32), ("n", c_int64), ("m", c_int64),
                ("A_f", c_void_p), ("b", c_void_p), ("n_total_rows", c_int64),
                ("eval_cb", EVAL_FN), ("pullback_cb", PULLBACK_FN), ("cb_user", c_void_p),
                ("g", ProxDesc), ("h", ProxDesc), ("A_lin", c_void_p),
                ("allreduce", ALLREDUCE_FN), ("allreduce_user", c_void_p), ("shard_mask", c_int32), ("_pad2", c_int32),
                ("comm", c_void_p)]


class Rule(C.Structure):
    _fields_ = [("kind", c_int32), ("_pad", c_int32), ("gamma", c_double), ("t", c_double),
                ("norm_A", c_double), ("delta", c_double), ("Theta", c_double),
                ("xi", c_double), ("nu", c_double), ("r", c_double)]


class Record(C.Structure):
    _fields_ = [("it", c_int64), ("gamma", c_double), ("sigma", c_double), ("norm_res", c_double),
                ("objective", c_double), ("f_x", c_double), ("f_evals", c_int64), ("grad_evals", c_int64)]


class Result(C.Structure):
    _fields_ = [("numit", c_int64), ("converged", c_int32), ("flags", c_int32),
                ("gamma", c_double), ("sigma", c_double), ("norm_res", c_double), ("objective", c_double),
                ("f_evals", c_int64), ("grad_f_evals", c_int64), ("prox_g_evals", c_int64),
                ("prox_h_evals", c_int64), ("A_evals", c_int64), ("At_evals", c_int64), ("n_records", c_int64)]


class TeamWaitSlot(C.Structure):
    _fields_ = [("waits", c_int64), ("long_waits", c_int64), ("max_polls", c_int64), ("max_wait_ns", c_int64),
                ("at_launch", c_int64), ("at_step", c_int64), ("at_clock_ns", c_int64),
                ("block", c_int32), ("team", c_int32), ("member", c_int32), ("xcc", c_int32), ("kind", c_int32), ("_pad", c_int32)]


class SolveOpts(C.Structure):
    _fields_ = [("records", C.POINTER(Record)), ("record_cap", c_int64), ("use_graph", c_int32),
                ("poll_every", c_int32)]


class LinesearchParams(C.Structure):
    _fields_ = [("gamma", c_double), ("eta", c_double), ("t", c_double), ("delta", c_double), ("Theta", c_double),
                ("r", c_double), ("R", c_double), ("max_linesearch", c_int64)]


class KernelStat(C.Structure):
    _fields_ = [("launches", c_int64), ("total_ms", c_double), ("bytes", c_double)]


# name -> (restype, argtypes); must list every symbol the header declares
_P = C.POINTER
PROTOTYPES = {
    "apx_abi_version": (c_int32, []),
    "apx_last_error": (C.c_char_p, []),
    "apx_ctx_create": (c_int32, [c_int32, c_void_p, _P(c_void_p)]),
    "apx_ctx_destroy": (c_int32, [c_void_p]),
    "apx_ctx_sync": (c_int32, [c_void_p]),
    "apx_ctx_stream": (c_int32, [c_void_p, _P(c_void_p)]),
    "apx_ctx_profile": (c_int32, [c_void_p, c_int32]),
    "apx_ctx_kernel_stats": (c_int32, [c_void_p, _P(KernelStat), c_int32]),
    "apx_ctx_set_gemv_config": (c_int32, [c_void_p, c_int32, c_int32, c_int64, c_int64]),
    "apx_ctx_exploit_sparsity": (c_int32, [c_void_p, c_int32]),
    "apx_ctx_fused_small": (c_int32, [c_void_p, c_int32]),
    "apx_ctx_one_pass": (c_int32, [c_void_p, c_int32]),
    "apx_ctx_kernel_name": (C.c_char_p, [c_void_p, c_int32]),
    "apx_ctx_set_option": (c_int32, [c_void_p, c_int32, c_int64]),
    "apx_ctx_get_option": (c_int32, [c_void_p, c_int32, _P(c_int64)]),
    "apx_last_incident": (C.c_char_p, []),
    "apx_problem_team_waits": (c_int32, [c_void_p, _P(c_int64)]),
    "apx_problem_team_wait_slots": (c_int32, [c_void_p, _P(TeamWaitSlot), c_int32, _P(c_int32)]),
    "apx_ctx_one_pass_threshold": (c_int32, [c_void_p, c_int64]),
    "apx_ctx_auto_transpose": (c_int32, [c_void_p, c_int32]),
    "apx_comm_unique_id": (c_int32, [c_void_p]),
    "apx_comm_init_rank": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, _P(c_void_p)]),
    "apx_comm_init_all": (c_int32, [_P(c_void_p), c_int32, _P(c_void_p)]),
    "apx_comm_info": (c_int32, [c_void_p, _P(c_int32), _P(c_int32)]),
    "apx_comm_allreduce_host": (c_int32, [c_void_p, _P(c_double), c_int64]),
    "apx_comm_destroy": (c_int32, [c_void_p]),
    "apx_comm_library": (C.c_char_p, []),
    "apx_adaptive_primal_dual_group": (c_int32, [c_int32, _P(c_void_p), _P(Rule), c_void_p, _P(c_void_p), c_double,
                                                 c_int64, _P(SolveOpts), c_void_p, _P(c_void_p), _P(Result)]),
    "apx_gemv_variant_count": (c_int32, [c_int32]),
    "apx_gemv_variant_name": (C.c_char_p, [c_int32, c_int32]),
    "apx_ctx_device_clock": (c_int32, [c_void_p, _P(c_int64), _P(c_int64), _P(c_int64)]),
    "apx_matrix_create": (c_int32, [c_void_p, c_int64, c_int64, c_int32, _P(c_void_p)]),
    "apx_matrix_upload": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, _P(c_void_p)]),
    "apx_matrix_create_layout": (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_int32, _P(c_void_p)]),
    "apx_matrix_upload_layout": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32, _P(c_void_p)]),
    "apx_matrix_drop_primary": (c_int32, [c_void_p]),
    "apx_matrix_layout": (c_int32, [c_void_p, _P(c_int32)]),
    "apx_matrix_upload_sparse": (c_int32, [c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                           c_int32, _P(c_void_p)]),
    "apx_matrix_build_transpose": (c_int32, [c_void_p]),
    "apx_matrix_drop_transpose": (c_int32, [c_void_p]),
    "apx_matrix_download": (c_int32, [c_void_p, c_void_p, c_int64]),
    "apx_matrix_free": (c_int32, [c_void_p]),
    "apx_matrix_shape": (c_int32, [c_void_p, _P(c_int64), _P(c_int64), _P(c_int32)]),
    "apx_matrix_fill_uniform": (c_int32, [c_void_p, C.c_uint64, c_double, c_double, c_int64, c_int64]),
    "apx_matrix_scale_columns": (c_int32, [c_void_p, c_void_p]),
    "apx_matrix_fill_column": (c_int32, [c_void_p, c_int64, c_double]),
    "apx_matrix_sumsq": (c_int32, [c_void_p, _P(c_double)]),
    "apx_gemv_n": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "apx_gemv_t": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "apx_problem_create": (c_int32, [c_void_p, _P(ProblemDesc), _P(c_void_p)]),
    "apx_problem_destroy": (c_int32, [c_void_p]),
    "apx_eval": (c_int32, [c_void_p, c_void_p, _P(c_double)]),
    "apx_pullback": (c_int32, [c_void_p, c_void_p]),
    "apx_prox": (c_int32, [c_void_p, _P(ProxDesc), c_int32, c_int32, c_void_p, c_int64, c_double, c_void_p,
                           _P(c_double)]),
    "apx_prox_value": (c_int32, [c_void_p, _P(ProxDesc), c_int32, c_void_p, c_int64, _P(c_double)]),
    "apx_adaptive_primal_dual": (c_int32, [c_void_p, _P(Rule), c_void_p, c_void_p, c_double, c_int64,
                                           _P(SolveOpts), c_void_p, c_void_p, _P(Result)]),
    "apx_adaptive_proxgrad": (c_int32, [c_void_p, _P(Rule), c_void_p, c_double, c_int64, _P(SolveOpts), c_void_p,
                                        _P(Result)]),
    "apx_fixed_proxgrad": (c_int32, [c_void_p, c_double, c_void_p, c_double, c_int64, _P(SolveOpts), c_void_p,
                                     _P(Result)]),
    "apx_condat_vu": (c_int32, [c_void_p, c_double, c_double, c_void_p, c_void_p, c_double, c_int64, _P(SolveOpts),
                                c_void_p, c_void_p, _P(Result)]),
    "apx_backtracking_proxgrad": (c_int32, [c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_int64,
                                            _P(SolveOpts), c_void_p, _P(Result)]),
    "apx_adaptive_linesearch_primal_dual": (c_int32, [c_void_p, _P(LinesearchParams), c_void_p, c_void_p, c_double,
                                                      c_int64, _P(SolveOpts), c_void_p, c_void_p, _P(Result)]),
    "apx_backtracking_nesterov": (c_int32, [c_void_p, c_void_p, c_double, c_double, c_double, c_int64, _P(SolveOpts),
                                            c_void_p, _P(Result)]),
    "apx_fixed_nesterov": (c_int32, [c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_double, c_int64,
                                     _P(SolveOpts), c_void_p, _P(Result)]),
    "apx_agraal": (c_int32, [c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_int64,
                             _P(SolveOpts), c_void_p, _P(Result)]),
    "apx_malitsky_pock": (c_int32, [c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_int64,
                                    _P(SolveOpts), c_void_p, c_void_p, _P(Result)]),
    "apx_solver_create": (c_int32, [c_void_p, _P(Rule), c_void_p, c_void_p, c_double, _P(SolveOpts), _P(c_void_p)]),
    "apx_solver_step": (c_int32, [c_void_p, c_int64, c_int32]),
    "apx_solver_result": (c_int32, [c_void_p, c_void_p, c_void_p, _P(Result)]),
    "apx_solver_destroy": (c_int32, [c_void_p]),
}


class ApxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"adaprox_hip error {code}: {msg}")
        self.code = code


class DomainError(ApxError, ValueError):
    """Julia would have thrown DomainError (sqrt of a negative number inside a rule)."""


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """Two HIP runtimes in one process do not share the GPU: whichever initialises second sees "no
    ROCm-capable device".  PyTorch-ROCm bundles its own libamdhip64; if it is already imported the loader
    resolves this library's DT_NEEDED to that copy by soname, but if THIS library were loaded first it would
    pull the system copy and a later `import torch` (the torch.distributed plumbing of the sharded path) would
    lose the device.  So when an installed torch ships a libamdhip64, map that copy first — without importing
    torch — and both libraries share one runtime whatever the import order."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    # (librccl is NOT mapped here: mapped before torch maps its own, it aborts the process at exit — `double free or
    # corruption` from the exit handlers; the library binds RCCL at its first apx_comm_* call, csrc/comm.hip, and
    # sharded.Comm imports torch first when it is installed so that the pair in the process is torch's own)
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> C.CDLL:
    """Loads libadaprox_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            f"(`make -C {os.path.join(_HERE, 'csrc')}` or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "This package has no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.apx_abi_version() != APX_ABI_VERSION:
        raise ImportError("libadaprox_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(code: int) -> None:
    if code == APX_OK:
        return
    msg = load().apx_last_error().decode("utf-8", "replace")
    if code == APX_ERR_DOMAIN:
        raise DomainError(code, msg)
    if code == APX_ERR_INVALID:
        raise ApxInvalid(code, msg)
    raise ApxError(code, msg)


class ApxInvalid(ApxError, ValueError):
    """Julia `error(...)` / `@assert` equivalents (bad arguments)."""

"""ctypes binding of libsolverforge_amd.so (the C ABI in include/solverforge_amd.h).

There is no CPU fallback: if the shared library or a HIP device is missing every call
raises.  The oracle under oracle/ is test infrastructure and is never imported here.
"""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
# SF_AMD_LIB: kernel-variant experiments only (scripts/); the product always loads the in-tree build
LIB_PATH = os.environ.get("SF_AMD_LIB") or os.path.join(_DIR, "libsolverforge_amd.so")

SF_OK = 0
ERRORS = {-1: "SF_ERR_INVALID", -2: "SF_ERR_NO_DEVICE", -3: "SF_ERR_HIP", -4: "SF_ERR_UNSUPPORTED", -5: "SF_ERR_CAPACITY"}

MOVE_DTYPE = np.dtype(
    [("kind", "<i4"), ("a", "<i4"), ("a_pos", "<i4"), ("b", "<i4"), ("b_pos", "<i4"), ("value", "<i4")]
)


class SolverConfigStruct(C.Structure):
    _fields_ = [
        ("acceptor", C.c_int32),
        ("late_acceptance_size", C.c_int32),
        ("forager", C.c_int32),
        ("accepted_count_limit", C.c_int32),
        ("random_ties", C.c_int32),
        ("selection_order", C.c_int32),
        ("random_seed", C.c_uint64),
    ]


class AnnealingConfigStruct(C.Structure):
    _fields_ = [
        ("mode", C.c_int32),
        ("never_accept_hard_regression", C.c_int32),
        ("calibration_sample_size", C.c_int32),
        ("reserved", C.c_int32),
        ("temperatures", C.c_double * 4),
        ("decay_rate", C.c_double),
        ("hill_climbing_temperature", C.c_double),
        ("target_acceptance_probability", C.c_double),
        ("fallback_temperature", C.c_double),
        ("seed", C.c_uint64),
    ]


class StatsStruct(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "step_count", "moves_generated", "moves_evaluated", "moves_accepted", "moves_applied",
        "score_calculations", "moves_not_doable", "candidates_scored", "sources_scanned", "reserved")]


class TraceDigestStruct(C.Structure):
    _fields_ = [("first", C.c_uint64), ("second", C.c_uint64)]


class TraceScopeStruct(C.Structure):
    _fields_ = [("phase_index", C.c_int32), ("phase_type", C.c_char_p), ("list_descriptor", C.c_int32),
                ("list_variable", C.c_char_p), ("scalar_descriptor", C.c_int32), ("scalar_variable", C.c_char_p)]


class SolverForgeError(RuntimeError):
    pass


# every symbol include/solverforge_amd.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "sf_ctx_create", "sf_ctx_destroy", "sf_last_error", "sf_device_count", "sf_sync",
    "sf_schema_add_entity_class", "sf_schema_add_scalar_variable", "sf_schema_add_list_variable",
    "sf_fact_matrix_i64", "sf_fact_column_i32", "sf_fact_column_u32", "sf_fact_csr_u32",
    "sf_constraint_add", "sf_constraint_add_list_precedence", "sf_selector_add", "sf_selector_add_sublist", "sf_selector_add_kopt", "sf_selector_add_permute", "sf_selector_add_precedence", "sf_list_set_precedence_policy", "sf_selector_add_ruin", "sf_selector_add_nearby_scalar", "sf_step_evaluate_compound", "sf_step_decide", "sf_step_decide_gated", "sf_step_decide_cursor", "sf_apply_compound", "sf_construct_list_cheapest", "sf_construct_list_regret", "sf_construct_list_clarke_wright", "sf_construct_list_round_robin", "sf_construct_list_k_opt", "sf_union_configure", "sf_schema_set_value_lists", "sf_initialize", "sf_evaluate_all", "sf_evaluate_each", "sf_get_scores",
    "sf_step_evaluate", "sf_apply", "sf_step_generate", "sf_solver_configure", "sf_default_local_search_components", "sf_solver_configure_default", "sf_solver_configure_annealing", "sf_solver_configure_diversified",
    "sf_get_annealing_state", "sf_solver_set_step_seeds",
    "sf_solver_set_engine", "sf_solver_get_engine", "sf_list_wave_layout", "sf_constraint_add_pair_join", "sf_constraint_add_uni_program", "sf_provider_declare", "sf_phase_start", "sf_solve_steps", "sf_solve_moves", "sf_solve_step_traced", "sf_get_stats", "sf_get_stats_sum", "sf_get_best_scores",
    "sf_profile_solve", "sf_download_scalar", "sf_download_list", "sf_portfolio_unique_id",
    "sf_portfolio_init", "sf_portfolio_allgather_best", "sf_portfolio_broadcast_best", "sf_portfolio_destroy", "sf_portfolio_migrate_local",
    "sf_trace_digest_init", "sf_trace_digest_update", "sf_trace_encode_step",
]

_lib = None


def load():
    """Load the HIP extension; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SolverForgeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The HIP path has no CPU fallback."
        )
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    i32, i64, u64, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_void_p
    L.sf_ctx_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    L.sf_ctx_destroy.argtypes = [vp]
    L.sf_ctx_destroy.restype = None
    L.sf_last_error.argtypes = [vp]
    L.sf_last_error.restype = C.c_char_p
    L.sf_device_count.argtypes = []
    L.sf_sync.argtypes = [vp]
    L.sf_schema_add_entity_class.argtypes = [vp, i32, i32]
    L.sf_schema_add_scalar_variable.argtypes = [vp, i32, i32, i32, i32, vp]
    L.sf_schema_add_list_variable.argtypes = [vp, i32, vp, vp, i32, i32]
    L.sf_fact_matrix_i64.argtypes = [vp, i32, i32, i32, vp]
    L.sf_fact_column_i32.argtypes = [vp, i32, i32, vp]
    L.sf_fact_column_u32.argtypes = [vp, i32, i32, vp]
    L.sf_fact_csr_u32.argtypes = [vp, i32, i32, vp, vp]
    L.sf_constraint_add.argtypes = [vp, i32, i32, i32, i32, i64, i32, i64]
    L.sf_constraint_add_list_precedence.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, i32]
    L.sf_selector_add.argtypes = [vp, i32, i32, i32, i32, i32]
    L.sf_initialize.argtypes = [vp, vp]
    L.sf_evaluate_all.argtypes = [vp, vp]
    L.sf_get_scores.argtypes = [vp, vp]
    L.sf_evaluate_each.argtypes = [vp, i32, vp, vp]
    L.sf_step_evaluate.argtypes = [vp, i32, vp, i64, vp, vp]
    L.sf_apply.argtypes = [vp, i32, vp]
    L.sf_step_generate.argtypes = [vp, i32, u64, u64, i32, vp, vp, vp, i64, vp]
    L.sf_selector_add_sublist.argtypes = [vp, i32, i32, i32, i32, i32]
    L.sf_selector_add_kopt.argtypes = [vp, i32, i32, i32, i32, i32]
    L.sf_selector_add_permute.argtypes = [vp, i32, i32, i32, i32]
    L.sf_selector_add_precedence.argtypes = [vp, i32, i32]
    L.sf_list_set_precedence_policy.argtypes = [vp, i32, i32, i32]
    L.sf_step_evaluate_compound.argtypes = [vp, i32, vp, vp, i64, vp, vp]
    L.sf_apply_compound.argtypes = [vp, i32, vp, i64]
    L.sf_step_decide.argtypes = [vp, i32, vp, vp, i64, i32, i64, vp, vp, vp, vp, vp, vp]
    L.sf_step_decide_gated.argtypes = [vp, i32, vp, vp, vp, i64, i32, i64, vp, vp, vp, vp, vp, vp]
    L.sf_step_decide_cursor.argtypes = [vp, i32, vp, vp, vp, i64, vp, vp, vp, vp]
    L.sf_schema_set_value_lists.argtypes = [vp, i32, i32, vp, vp]
    L.sf_union_configure.argtypes = [vp, i32, vp, i32]
    L.sf_construct_list_cheapest.argtypes = [vp, i32, vp, i32, vp]
    L.sf_construct_list_regret.argtypes = [vp, i32, vp, i32, vp, vp, vp]
    L.sf_construct_list_clarke_wright.argtypes = [vp, i32, vp, i32, i32, vp, vp]
    L.sf_construct_list_round_robin.argtypes = [vp, i32, vp, i32, vp, vp, vp]
    L.sf_construct_list_k_opt.argtypes = [vp, i32, i32, i32, i32, vp]
    L.sf_selector_add_nearby_scalar.argtypes = [vp, i32, i32, i32, i32, i64, vp, vp, vp, i32]
    L.sf_selector_add_ruin.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, C.c_char_p]
    L.sf_solver_configure.argtypes = [vp, C.POINTER(SolverConfigStruct)]
    L.sf_default_local_search_components.argtypes = [i32, i32, i32, i32, i32, u64, C.POINTER(SolverConfigStruct)]
    L.sf_solver_configure_default.argtypes = [vp, u64, i32, i32, C.POINTER(SolverConfigStruct)]
    L.sf_solver_configure_annealing.argtypes = [vp, C.POINTER(AnnealingConfigStruct)]
    L.sf_solver_configure_diversified.argtypes = [vp, C.c_double]
    L.sf_get_annealing_state.argtypes = [vp, i32, vp, C.POINTER(i32)]
    L.sf_solver_set_engine.argtypes = [vp, i32]
    L.sf_solver_get_engine.argtypes = [vp, C.POINTER(i32)]
    L.sf_list_wave_layout.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.sf_constraint_add_pair_join.argtypes = [vp, i32, i32, vp, i32, i32, C.c_int64]
    L.sf_constraint_add_uni_program.argtypes = [vp, i32, i32, vp, i32, vp, i32, C.c_int64]
    L.sf_provider_declare.argtypes = [vp, i32, C.c_char_p]
    L.sf_solver_set_step_seeds.argtypes = [vp, vp, i64]
    L.sf_phase_start.argtypes = [vp]
    L.sf_solve_steps.argtypes = [vp, i64]
    L.sf_solve_moves.argtypes = [vp, i64, i64]
    L.sf_solve_step_traced.argtypes = [vp, i32, vp, vp, vp, i64, vp, vp, vp]
    L.sf_get_stats.argtypes = [vp, i32, C.POINTER(StatsStruct)]
    L.sf_get_stats_sum.argtypes = [vp, C.POINTER(StatsStruct)]
    L.sf_get_best_scores.argtypes = [vp, vp]
    L.sf_profile_solve.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64)]
    L.sf_download_scalar.argtypes = [vp, i32, i32, i32, vp, i32]
    L.sf_download_list.argtypes = [vp, i32, i32, vp, vp, i32]
    L.sf_portfolio_unique_id.argtypes = [vp]
    L.sf_portfolio_init.argtypes = [vp, vp, i32, i32]
    L.sf_portfolio_allgather_best.argtypes = [vp, vp, vp, vp]
    L.sf_portfolio_destroy.argtypes = [vp]
    L.sf_portfolio_broadcast_best.argtypes = [vp, i32, i32, vp, vp]
    L.sf_portfolio_migrate_local.argtypes = [vp, i32, i32, C.POINTER(i32)]
    L.sf_trace_digest_init.argtypes = [C.POINTER(TraceDigestStruct)]
    L.sf_trace_digest_update.argtypes = [C.POINTER(TraceDigestStruct), vp, C.c_size_t]
    L.sf_trace_encode_step.argtypes = [C.POINTER(TraceScopeStruct), u64, u64, vp, vp, i64, vp, i64, C.POINTER(TraceDigestStruct)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name in ("sf_trace_digest_init", "sf_trace_digest_update"):
            fn.restype = None
        elif name == "sf_trace_encode_step":
            fn.restype = i64
        elif name not in ("sf_ctx_destroy", "sf_last_error"):
            fn.restype = i32
    _lib = L
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def check(rc, ctx=None):
    if rc == SF_OK:
        return
    msg = load().sf_last_error(ctx)
    raise SolverForgeError(f"{ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")

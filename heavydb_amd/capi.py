"""ctypes binding of include/mi355q.h (the C-ABI drop-in boundary).

Every structure here mirrors the header field-for-field; tests assert the sizes agree
with the library's own `mi355q_abi_sizeof_*` probes.  The library is built in-tree by
`heavydb_amd/_build.py` (hipcc, gfx950) and loading it FAILS LOUDLY if it is missing —
there is no CPU fallback for the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

MAX_COLS = 16
MAX_QUALS = 8
MAX_TARGETS = 8
MAX_SLOTS = 16
MAX_GROUP_COLS = 4
MAX_EXPRS = 8
MAX_EXPR_NODES = 24
MAX_EXPR_STACK = 8
ABI_VERSION = 7

# mi355q_type
INT8, INT16, INT32, INT64, DOUBLE, FLOAT = 1, 2, 3, 4, 5, 6
# mi355q_encoding
ENC_NONE, ENC_FIXED, ENC_DICT, ENC_DATE_IN_DAYS = 0, 1, 2, 3
# mi355q_op (SQLOps values)
EQ, NE, LT, GT, LE, GE = 0, 2, 3, 4, 5, 6
IS_NULL, IS_NOT_NULL = 16, 17  # unary quals (kISNULL, kISNOTNULL): no literal


class OrderEntry(C.Structure):
    """mi355q_order_entry (Analyzer::OrderEntry: tle_no - 1, is_desc, nulls_first)."""
    _fields_ = [("target_idx", C.c_int32), ("descending", C.c_int32), ("nulls_first", C.c_int32),
                ("reserved", C.c_int32)]
# mi355q_agg (SQLAgg values)
AVG, MIN, MAX, SUM, COUNT, PROJECT_KEY = 0, 1, 2, 3, 4, 100
PROJECT = 101  # a Projection step's non-aggregate target (the value of a column / expression)
COUNT_IF, SUM_IF = 10, 11
# mi355q_join_kind
JOIN_INNER, JOIN_LEFT = 0, 1
# mi355q_desc_type
GROUP_BY_PERFECT_HASH, GROUP_BY_BASELINE_HASH, NON_GROUPED_AGGREGATE = 0, 1, 4
PROJECTION = 2
OUTPUT_ROWWISE, OUTPUT_COLUMNAR, OUTPUT_ROWWISE_COLUMNAR_DECISIONS = 0, 1, 2  # mi355q_columnar_hint
# generator kinds
GEN_I32_UNIFORM31, GEN_I32_MOD, GEN_I64_MOD, GEN_I64_MOD_MUL, GEN_F64_UNIT = 1, 2, 3, 4, 5

# mi355q_expr_op (projected expressions)
EX_COL, EX_LIT, EX_CAST, EX_ADD, EX_SUB, EX_MUL, EX_DIV, EX_MOD = 1, 2, 3, 4, 5, 6, 7, 8
EX_EQ, EX_NE, EX_LT, EX_LE, EX_GT, EX_GE, EX_CASE = 9, 10, 11, 12, 13, 14, 15
EX_NOT, EX_AND, EX_OR, EX_IS_NULL, EX_UMINUS = 16, 17, 18, 19, 20

OK = 0
ERR_DIV_BY_ZERO = 1
ERR_OUT_OF_SLOTS = 3
ERR_OVERFLOW_OR_UNDERFLOW = 7
ERR_INVALID_PLAN = 100
ERR_UNSUPPORTED = 101
ERR_HIP = 102
ERR_JOIN_NOT_ONE_TO_ONE = 103
ERR_JOIN_TABLE_FULL = 104
STEP_RECOMPUTED = 110  # mi355q_wait: result complete, but re-run after the async call returned

TYPE_WIDTH = {INT8: 1, INT16: 2, INT32: 4, INT64: 8, DOUBLE: 8, FLOAT: 4}


class ColDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("nullable", C.c_int32), ("encoding", C.c_int32),
                ("logical_type", C.c_int32)]


class Qual(C.Structure):
    _fields_ = [("col", C.c_int32), ("op", C.c_int32), ("ival", C.c_int64), ("fval", C.c_double)]


class Target(C.Structure):
    _fields_ = [("agg", C.c_int32), ("col", C.c_int32), ("table", C.c_int32),
                ("reserved", C.c_int32), ("cond", Qual)]


class Range(C.Structure):
    _fields_ = [("valid", C.c_int32), ("has_nulls", C.c_int32), ("min", C.c_int64),
                ("max", C.c_int64), ("fp_min", C.c_double), ("fp_max", C.c_double),
                ("bucket", C.c_int64)]


class ExprNode(C.Structure):
    _fields_ = [("op", C.c_int32), ("type", C.c_int32), ("arg", C.c_int32), ("reserved", C.c_int32),
                ("ilit", C.c_int64), ("flit", C.c_double)]


class Expr(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("reserved", C.c_int32), ("nodes", ExprNode * MAX_EXPR_NODES),
                ("range", Range)]


class Plan(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_cols", C.c_int32),
        ("cols", ColDesc * MAX_COLS),
        ("col_ranges", Range * MAX_COLS),
        ("n_inner_cols", C.c_int32),
        ("inner_cols", ColDesc * MAX_COLS),
        ("inner_col_ranges", Range * MAX_COLS),
        ("n_quals", C.c_int32),
        ("quals", Qual * MAX_QUALS),
        ("n_group_cols", C.c_int32),
        ("group_cols", C.c_int32 * MAX_GROUP_COLS),
        ("n_targets", C.c_int32),
        ("targets", Target * MAX_TARGETS),
        ("join_outer_col", C.c_int32),
        ("join_table", C.c_void_p),
        ("n_join_cols", C.c_int32),
        ("join_outer_cols", C.c_int32 * MAX_GROUP_COLS),
        ("join_kind", C.c_int32),
        ("reserved2", C.c_int32),
        ("max_groups_buffer_entry_guess", C.c_int64),
        ("bigint_count", C.c_int32),
        ("output_columnar_hint", C.c_int32),
        ("num_tuples", C.c_int64),
        ("n_exprs", C.c_int32),
        ("reserved3", C.c_int32),
        ("exprs", Expr * MAX_EXPRS),
        ("scan_limit", C.c_int64),
    ]


class QMD(C.Structure):
    _fields_ = [
        ("desc_type", C.c_int32),
        ("keyless", C.c_int32),
        ("idx_target_as_key", C.c_int32),
        ("key_width", C.c_int32),
        ("group_col_count", C.c_int32),
        ("slot_count", C.c_int32),
        ("entry_count", C.c_int64),
        ("min_val", C.c_int64),
        ("max_val", C.c_int64),
        ("bucket", C.c_int64),
        ("group_min", C.c_int64 * MAX_GROUP_COLS),
        ("group_card", C.c_int64 * MAX_GROUP_COLS),
        ("group_null_key", C.c_int64 * MAX_GROUP_COLS),
        ("group_bucket", C.c_int64 * MAX_GROUP_COLS),
        ("group_has_nulls", C.c_int32 * MAX_GROUP_COLS),
        ("has_nulls", C.c_int32),
        ("row_size", C.c_int32),
        ("slot_width", C.c_int32),
        ("output_columnar", C.c_int32),
        ("key_bytes", C.c_int32),
        ("n_targets", C.c_int32),
        ("target_slot", C.c_int32 * MAX_TARGETS),
        ("target_key_idx", C.c_int32 * MAX_TARGETS),
        ("target_skip_null", C.c_int32 * MAX_TARGETS),
        ("target_is_fp", C.c_int32 * MAX_TARGETS),
        ("target_agg", C.c_int32 * MAX_TARGETS),
        ("target_arg_is_fp", C.c_int32 * MAX_TARGETS),
        ("target_arg_is_f32", C.c_int32 * MAX_TARGETS),
        ("target_null", C.c_int64 * MAX_TARGETS),
        ("init_vals", C.c_int64 * MAX_SLOTS),
        ("slot_bytes", C.c_int32 * MAX_SLOTS),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


class Inputs(C.Structure):
    _fields_ = [
        ("device_id", C.c_int32),
        ("n_frags", C.c_int32),
        ("col_buffers", C.POINTER(C.c_void_p)),
        ("num_rows", C.POINTER(C.c_int64)),
        ("inner_col_buffers", C.POINTER(C.c_void_p)),
        ("inner_num_rows", C.c_int64),
        ("inner_version", C.c_int64),
    ]


class ExecOptions(C.Structure):
    _fields_ = [
        ("stream", C.c_void_p),
        ("out_buffer", C.c_void_p),
        ("force_generic", C.c_int32),
        ("kernel_variant", C.c_int32),
        ("scratch_bytes", C.c_int64),
        ("tune_blocks_per_cu", C.c_int32),
        ("probe_keyed_passes", C.c_int32),
        ("pass_rows", C.c_int64),
        ("flags", C.c_uint32),
        ("tune_cus", C.c_int32),
        ("tune_overlap_cus", C.c_int32),
        ("reserved0", C.c_int32),
    ]


OPT_TRACE, OPT_NO_PAIR_RENDEZVOUS, OPT_PROBE_NO_PACING, OPT_NO_LDS_BASELINE, OPT_LDS_BASELINE_LARGE = 1, 2, 4, 8, 16
OPT_LDS_BASELINE_WINDOWS = 32
OPT_LDS_GENERIC_MEMBER = 64
OPT_NO_IDX_PART = 128
OPT_NO_COMPILED_FILTER = 256
OPT_FILTER_PREPASS = 512
OPT_NO_IDX_PACK = 1024


class ExecReport(C.Structure):
    _fields_ = [
        ("kernel_name", C.c_char * 64),
        ("kernel_ms", C.c_float),
        ("total_ms", C.c_float),
        ("n_launches", C.c_int32),
        ("variant", C.c_int32),
        ("rows_scanned", C.c_int64),
        ("algorithmic_bytes", C.c_int64),
        ("spilled_rows", C.c_int64),
    ]


class JoinSpec(C.Structure):
    _fields_ = [
        ("device_id", C.c_int32),
        ("key_type", C.c_int32),
        ("key_nullable", C.c_int32),
        ("prefer_baseline", C.c_int32),
        ("key_buffer", C.c_void_p),
        ("num_rows", C.c_int64),
        ("key_range", Range),
        ("max_perfect_entries", C.c_int64),
        ("n_keys", C.c_int32),
        ("one_to_many", C.c_int32),
        ("more_key_types", C.c_int32 * (MAX_GROUP_COLS - 1)),
        ("more_key_nullables", C.c_int32 * (MAX_GROUP_COLS - 1)),
        ("more_key_buffers", C.c_void_p * (MAX_GROUP_COLS - 1)),
        ("keyed_entry_count", C.c_int64),
    ]


# every symbol include/mi355q.h declares: (name, restype, argtypes)
_P = C.POINTER
SYMBOLS = [
    ("mi355q_abi_version", C.c_int32, []),
    ("mi355q_abi_sizeof", C.c_int64, [C.c_int32]),
    ("mi355q_error_string", C.c_char_p, [C.c_int32]),
    ("mi355q_device_count", C.c_int32, []),
    ("mi355q_release_workspace", C.c_int32, [C.c_int32]),
    ("mi355q_device_info", C.c_int32,
     [C.c_int32, C.c_char_p, _P(C.c_int32), _P(C.c_int64), _P(C.c_int64), _P(C.c_int32),
      _P(C.c_int32)]),
    ("mi355q_qmd_init", C.c_int32, [_P(Plan), _P(QMD)]),
    ("mi355q_qmd_buffer_bytes", C.c_int64, [_P(QMD)]),
    ("mi355q_qmd_group_col_offset", C.c_int64, [_P(QMD), C.c_int32]),
    ("mi355q_qmd_slot_col_offset", C.c_int64, [_P(QMD), C.c_int32]),
    ("mi355q_execute", C.c_int32,
     [_P(Plan), _P(Inputs), _P(ExecOptions), _P(C.c_void_p), _P(ExecReport)]),
    ("mi355q_execute_async", C.c_int32,
     [_P(Plan), _P(Inputs), _P(ExecOptions), _P(C.c_void_p), _P(C.c_void_p)]),
    ("mi355q_wait", C.c_int32, [C.c_void_p, _P(ExecReport)]),
    ("mi355q_reserve_workspace", C.c_int32, [_P(Plan), _P(Inputs), _P(ExecOptions), _P(C.c_int64)]),
    ("mi355q_result_create", C.c_int32, [_P(QMD), C.c_int32, C.c_void_p, _P(C.c_void_p)]),
    ("mi355q_result_wrap", C.c_int32, [_P(QMD), C.c_int32, C.c_void_p, _P(C.c_void_p)]),
    ("mi355q_result_free", None, [C.c_void_p]),
    ("mi355q_result_topk", C.c_int32,
     [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, _P(C.c_int64), C.c_void_p]),
    ("mi355q_shard_pads", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355q_shard_merge_range", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    ("mi355q_shard_merge_slices", C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32,
                                              C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    ("mi355q_explain", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p]),
    ("mi355q_result_export_arrow", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355q_result_sort", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, _P(C.c_int64), C.c_void_p]),
    ("mi355q_result_qmd", C.c_int32, [C.c_void_p, _P(QMD)]),
    ("mi355q_result_device_ptr", C.c_void_p, [C.c_void_p]),
    ("mi355q_result_bytes", C.c_int64, [C.c_void_p]),
    ("mi355q_result_copy_to_host", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("mi355q_result_reduce", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355q_result_row_count", C.c_int64, [C.c_void_p]),
    ("mi355q_result_total_matched", C.c_int64, [C.c_void_p]),
    ("mi355q_result_append", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355q_result_fetch_rows", C.c_int32,
     [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_int64)]),
    ("mi355q_result_to_columns", C.c_int32, [C.c_void_p, _P(C.c_void_p), C.c_int32, _P(C.c_int64), C.c_void_p]),
    ("mi355q_join_build", C.c_int32, [_P(JoinSpec), C.c_void_p, _P(C.c_void_p)]),
    ("mi355q_join_free", None, [C.c_void_p]),
    ("mi355q_join_key_shape", C.c_int32, [C.c_void_p, _P(C.c_int32), _P(C.c_int32)]),
    ("mi355q_join_invalidate_payload", C.c_int32, [C.c_void_p]),
    ("mi355q_join_payload_info", C.c_int32, [C.c_void_p, _P(C.c_int64), _P(C.c_float), _P(C.c_int64)]),
    ("mi355q_join_info", C.c_int32,
     [C.c_void_p, _P(C.c_int32), _P(C.c_int64), _P(C.c_int64), _P(C.c_int64), _P(C.c_void_p),
      _P(C.c_int64), _P(C.c_float)]),
    ("mi355q_shard_partition", C.c_int32,
     [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355q_shard_merge_rows", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("mi355q_generate_column", C.c_int32,
     [C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_uint64, C.c_int64, C.c_int64,
      C.c_int64, C.c_double, C.c_int32, C.c_void_p]),
]

LIB_NAME = "libmi355q.so"


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", LIB_NAME)


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen libmi355q.so and bind every declared symbol.  Raises if absent: the product
    path has no fallback (the CPU oracle is test infrastructure only)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or lib_path()
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). "
            "heavydb_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64, and device pointers
    # / streams only interoperate when both sides resolve to the same copy.  If torch is going
    # to be used in this process it must be the first to load the runtime, so import it (when
    # installed) before dlopen-ing the library; loading the library first and torch afterwards
    # makes the first launch fail with a HIP error.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional for the C-ABI itself
            pass
    lib = C.CDLL(p)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.mi355q_abi_version() != ABI_VERSION:
        raise RuntimeError("mi355q ABI version mismatch")
    for which, st in ((1, Plan), (2, QMD), (3, Inputs), (4, ExecOptions), (5, ExecReport),
                      (6, JoinSpec)):
        if lib.mi355q_abi_sizeof(which) != C.sizeof(st):
            raise RuntimeError(f"ctypes mirror of struct #{which} disagrees with the library")
    if path is None:
        _lib = lib
    return lib


class Mi355qError(RuntimeError):
    def __init__(self, code: int, where: str = ""):
        self.code = code
        msg = None
        try:
            msg = load_library().mi355q_error_string(code).decode()
        except Exception:  # pragma: no cover
            pass
        super().__init__(f"mi355q error {code} ({msg}) {where}")


def check(code: int, where: str = "") -> None:
    if code != 0:
        raise Mi355qError(code, where)

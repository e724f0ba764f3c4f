"""Projection (row-emitting filter / project) cases shared by the CPU tests (host simulation of the library) and the
`-m gpu` tests: `SELECT c0, c1, ... FROM t WHERE quals [LIMIT n]` as a RelAlgExecutionUnit whose targets are all
capi.PROJECT.  Shapes follow the reference's own projection tests (Tests/ExecuteTest.cpp `SELECT x, y FROM test WHERE
...`, Select.FilterAndSimpleAggregation's filters without the aggregate; Tests/ResultSetTest.cpp Iterate.* over
Projection storage) and its runtime (get_scan_output_slot / get_columnar_scan_output_offset, GroupByRuntime.cpp:242-269)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from heavydb_amd import capi
from heavydb_amd.executor import Expr, ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr

NP = {capi.INT8: np.int8, capi.INT16: np.int16, capi.INT32: np.int32, capi.INT64: np.int64, capi.DOUBLE: np.float64,
      capi.FLOAT: np.float32}
INT_NULL = {capi.INT8: -2**7, capi.INT16: -2**15, capi.INT32: -2**31, capi.INT64: -2**63}


@dataclass
class ProjCase:
    name: str
    ra: RelAlgExecutionUnit
    frags: List[List[np.ndarray]]
    expect_error: Optional[int] = None   # > 0: that code; < 0: any negative code (buffer full)
    min_rows: int = 0                    # the case is meant to produce at least this many rows
    # a join (the fields of tests/cases.py Case, read by test_hostsim_flow._build_join / _oracle_join)
    inner: List[np.ndarray] = field(default_factory=list)
    join_keys: object = None
    join_key_type: object = capi.INT64
    join_range: Optional[ExpressionRange] = None
    join_prefer_baseline: bool = False
    join_one_to_many: int = 0
    join_key_nullable: object = False


def _col(rng, t, n, lo=-1000, hi=1000, null_every=0):
    if t == capi.DOUBLE:
        a = rng.uniform(lo, hi, n).astype(np.float64)
        if null_every:
            a[::null_every] = np.finfo(np.float64).tiny  # NULL_DOUBLE = DBL_MIN
        return a
    if t == capi.FLOAT:
        a = rng.uniform(lo, hi, n).astype(np.float32)
        if null_every:
            a[::null_every] = np.finfo(np.float32).tiny  # NULL_FLOAT = FLT_MIN
        return a
    info = np.iinfo(NP[t])
    a = rng.integers(max(lo, info.min + 1), min(hi, info.max) + 1, n).astype(NP[t])
    if null_every:
        a[::null_every] = INT_NULL[t]
    return a


def _split(cols, sizes):
    out, o = [], 0
    for s in sizes:
        out.append([c[o:o + s] for c in cols])
        o += s
    return out


def build_cases(scale: int = 1) -> List[ProjCase]:
    """scale multiplies the row counts (1: a few 10 K rows — several 16 K-row tiles per fragment, ragged ends)."""
    rng = np.random.default_rng(20250923)
    cases: List[ProjCase] = []
    D = InputColDescriptor
    R = ExpressionRange

    def add(name, descs, cols, targets, quals, sizes, **kw):
        err = kw.pop("expect_error", None)
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT, c) for c in targets], quals, **kw)
        cases.append(ProjCase(name, ra, _split(cols, sizes), err))

    n = 40_000 * scale
    # 1. one int32 filter, three int64 / double columns out; two fragments with ragged sizes
    f = _col(rng, capi.INT32, n, 0, 999)
    a = _col(rng, capi.INT64, n, -10**12, 10**12)
    b = _col(rng, capi.DOUBLE, n)
    c = _col(rng, capi.INT64, n, 0, 10**6)
    descs = [D(capi.INT32), D(capi.INT64), D(capi.DOUBLE), D(capi.INT64)]
    for sel, k in (("1pct", 10), ("50pct", 500), ("99pct", 990)):
        add(f"i32_filter_{sel}_3cols", descs, [f, a, b, c], [1, 2, 3], [Qual(0, capi.LT, k)], [n // 2 + 3, n - n // 2 - 3],
            max_groups_buffer_entry_guess=n)
    add("i32_filter_50pct_1col", descs, [f, a, b, c], [1], [Qual(0, capi.LT, 500)], [n], max_groups_buffer_entry_guess=n)
    add("i32_filter_columnar_3cols", descs, [f, a, b, c], [1, 2, 3], [Qual(0, capi.LT, 500)], [17_001, n - 17_001],
        max_groups_buffer_entry_guess=n, output_columnar_hint=capi.OUTPUT_COLUMNAR)
    # 2. no filter at all: every row, in order; the filter column itself projected
    add("no_filter_2cols", descs, [f, a, b, c], [0, 3], [], [n], max_groups_buffer_entry_guess=n)
    # 3. nothing matches / empty input / empty fragment in the middle
    add("nothing_matches", descs, [f, a, b, c], [1, 2], [Qual(0, capi.LT, -5)], [n], max_groups_buffer_entry_guess=64)
    add("empty_input", descs, [f[:0], a[:0], b[:0], c[:0]], [1], [Qual(0, capi.LT, 500)], [0], max_groups_buffer_entry_guess=64)
    add("empty_middle_fragment", descs, [f, a, b, c], [1, 3], [Qual(0, capi.GE, 900)], [1000, 0, n - 1000],
        max_groups_buffer_entry_guess=n)
    # 4. every type, nullable, as filter and as output; row-wise and columnar (logical-sized slot columns)
    m = 21_000 * scale
    tcols = [_col(rng, capi.INT8, m, -100, 100, 7), _col(rng, capi.INT16, m, -3000, 3000, 11), _col(rng, capi.INT32, m, -10**6, 10**6, 13),
             _col(rng, capi.INT64, m, -10**15, 10**15, 17), _col(rng, capi.DOUBLE, m, -1e6, 1e6, 19), _col(rng, capi.FLOAT, m, -1e3, 1e3, 23)]
    tdescs = [D(capi.INT8, True), D(capi.INT16, True), D(capi.INT32, True), D(capi.INT64, True), D(capi.DOUBLE, True), D(capi.FLOAT, True)]
    for hint, tag in ((0, "rowwise"), (capi.OUTPUT_COLUMNAR, "columnar")):
        add(f"all_types_nullable_{tag}", tdescs, tcols, [0, 1, 2, 3, 4, 5], [Qual(1, capi.GT, -1500), Qual(4, capi.LT, 5e5)],
            [m // 3, m - m // 3], max_groups_buffer_entry_guess=m, output_columnar_hint=hint)
        add(f"is_null_filter_{tag}", tdescs, tcols, [2, 5, 0], [Qual(2, capi.IS_NULL)], [m], max_groups_buffer_entry_guess=m,
            output_columnar_hint=hint)
        add(f"float_filter_{tag}", tdescs, tcols, [5, 3], [Qual(5, capi.GE, 0.0), Qual(3, capi.IS_NOT_NULL)], [m - 5, 5],
            max_groups_buffer_entry_guess=m, output_columnar_hint=hint)
    # NOT NULL columns that hold the sentinel pattern: never NULL (ResultSet::isNull tests the type first)
    s32 = _col(rng, capi.INT32, 4096, -5, 5)
    s32[::5] = INT_NULL[capi.INT32]
    add("not_null_holding_sentinel", [D(capi.INT32)], [s32], [0], [], [4096], max_groups_buffer_entry_guess=4096)
    # 5. encoded inputs: kENCODING_FIXED (int64 stored as int16), dictionary ids (uint8), DATE in days (int32 -> seconds)
    e16 = _col(rng, capi.INT16, m, -30000, 30000, 9)
    d8 = rng.integers(0, 255, m).astype(np.uint8).view(np.int8)
    d8.view(np.uint8)[::13] = 255
    days = _col(rng, capi.INT32, m, -20000, 20000, 29)
    add("encoded_columns", [D(capi.INT16, True, R(), capi.ENC_FIXED, capi.INT64), D(capi.INT8, True, R(), capi.ENC_DICT),
                            D(capi.INT32, True, R(), capi.ENC_DATE_IN_DAYS)],
        [e16, d8, days], [0, 1, 2], [Qual(0, capi.GT, -10000)], [m], max_groups_buffer_entry_guess=m)
    add("encoded_columns_columnar", [D(capi.INT16, True, R(), capi.ENC_FIXED, capi.INT64), D(capi.INT8, True, R(), capi.ENC_DICT),
                                     D(capi.INT32, True, R(), capi.ENC_DATE_IN_DAYS)],
        [e16, d8, days], [2, 1, 0], [Qual(1, capi.LT, 100)], [m // 2, m - m // 2], max_groups_buffer_entry_guess=m,
        output_columnar_hint=capi.OUTPUT_COLUMNAR)
    # 6. LIMIT: the first scan_limit matches in (fragment, row) order, the step ends normally; the limit is never reached
    add("scan_limit_cuts", descs, [f, a, b, c], [1, 3], [Qual(0, capi.LT, 500)], [n // 2, n - n // 2], scan_limit=1000)
    add("scan_limit_cuts_columnar", descs, [f, a, b, c], [1, 3], [Qual(0, capi.LT, 500)], [n // 2, n - n // 2], scan_limit=777,
        output_columnar_hint=capi.OUTPUT_COLUMNAR)
    add("scan_limit_not_reached", descs, [f, a, b, c], [2], [Qual(0, capi.LT, 3)], [n], scan_limit=n)
    add("scan_limit_inside_first_tile", descs, [f, a, b, c], [0], [Qual(0, capi.GE, 0)], [n], scan_limit=5)
    # 7. the buffer is too small and there is no limit: a negative code (the caller resizes and retries)
    add("buffer_full_no_limit", descs, [f, a, b, c], [1], [Qual(0, capi.LT, 500)], [n], max_groups_buffer_entry_guess=100, expect_error=-1)
    # 8. a disjunction among the quals (OR group)
    add("or_group", descs, [f, a, b, c], [0, 3], [Qual(0, capi.LT, 100, 1), Qual(0, capi.GT, 900, 1), Qual(3, capi.GE, 1000)], [n],
        max_groups_buffer_entry_guess=n)
    # 8b. literals outside the column's type: `i32 = 5000000000` matches nothing, `<>` every row, `<=` every row —
    # the members that compare an INT32 column in 32 bits must not wrap the bound
    fw = _col(rng, capi.INT32, n, 700_000_000, 2_000_000_000)
    for tag, op, lit in (("eq", capi.EQ, 5_000_000_000), ("ne", capi.NE, 5_000_000_000), ("le", capi.LE, 5_000_000_000),
                         ("ge_neg", capi.GE, -5_000_000_000), ("lt_neg", capi.LT, -5_000_000_000)):
        add(f"literal_beyond_int32_{tag}", descs, [fw, a, b, c], [0, 1], [Qual(0, op, lit)], [n // 2, n - n // 2],
            max_groups_buffer_entry_guess=n)
    # 9. many fragments, some smaller than a tile, unaligned chunk starts (the scalar path of the loads)
    sizes = [1, 3, 16384, 16385, 5, 70, 1000] + [n - 33848]
    add("many_small_fragments", descs, [f, a, b, c], [1, 2, 3], [Qual(0, capi.LT, 700)], sizes, max_groups_buffer_entry_guess=n)
    # 10. eight output columns (the widest row: sub-tiles of 1 024 rows)
    w8 = [_col(rng, capi.INT64, m, -10**9, 10**9) for _ in range(8)] + [_col(rng, capi.INT32, m, 0, 99)]
    add("eight_columns", [D(capi.INT64)] * 8 + [D(capi.INT32)], w8, list(range(8)), [Qual(8, capi.LT, 60)], [m],
        max_groups_buffer_entry_guess=m)

    # 11. projected expressions and expressions in quals, evaluated in the compaction kernel's registers
    def add_x(name, descs_, cols_, exprs, targets, quals, sizes_, **kw):
        err = kw.pop("expect_error", None)
        ra = RelAlgExecutionUnit(descs_, [TargetExpr(capi.PROJECT, t) for t in targets], quals, exprs=exprs, **kw)
        cases.append(ProjCase(name, ra, _split(cols_, sizes_), err))
    x = _col(rng, capi.INT32, m, -1000, 1000, 31)
    y = _col(rng, capi.INT32, m, -50, 50)
    z = _col(rng, capi.DOUBLE, m, -10, 10, 37)
    xd = [D(capi.INT32, True), D(capi.INT32), D(capi.DOUBLE, True)]
    I32, F64 = capi.INT32, capi.DOUBLE
    C_, L = Expr.col, Expr.lit
    add_x("expr_targets", xd, [x, y, z], [C_(0).add(C_(1), I32), C_(2).mul(L(F64, 2.5), F64), C_(0).cast(F64)], [3, 4, 5, 1],
          [Qual(1, capi.GT, 0)], [m // 2, m - m // 2], max_groups_buffer_entry_guess=m)
    add_x("expr_targets_columnar", xd, [x, y, z], [C_(0).add(C_(1), I32), C_(0).cast(capi.INT64).mul(L(capi.INT64, 1000), capi.INT64)], [3, 4, 0],
          [Qual(1, capi.GT, 0)], [m], max_groups_buffer_entry_guess=m, output_columnar_hint=capi.OUTPUT_COLUMNAR)
    # round 6: targets that are FORMS — [CAST](column) <op> literal — evaluated by the fast member on the quad it loads
    I64 = capi.INT64
    add_x("expr_form_targets", xd, [x, y, z], [C_(0).add(L(I32, 5), I32), C_(2).mul(L(F64, 2.5), F64), C_(0).cast(I64).mul(L(I64, 1000), I64),
                                             ], [3, 4, 5, 1],
          [Qual(1, capi.GT, -10)], [m // 2 + 1, m - m // 2 - 1], max_groups_buffer_entry_guess=m)
    add_x("expr_form_targets_literal_first_and_cast", xd, [x, y, z], [L(I32, 10).sub(C_(1), I32), C_(1).cast(F64)], [3, 4, 0],
          [Qual(1, capi.GT, -10)], [m], max_groups_buffer_entry_guess=m)
    # (forms are instantiated for <= 4 targets: a wider row keeps the general member's interpreter)
    add_x("expr_targets_six_wide", xd, [x, y, z], [C_(0).add(L(I32, 5), I32), C_(2).mul(L(F64, 2.5), F64), C_(0).cast(I64).mul(L(I64, 1000), I64),
                                                 L(I32, 10).sub(C_(1), I32), C_(1).cast(F64)], [3, 4, 5, 6, 7, 1],
          [Qual(1, capi.GT, -10)], [m // 2 + 1, m - m // 2 - 1], max_groups_buffer_entry_guess=m)
    add_x("expr_form_targets_columnar", xd, [x, y, z], [C_(0).add(L(I32, 5), I32), C_(2).sub(L(F64, 0.5), F64), C_(0).cast(I64).mul(L(I64, 1000), I64)],
          [3, 4, 5, 2], [Qual(1, capi.GT, 0)], [m], max_groups_buffer_entry_guess=m, output_columnar_hint=capi.OUTPUT_COLUMNAR)
    # ... an overflow in an emitted row raises error 7; in a row the filter drops, or one past the LIMIT, none
    big = x.astype(np.int32).copy()
    big[m // 2:] = 2**31 - 3
    add_x("expr_form_overflow_counts", xd, [big, y, z], [C_(0).add(L(I32, 5), I32)], [3], [Qual(1, capi.GE, -50)], [m], max_groups_buffer_entry_guess=m,
          expect_error=capi.ERR_OVERFLOW_OR_UNDERFLOW)
    add_x("expr_form_overflow_past_the_limit", xd, [big, y, z], [C_(0).add(L(I32, 5), I32)], [3, 1], [Qual(1, capi.GE, -50)], [m], scan_limit=50)
    add_x("expr_form_overflow_filtered_out", xd, [big, np.where(np.arange(m) >= m // 2, -60, y).astype(np.int32), z], [C_(0).add(L(I32, 5), I32)], [3, 1],
          [Qual(1, capi.GE, -50)], [m], max_groups_buffer_entry_guess=m)
    # round 6: expressions that all belong to the FILTER: compiled, evaluated by the row-mask pre-pass, the Projection runs on `mask = 1`
    add_x("expr_filter_sum", xd, [x, y, z], [C_(0).add(C_(1), I32).cmp(capi.EX_GT, L(I32, 100))], [0, 1, 2], [Qual(3, capi.EQ, 1), Qual(1, capi.LT, 40)],
          [m // 2 + 3, m - m // 2 - 3], max_groups_buffer_entry_guess=m)
    add_x("expr_filter_guarded_div_columnar", xd, [x, y, z],
          [C_(1).cmp(capi.EX_NE, L(I32, 0)).logical(capi.EX_AND, C_(0).div(C_(1), I32).cmp(capi.EX_GT, L(I32, 3)), True)], [0, 2], [Qual(3, capi.EQ, 1)],
          [m], max_groups_buffer_entry_guess=m, output_columnar_hint=capi.OUTPUT_COLUMNAR)
    add_x("expr_filter_double_and_limit", xd, [x, y, z], [C_(2).cmp(capi.EX_LT, C_(1).cast(F64))], [0, 2], [Qual(3, capi.EQ, 1)], [m], scan_limit=500)
    add_x("expr_filter_unguarded_div_raises", xd, [x, y, z], [C_(0).div(C_(1), I32).cmp(capi.EX_GT, L(I32, 3))], [0], [Qual(3, capi.EQ, 1)], [m],
          max_groups_buffer_entry_guess=m, expect_error=capi.ERR_DIV_BY_ZERO)
    # (a filter that can raise, under a LIMIT, keeps the general member; its pass A evaluates a whole tile's quals before the
    # limit is known: an error in a row of the tile that reaches the limit counts even behind the last row kept — DESIGN §3.2)
    # WHERE x + y > 100 (a BOOLEAN expression = 1) AND y < 40; the CASE of a guarded division as output
    cond = C_(0).add(C_(1), I32).cmp(capi.EX_GT, L(I32, 100))
    guarded = Expr.case(C_(1).cmp(capi.EX_NE, L(I32, 0)), C_(0).div(C_(1), I32), L(I32, 0), I32)
    add_x("expr_in_qual_and_case", xd, [x, y, z], [cond, guarded], [0, 1, 4], [Qual(3, capi.EQ, 1), Qual(1, capi.LT, 40)], [m],
          max_groups_buffer_entry_guess=m)
    # an expression that reads an earlier one; the root is filtered on
    band = C_(0).cmp(capi.EX_GT, L(I32, -200)).logical(capi.EX_AND, C_(0).cmp(capi.EX_LT, L(I32, 200)))
    add_x("expr_reads_expr", xd, [x, y, z], [band, C_(3).logical(capi.EX_OR, C_(1).cmp(capi.EX_EQ, L(I32, 7)))], [0, 1], [Qual(4, capi.EQ, 1)],
          [m], max_groups_buffer_entry_guess=m)
    # a division by zero in a row that passes: error 1; in a row the filter drops: none
    add_x("div_by_zero_counts", xd, [x, y, z], [C_(0).div(C_(1), I32)], [3], [Qual(1, capi.GE, 0)], [m], max_groups_buffer_entry_guess=m,
          expect_error=capi.ERR_DIV_BY_ZERO)
    # ... in a row that passes but lies past the LIMIT: never written, never raised (the reference's loop stops at max_matched)
    y_late = y.copy()
    y_late[:m // 2] = np.where(y_late[:m // 2] == 0, 1, y_late[:m // 2])
    assert (y_late[m // 2:] == 0).any()
    add_x("div_by_zero_past_the_limit", xd, [x, y_late, z], [C_(0).div(C_(1), I32)], [3], [Qual(1, capi.GE, 0)], [m], scan_limit=100)
    add_x("div_by_zero_filtered_out", xd, [x, y, z], [C_(0).div(C_(1), I32)], [3], [Qual(1, capi.GT, 0)], [m], max_groups_buffer_entry_guess=m)
    return cases


def build_join_cases() -> List[ProjCase]:
    """SELECT t.a, d.w, d.f, ... FROM t JOIN d ON t.k = d.k WHERE ...: every joined row is one output entry (Joins_* of
    Tests/ExecuteTest.cpp without the aggregate; the join loops enclose the row function's body, IRCodegen.cpp buildJoinLoops;
    an unmatched row of a LEFT join shows the inner columns' NULLs, ColumnIR.cpp codegenOuterJoinNullPlaceholder)."""
    rng = np.random.default_rng(77)
    D, R = InputColDescriptor, ExpressionRange
    cases: List[ProjCase] = []
    m, n = 700, 30_000
    dk = rng.permutation(m).astype(np.int64)                      # dense unique keys 0 .. m-1
    dw = rng.integers(-1000, 1000, m).astype(np.int64)
    df = rng.random(m)
    dg = rng.random(m).astype(np.float32)
    ds = rng.integers(-100, 100, m).astype(np.int16)
    ds[::9] = INT_NULL[capi.INT16]
    inner_descs = [D(capi.INT64, False, R(True, 0, m - 1)), D(capi.INT64, False, R(True, -1000, 999)), D(capi.DOUBLE), D(capi.FLOAT),
                   D(capi.INT16, True, R(True, -100, 99, True))]
    inner = [dk, dw, df, dg, ds]
    k = rng.integers(-60, m + 60, n).astype(np.int64)              # some keys miss
    a = rng.integers(-10**9, 10**9, n).astype(np.int64)
    f = rng.integers(0, 1000, n).astype(np.int32)
    k32 = rng.integers(0, m + 40, n).astype(np.int32)
    k32[::11] = INT_NULL[capi.INT32]                               # a NULL key matches nothing
    descs = [D(capi.INT64, False, R(True, -60, m + 59)), D(capi.INT64), D(capi.INT32, False, R(True, 0, 999)),
             D(capi.INT32, True, R(True, 0, m + 39, True))]
    frags = _split([k, a, f, k32], [n // 2 + 5, n - n // 2 - 5])
    dense = R(True, 0, m - 1)

    def add(name, targets, quals=(), outer_col=0, kind=capi.JOIN_INNER, keyed=False, **kw):
        err = kw.pop("expect_error", None)
        otm = kw.pop("one_to_many", 0)
        jk = kw.pop("join_keys", dk)
        inn = kw.pop("inner", inner)
        ra = RelAlgExecutionUnit(list(descs), [TargetExpr(capi.PROJECT, c, t) for c, t in targets], list(quals),
                                 inner_col_descs=list(inner_descs), join_outer_col=outer_col, join_kind=kind,
                                 max_groups_buffer_entry_guess=kw.pop("guess", n), **kw)
        cases.append(ProjCase(name, ra, frags, err, 0, list(inn), jk, capi.INT64, dense if not otm else R(True, 0, m - 1), keyed, otm,
                              False))

    both = [(1, 0), (1, 1), (2, 1), (3, 1), (4, 1)]                # t.a, d.w, d.f, d.g (FLOAT), d.s (nullable INT16)
    for keyed in (False, True):
        tag = "keyed" if keyed else "perfect"
        add(f"join_inner_{tag}", both, keyed=keyed)
        add(f"join_inner_{tag}_filtered", both, [Qual(2, capi.LT, 300)], keyed=keyed)
        add(f"join_left_{tag}", both, kind=capi.JOIN_LEFT, keyed=keyed)
        add(f"join_left_{tag}_columnar", both, [Qual(2, capi.GE, 100)], kind=capi.JOIN_LEFT, keyed=keyed,
            output_columnar_hint=capi.OUTPUT_COLUMNAR)
        add(f"join_inner_{tag}_nullable_int32_key", [(0, 0), (1, 1), (4, 1)], outer_col=3, keyed=keyed)
    add("join_outer_columns_only", [(1, 0), (2, 0)], [Qual(2, capi.LT, 500)])            # a semi-join: the match filters
    add("join_inner_columnar", both, output_columnar_hint=capi.OUTPUT_COLUMNAR)
    add("join_scan_limit", both, scan_limit=1234)
    add("join_left_nothing_matches", [(1, 0), (1, 1), (2, 1)], [Qual(0, capi.LT, 0)], kind=capi.JOIN_LEFT)
    add("join_inner_nothing_matches", [(1, 0), (1, 1)], [Qual(0, capi.LT, 0)])
    # duplicates on the inner side: a one-to-many table — every match is an entry, a row's matches in payload order (round 6:
    # k_proj_join_1n; HashJoin::codegenMatchingSet + the join loop around the body)
    dup = np.concatenate([dk[:m - 50], dk[:50], dk[:20], dk[5:8]])        # keys with 1, 2, 3 and 4 matches; 50 keys with none
    m2 = len(dup)
    inner2 = [dup, rng.integers(-1000, 1000, m2).astype(np.int64), rng.random(m2), rng.random(m2).astype(np.float32),
              np.where(np.arange(m2) % 9 == 0, INT_NULL[capi.INT16], rng.integers(-100, 100, m2)).astype(np.int16)]
    for keyed in (False, True):
        tag = "keyed" if keyed else "perfect"
        add(f"join_1n_inner_{tag}", both, one_to_many=1, join_keys=dup, inner=inner2, keyed=keyed, guess=3 * n)
        add(f"join_1n_left_{tag}_filtered", both, [Qual(2, capi.LT, 700)], kind=capi.JOIN_LEFT, one_to_many=1, join_keys=dup, inner=inner2,
            keyed=keyed, guess=3 * n)
    add("join_1n_inner_columnar", both, one_to_many=1, join_keys=dup, inner=inner2, guess=3 * n, output_columnar_hint=capi.OUTPUT_COLUMNAR)
    add("join_1n_left_nullable_int32_key", [(0, 0), (1, 1), (4, 1)], outer_col=3, kind=capi.JOIN_LEFT, one_to_many=1, join_keys=dup, inner=inner2,
        guess=3 * n)
    add("join_1n_scan_limit", both, one_to_many=1, join_keys=dup, inner=inner2, scan_limit=1777)
    add("join_1n_buffer_full", [(1, 0), (1, 1)], one_to_many=1, join_keys=dup, inner=inner2, guess=500, expect_error=-1)
    add("join_1n_outer_columns_only", [(1, 0), (2, 0)], [Qual(2, capi.LT, 500)], one_to_many=1, join_keys=dup, inner=inner2, guess=3 * n)
    return cases

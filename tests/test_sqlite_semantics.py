"""SQLite as a second, fully independent oracle — the method of the reference's own ExecuteTest
(Tests/ExecuteTest.cpp `c(query, dt)` runs every query on HeavyDB and on SQLite and compares;
SURVEY §8c item 4).  Every case of the matrix (tests/cases.py) is translated from its
RelAlgExecutionUnit into SQL, the decoded column values go into an in-memory SQLite table (SQL NULL
for the inline NULL sentinels), and the rows SQLite returns are compared with the rows the oracle's
step + ResultSet iteration produce: integers and NULLs exactly, doubles to 1e-9 (FLOAT arguments:
the single-precision tolerance of tests/helpers.py)."""
import math
import sqlite3

import numpy as np
import pytest

from heavydb_amd import capi
from tests import cases as cases_mod
from tests.helpers import F32_ATOL, F32_RTOL

CASES = [c for c in cases_mod.build_cases() if c.expect_error is None]

INT_NULL = {capi.INT8: -2**7, capi.INT16: -2**15, capi.INT32: -2**31, capi.INT64: -2**63}
DBL_NULL = float(np.finfo(np.float64).tiny)   # NULL_DOUBLE = DBL_MIN
FLT_NULL = float(np.finfo(np.float32).tiny)   # NULL_FLOAT = FLT_MIN
OPS = {capi.EQ: "=", capi.NE: "<>", capi.LT: "<", capi.GT: ">", capi.LE: "<=", capi.GE: ">="}


def _decoded_values(d, a):
    """Python values (None = SQL NULL) of a column chunk: DecodersImpl.h restated with numpy."""
    a = np.asarray(a)
    if d.type == capi.DOUBLE:
        return [None if (d.nullable and x == DBL_NULL) else float(x) for x in a]
    if d.type == capi.FLOAT:
        return [None if (d.nullable and np.float32(x) == np.float32(FLT_NULL)) else float(x) for x in a]
    v = a.astype(np.int64)
    null = (v == INT_NULL[d.type]) if d.nullable else np.zeros(len(v), bool)
    if d.encoding == capi.ENC_DICT and d.type in (capi.INT8, capi.INT16):
        bits = 8 if d.type == capi.INT8 else 16
        v = v & ((1 << bits) - 1)                       # unsigned ids
        null = (v == (1 << bits) - 1) if d.nullable else null
    elif d.encoding == capi.ENC_DATE_IN_DAYS:
        null = v == INT_NULL[d.type]
        v = v * 86400
    return [None if n else int(x) for x, n in zip(v, null)]


def _lit(col_type, q):
    return repr(float(q.literal)) if col_type in (capi.DOUBLE, capi.FLOAT) else str(int(q.literal))


def _expr_sql(e, descs, prior=()):
    """A projected expression (postfix micro-ops) as SQLite text over the base table's columns: integer
    arithmetic is 64-bit in SQLite (the cases keep clear of overflow), CAST(.. AS REAL) for the casts to
    DOUBLE / FLOAT (the FLOAT cases use values single precision holds exactly), ROUND() — half away from
    zero, like DEF_ROUND_NULLABLE — for floating point -> integer."""
    st = []
    for n in e.nodes:
        if n.op == capi.EX_COL and n.arg >= len(descs):   # the value of an earlier expression of the plan: its text
            j = n.arg - len(descs)
            st.append((_expr_sql(prior[j], descs, prior[:j]), prior[j].result(descs, prior[:j])[0] in (capi.DOUBLE, capi.FLOAT)))
        elif n.op == capi.EX_COL:
            st.append((f"c{n.arg}", descs[n.arg].type in (capi.DOUBLE, capi.FLOAT)))
        elif n.op == capi.EX_LIT:
            fp = n.type in (capi.DOUBLE, capi.FLOAT)
            st.append(("NULL" if n.null_lit else repr(float(n.flit)) if fp else str(int(n.ilit)), fp))
        elif n.op == capi.EX_CASE:   # stack: ELSE, THEN, condition
            (c, _), (t, fp), (e, _) = st.pop(), st.pop(), st.pop()
            st.append((f"(CASE WHEN {c} THEN {t} ELSE {e} END)", fp))
        elif capi.EX_EQ <= n.op <= capi.EX_GE:
            (b, _), (a, _) = st.pop(), st.pop()
            sym = {capi.EX_EQ: "=", capi.EX_NE: "<>", capi.EX_LT: "<", capi.EX_LE: "<=", capi.EX_GT: ">", capi.EX_GE: ">="}[n.op]
            st.append((f"({a} {sym} {b})", False))
        elif n.op == capi.EX_NOT:
            x, _ = st.pop()
            st.append((f"(NOT {x})", False))
        elif n.op in (capi.EX_AND, capi.EX_OR):
            (b, _), (a, _) = st.pop(), st.pop()
            st.append((f"({a} {'AND' if n.op == capi.EX_AND else 'OR'} {b})", False))
        elif n.op == capi.EX_IS_NULL:
            x, _ = st.pop()
            st.append((f"({x} IS NULL)", False))
        elif n.op == capi.EX_UMINUS:
            x, fp = st.pop()
            st.append((f"(-{x})", fp))
        elif n.op == capi.EX_CAST:
            x, fp = st.pop()
            to_fp = n.type in (capi.DOUBLE, capi.FLOAT)
            if to_fp:
                st.append((f"CAST({x} AS REAL)", True))
            elif fp:
                st.append((f"CAST(ROUND({x}) AS INTEGER)", False))
            else:
                st.append((x, False))
        else:
            (b, _), (a, fp) = st.pop(), st.pop()
            sym = {capi.EX_ADD: "+", capi.EX_SUB: "-", capi.EX_MUL: "*", capi.EX_DIV: "/", capi.EX_MOD: "%"}[n.op]
            st.append((f"({a} {sym} {b})", fp))
    return st[0][0]


def _sql_for(case):
    ra = case.ra
    descs = ra.input_col_descs
    join_cols = ra.join_outer_col if isinstance(ra.join_outer_col, (list, tuple)) else \
        ([ra.join_outer_col] if ra.join_outer_col >= 0 else [])

    def col(t):
        return f"d.i{t.col}" if t.table else f"f.c{t.col}"

    def cond(q):
        if q.op in (capi.IS_NULL, capi.IS_NOT_NULL):
            return f"f.c{q.col} IS {'NOT ' if q.op == capi.IS_NOT_NULL else ''}NULL"
        return f"f.c{q.col} {OPS[q.op]} {_lit(ra.col_type(q.col), q)}"
    sel = []
    for t in ra.target_exprs:
        if t.agg == capi.PROJECT:   # a Projection step: the column itself, of either side of the join
            sel.append(col(t))
        elif t.agg == capi.PROJECT_KEY:
            sel.append(f"f.c{ra.groupby_exprs[max(t.col, 0)]}")
        elif t.agg == capi.COUNT:
            sel.append("COUNT(*)" if t.col < 0 else f"COUNT({col(t)})")
        elif t.agg == capi.COUNT_IF:
            sel.append(f"COUNT(CASE WHEN {cond(t.cond)} THEN 1 END)")
        elif t.agg == capi.SUM_IF:
            e = f"SUM(CASE WHEN {cond(t.cond)} THEN {col(t)} END)"
            # the reference's own SumIf test (ExecuteTest.cpp:4131-4199) groups by nullable columns
            # only; over a NOT NULL argument a grouped SUM_IF starts at 0 like SUM
            # (OutputBufferInitialization.cpp:140, set_notnull only for non-grouped :281-286), so a
            # group without a qualifying row reads 0 where SQL says NULL
            arg_d = ra.inner_col_descs[t.col] if t.table else descs[t.col]
            left_inner = t.table and ra.join_kind == capi.JOIN_LEFT
            # ... and so does an argument constrained by a qual `arg IS NOT NULL` (constrained_not_null
            # -> set_notnull(target, true), OutputBufferInitialization.cpp:287)
            constrained = (not t.table) and any(q.op == capi.IS_NOT_NULL and q.col == t.col and not q.or_group for q in ra.simple_quals)
            if ra.groupby_exprs and (not arg_d.nullable or constrained) and not left_inner:
                e = f"COALESCE({e}, 0)"
            sel.append(e)
        else:
            fn = {capi.SUM: "SUM", capi.AVG: "AVG", capi.MIN: "MIN", capi.MAX: "MAX"}[t.agg]
            sel.append(f"{fn}({col(t)})")
    sql = "SELECT " + ", ".join(sel) + " FROM f"
    if join_cols:
        on = " AND ".join(f"f.c{c} = d.k{i}" for i, c in enumerate(join_cols))
        sql += (" LEFT JOIN" if ra.join_kind == capi.JOIN_LEFT else " JOIN") + f" d ON {on}"
    if ra.simple_quals:
        parts = [cond(q) for q in ra.simple_quals if not q.or_group]
        for g in sorted({q.or_group for q in ra.simple_quals if q.or_group}):
            parts.append("(" + " OR ".join(cond(q) for q in ra.simple_quals if q.or_group == g) + ")")
        sql += " WHERE " + " AND ".join(parts)
    if ra.groupby_exprs:
        sql += " GROUP BY " + ", ".join(f"f.c{g}" for g in ra.groupby_exprs)
    return sql


def _load(case):
    db = sqlite3.connect(":memory:")
    ra = case.ra
    n_cols = len(ra.input_col_descs)
    # expressions: f is a view that adds one computed column per expression (c<n_cols + k>) to the base table
    base = "fb" if ra.exprs else "f"
    db.execute(f"CREATE TABLE {base} (" + ", ".join(f"c{i}" for i in range(n_cols)) + ")")
    for cols in case.frags:
        vals = [_decoded_values(d, a) for d, a in zip(ra.input_col_descs, cols)]
        db.executemany(f"INSERT INTO {base} VALUES (" + ",".join("?" * n_cols) + ")", list(zip(*vals)))
    if ra.exprs:
        db.execute("CREATE VIEW f AS SELECT " + ", ".join(f"c{i}" for i in range(n_cols)) + ", " +
                   ", ".join(f"{_expr_sql(e, ra.input_col_descs, ra.exprs[:k])} AS c{n_cols + k}" for k, e in enumerate(ra.exprs)) +
                   " FROM fb")
    if case.join_keys is not None:
        keys = case.join_keys if isinstance(case.join_keys, (list, tuple)) else [case.join_keys]
        ktypes = case.join_key_type if isinstance(case.join_key_type, (list, tuple)) else [case.join_key_type]
        knull = case.join_key_nullable if isinstance(case.join_key_nullable, (list, tuple)) else [case.join_key_nullable] * len(keys)
        from heavydb_amd.executor import InputColDescriptor
        kv = [_decoded_values(InputColDescriptor(t, bool(nl)), k) for k, t, nl in zip(keys, ktypes, knull)]
        iv = [_decoded_values(d, a) for d, a in zip(ra.inner_col_descs, case.inner)]
        names = [f"k{i}" for i in range(len(kv))] + [f"i{i}" for i in range(len(iv))]
        db.execute("CREATE TABLE d (" + ", ".join(names) + ")")
        db.executemany("INSERT INTO d VALUES (" + ",".join("?" * len(names)) + ")", list(zip(*(kv + iv))))
    return db


def _oracle_rows(oracle, case, q, buf):
    iv, dv, nu = oracle.fetch_rows(q, buf)
    rows = []
    for r in range(iv.shape[0]):
        row = []
        for t in range(q.n_targets):
            if nu[r, t]:
                row.append(None)
            elif q.target_is_fp[t]:
                row.append(float(dv[r, t]))
            else:
                row.append(int(iv[r, t]))
        rows.append(tuple(row))
    return rows


def _key(row):
    """integers (and NULL flags) first, then the doubles at full precision: rows that tie on every
    integer are far further apart than the tolerance"""
    ints = tuple((0, 0) if v is None else (1, v) for v in row if not isinstance(v, float))
    flts = tuple(v for v in row if isinstance(v, float))
    return ints, flts


def _check_case(oracle, case, no_nulls_in_data=False):
    from tests.test_rowlogic_emu import _oracle_join
    plan = case.ra.to_plan()
    try:
        q, buf, code = oracle.execute(plan, case.frags, case.inner, _oracle_join(oracle, case), n_threads=2)
    except capi.Mi355qError:
        return "rejected"
    if code != 0:
        return "error"
    # the target whose slot tells an empty entry from a group (AVG: its COUNT slot, one after the slot it is listed at)
    key_t = [t for t in range(q.n_targets) if q.keyless and
             (q.target_slot[t] == q.idx_target_as_key or
              (q.target_agg[t] == capi.AVG and q.target_slot[t] == q.idx_target_as_key - 1))]
    if key_t and q.target_skip_null[key_t[0]] and not no_nulls_in_data:
        return "keyless-null-aware"   # (without NULLs in the data no group can look empty)
    got = sorted(_oracle_rows(oracle, case, q, buf), key=_key)
    sql = _sql_for(case)
    fp = [bool(q.target_is_fp[t]) for t in range(q.n_targets)]
    want = sorted((tuple(float(v) if f and v is not None else v for v, f in zip(r, fp))
                   for r in _load(case).execute(sql).fetchall()), key=_key)
    assert len(want) == len(got), (sql, len(want), len(got))
    for w, g in zip(want, got):
        for t, (a, b) in enumerate(zip(w, g)):
            if a is None or b is None:
                assert a is None and b is None, (sql, t, w, g)
            elif isinstance(b, float) or isinstance(a, float):
                rt, at = (F32_RTOL, F32_ATOL) if q.target_arg_is_f32[t] else (max(case.fp_rtol, 1e-9), 1e-9)
                assert math.isclose(float(a), b, rel_tol=rt, abs_tol=at), (sql, t, w, g)
            else:
                assert a == b, (sql, t, w, g)
    return "ok"


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_oracle_agrees_with_sqlite(oracle, case):
    r = _check_case(oracle, case)
    if r == "keyless-null-aware":
        pytest.skip("keyless key target is NULL-aware: the reference hides all-NULL groups (DESIGN §2)")
    assert r == "ok", r


def test_random_plans_agree_with_sqlite(oracle):
    """The random tables x plans of tests/test_plan_fuzz.py (encoded columns, 0-3 group columns,
    every aggregate kind, a qual) against SQLite."""
    from tests.cases import Case
    from tests.test_plan_fuzz import _fuzz_row_plan, _fuzz_table
    rng = np.random.default_rng(20260922)
    tally = {}
    for i in range(250):
        n_rows = int(rng.integers(1, 300))
        descs, cols = _fuzz_table(rng, n_rows)
        ra = _fuzz_row_plan(rng, descs)
        cut = n_rows // 2
        r = _check_case(oracle, Case(f"sqlfuzz{i}", ra, [[c[:cut] for c in cols], [c[cut:] for c in cols]]))
        tally[r] = tally.get(r, 0) + 1
    assert tally.get("ok", 0) > 150, tally


def test_random_joins_agree_with_sqlite(oracle):
    """Random joins (1-3 key components, duplicates and NULL keys on both sides, INNER / LEFT, grouped
    or not) against SQLite's JOIN / LEFT JOIN."""
    from tests.test_plan_fuzz import _fuzz_join
    rng = np.random.default_rng(424242)
    tally = {}
    for i in range(150):
        r = _check_case(oracle, _fuzz_join(rng))
        tally[r] = tally.get(r, 0) + 1
    assert tally.get("ok", 0) > 120, tally


def test_benchmark_shapes_at_a_million_rows_agree_with_sqlite(oracle):
    """The shapes BASELINE.json benchmarks (filtered baseline GROUP BY, perfect-hash GROUP BY, filtered
    scan, join probe + SUM, one-to-many LEFT join) at 1 M rows: oracle == SQLite.  The GPU leg
    (tests/test_zz_gpu_sqlite_scale.py) runs the same cases through the kernel families."""
    from tests.test_zz_gpu_sqlite_scale import SHAPES
    for name, _, case in SHAPES:
        assert _check_case(oracle, case) == "ok", name


# ---- Projection steps (round 6): every joined row is one output entry — the oracle's restatement of the join
# loop nest (IRCodegen.cpp buildJoinLoops: no reference-RUN vector exists for it, the loop is LLVM-generated) is pinned here
# by the arbiter the reference itself uses: the row MULTISET of SQLite's JOIN / LEFT JOIN over the same tables
def projection_rows_sqlite(case):
    """the rows SQLite returns for a Projection case, sorted (order within a projection without ORDER BY is unspecified)"""
    return sorted(_load(case).execute(_sql_for(case)).fetchall(), key=_key)


def rows_agree(case, q, want, got):
    assert len(want) == len(got), (len(want), len(got))
    for w, g in zip(want, got):
        for t, (a, b) in enumerate(zip(w, g)):
            if a is None or b is None:
                assert a is None and b is None, (t, w, g)
            elif isinstance(b, float) or isinstance(a, float):
                rt, at = (F32_RTOL, F32_ATOL) if q.target_arg_is_f32[t] else (1e-12, 0.0)
                assert math.isclose(float(a), b, rel_tol=rt, abs_tol=at), (t, w, g)
            else:
                assert a == b, (t, w, g)


def _projection_cases():
    from tests import proj_cases
    out = []
    for c in proj_cases.build_join_cases() + proj_cases.build_cases():
        if c.expect_error is not None or c.ra.scan_limit:     # (a LIMIT keeps the first n in (fragment, row) order: SQLite's pick differs)
            continue
        if any(d.encoding for d in c.ra.input_col_descs):     # (encoded inputs: covered by the aggregate cases above)
            continue
        out.append(c)
    return out


PROJ_CASES = _projection_cases()


@pytest.mark.parametrize("case", PROJ_CASES, ids=[c.name for c in PROJ_CASES])
def test_projection_rows_agree_with_sqlite(oracle, case):
    from tests.test_hostsim_flow import _oracle_join
    q, buf, code = oracle.execute(case.ra.to_plan(), case.frags, case.inner, _oracle_join(oracle, case) if case.join_keys is not None else None)
    assert code == 0
    got = sorted(_oracle_rows(oracle, case, q, buf), key=_key)
    fp = [bool(q.target_is_fp[t]) for t in range(q.n_targets)]
    want = sorted((tuple(float(v) if f and v is not None else v for v, f in zip(r, fp)) for r in projection_rows_sqlite(case)), key=_key)
    if case.join_keys is not None:
        assert len(want) > 0 or "nothing_matches" in case.name
    rows_agree(case, q, want, got)

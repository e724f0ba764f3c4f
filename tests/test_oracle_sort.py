"""The oracle's ORDER BY (orc_sort: ResultSet::sort + ResultSetComparator restated) against SQLite's
ORDER BY ... [ASC | DESC] NULLS FIRST | LAST — the reference's own method for its ORDER BY tests
(ExecuteTest `c(query, dt)`).  Cases of the matrix with at least two targets, random order entries."""
import sqlite3

import numpy as np
import pytest

from heavydb_amd import capi
from tests import cases as cases_mod
from tests.test_sqlite_semantics import _load, _oracle_rows, _sql_for

NAMES = ["perfect_minmax_f64_i64", "perfect_filtered", "baseline_nullable_args", "perfect_nullable_key_and_args",
         "constrained_baseline_avg_min", "isnull_grouped_baseline", "baseline_count_avg"]
CASES = [c for c in cases_mod.build_cases() if c.name in NAMES]


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_oracle_sort_agrees_with_sqlite(oracle, case):
    from tests.test_rowlogic_emu import _oracle_join
    rng = np.random.default_rng(abs(hash(case.name)) % 2**32)
    q, buf, code = oracle.execute(case.ra.to_plan(), case.frags, case.inner, _oracle_join(oracle, case), n_threads=2)
    assert code == 0
    rows = _oracle_rows(oracle, case, q, buf)          # in entry order
    sql = _sql_for(case)
    db = _load(case)
    for trial in range(6):
        n_order = int(rng.integers(1, min(3, q.n_targets) + 1))
        targets = [int(x) for x in rng.choice(q.n_targets, size=n_order, replace=False)]
        order = [(t, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))) for t in targets]
        limit = int(rng.choice([0, 7, 100]))
        perm = oracle.sort(q, buf, order, limit=limit)
        # rows of the sorted permutation: _oracle_rows iterates in entry order, and the live entries in
        # entry order are the (index-sorted) output of an unlimited sort
        all_live = np.sort(oracle.sort(q, buf, [(0, False, False)], limit=0))
        pos = {int(e): i for i, e in enumerate(all_live)}
        got = [rows[pos[int(e)]] for e in perm]
        ob = ", ".join(f"{t + 1} {'DESC' if d else 'ASC'} NULLS {'FIRST' if nf else 'LAST'}" for t, d, nf in order)
        want = db.execute(sql + " ORDER BY " + ob + (f" LIMIT {limit}" if limit else "")).fetchall()
        assert len(want) == len(got)
        # compare the ORDER columns position by position (ties among the other columns are free)
        for w, g in zip(want, got):
            for t, _, _ in order:
                a, b = w[t], g[t]
                if a is None or b is None:
                    assert a is None and b is None, (ob, w, g)
                else:
                    assert abs(float(a) - float(b)) <= 1e-9 * max(1.0, abs(float(a))), (ob, w, g)


# ---- Select.GpuSort / Select.SpeculativeTopNSort (ExecuteTest.cpp:11338-11378) on the reference's gpu_sort_test
# table (import_gpu_sort_test, :9583-9603: four rows (2, 2, 2, 2), six rows (16000, 16000, 16000, 127),
# fragment_size = 2)
GPU_SORT_COLS = {"x": capi.INT64, "y": capi.INT32, "z": capi.INT16, "t": capi.INT8}
GPU_SORT_ROWS = [(2, 2, 2, 2)] * 4 + [(16000, 16000, 16000, 127)] * 6
# (the reference's SQL, group column, targets after the key, ORDER BY target index, DESC?, LIMIT, projected columns)
GPU_SORT_QUERIES = [
    ("SELECT x, COUNT(*) AS val FROM gpu_sort_test GROUP BY x ORDER BY val DESC;", "x", 1, 1, True, 0, None),
    ("SELECT y, COUNT(*) AS val FROM gpu_sort_test GROUP BY y ORDER BY val DESC;", "y", 1, 1, True, 0, None),
    ("SELECT y, COUNT(*), COUNT(*) AS val FROM gpu_sort_test GROUP BY y ORDER BY val DESC;", "y", 2, 2, True, 0, None),
    ("SELECT z, COUNT(*) AS val FROM gpu_sort_test GROUP BY z ORDER BY val DESC;", "z", 1, 1, True, 0, None),
    ("SELECT t, COUNT(*) AS val FROM gpu_sort_test GROUP BY t ORDER BY val DESC;", "t", 1, 1, True, 0, None),
    ("SELECT x, COUNT(*) AS val FROM gpu_sort_test GROUP BY x ORDER BY val DESC LIMIT 2;", "x", 1, 1, True, 2, None),
    ("SELECT x from (SELECT COUNT(*) AS val, x FROM gpu_sort_test GROUP BY x ORDER BY val ASC LIMIT 3);", "x", 1, 1, False, 3, [0]),
    ("SELECT val from (SELECT y, COUNT(*) AS val FROM gpu_sort_test GROUP BY y ORDER BY val DESC LIMIT 3);", "y", 1, 1, True, 3, [1]),
]


def gpu_sort_unit(col, n_counts):
    from heavydb_amd.executor import InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    from tests.cases import col_range
    t = GPU_SORT_COLS[col]
    i = list(GPU_SORT_COLS).index(col)
    npt = {capi.INT64: np.int64, capi.INT32: np.int32, capi.INT16: np.int16, capi.INT8: np.int8}[t]
    a = np.array([r[i] for r in GPU_SORT_ROWS], dtype=npt)
    ra = RelAlgExecutionUnit([InputColDescriptor(t, True, col_range([a], t, True))],
                             [TargetExpr(capi.PROJECT_KEY, 0)] + [TargetExpr(capi.COUNT)] * n_counts, [], [0],
                             num_tuples=len(a))
    return ra, [[a[j:j + 2]] for j in range(0, len(a), 2)]


def gpu_sort_db():
    db = sqlite3.connect(":memory:")
    db.execute("CREATE TABLE gpu_sort_test (x bigint, y int, z smallint, t tinyint)")
    db.executemany("INSERT INTO gpu_sort_test VALUES (?, ?, ?, ?)", GPU_SORT_ROWS)
    return db


@pytest.mark.parametrize("qi", range(len(GPU_SORT_QUERIES)))
def test_select_gpu_sort_and_speculative_top_n(oracle, qi):
    sql, col, n_counts, order_t, desc, limit, project = GPU_SORT_QUERIES[qi]
    ra, frags = gpu_sort_unit(col, n_counts)
    q, buf, code = oracle.execute(ra.to_plan(), frags, n_threads=2)
    assert code == 0
    iv, dv, nu = oracle.fetch_rows(q, buf)
    all_live = np.sort(oracle.sort(q, buf, [(0, False, False)], limit=0))
    pos = {int(e): i for i, e in enumerate(all_live)}
    perm = oracle.sort(q, buf, [(order_t, desc, False)], limit=limit)
    got = [tuple(int(v) for v in iv[pos[int(e)]]) for e in perm]
    if project is not None:
        got = [tuple(r[c] for c in project) for r in got]
    want = [tuple(r) for r in gpu_sort_db().execute(sql).fetchall()]
    assert got == want, (sql, got, want)   # no ties in this table: the order is total

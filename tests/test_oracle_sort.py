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

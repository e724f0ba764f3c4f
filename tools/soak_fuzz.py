#!/usr/bin/env python
"""Soak run of the CPU fuzzers with fresh seeds (the committed tests use fixed seeds): random plans
and joins through the product's row logic (host emulation) vs the oracle vs SQLite, row-wise and
columnar.  usage: soak_fuzz.py <first_seed> <n_seeds>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heavydb_amd import capi  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.cases import Case  # noqa: E402
from tests.helpers import columnar_to_rows, compare_buffers, qmd_equal, rowwise_qmd  # noqa: E402
from tests.test_plan_fuzz import _fuzz_join, _fuzz_row_plan, _fuzz_table  # noqa: E402
from tests.test_rowlogic_emu import _emu_execute, _oracle_join  # noqa: E402
from tests.test_sqlite_semantics import _check_case  # noqa: E402


def one(case, hint):
    case.ra.output_columnar_hint = hint
    plan = case.ra.to_plan()
    oj = _oracle_join(oracle, case)
    try:
        q, want, code = oracle.execute(plan, case.frags, case.inner, oj, n_threads=2)
    except capi.Mi355qError:
        return "rejected"
    eq, got, ecode = _emu_execute(case, plan, oj)
    if code or ecode:
        assert code and ecode, (code, ecode)
        return "error"
    qmd_equal(q, eq)
    if q.output_columnar:
        compare_buffers(rowwise_qmd(q), columnar_to_rows(q, want), columnar_to_rows(q, got), 1e-9)
    else:
        compare_buffers(q, want, got, 1e-9)
    return "ok"


def main():
    first, n = int(sys.argv[1]), int(sys.argv[2])
    tally = {}
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        for i in range(120):
            n_rows = int(rng.integers(1, 400))
            descs, cols = _fuzz_table(rng, n_rows)
            ra = _fuzz_row_plan(rng, descs)
            cut = n_rows // 2
            case = Case(f"soak{seed}_{i}", ra, [[c[:cut] for c in cols], [c[cut:] for c in cols]])
            for hint in (0, 1):
                try:
                    r = one(case, hint)
                except AssertionError:
                    print("MISMATCH emu/oracle seed", seed, "iter", i, "hint", hint)
                    raise
                tally[r] = tally.get(r, 0) + 1
            case.ra.output_columnar_hint = 0
            r = _check_case(oracle, case)
            tally["sql_" + r] = tally.get("sql_" + r, 0) + 1
        for i in range(60):
            case = _fuzz_join(rng)
            for hint in (0, 1):
                r = one(case, hint)
                tally["join_" + r] = tally.get("join_" + r, 0) + 1
            case.ra.output_columnar_hint = 0
            r = _check_case(oracle, case)
            tally["joinsql_" + r] = tally.get("joinsql_" + r, 0) + 1
        print("seed", seed, tally, flush=True)


if __name__ == "__main__":
    main()

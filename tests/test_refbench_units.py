"""The execution units tools/refbench.py builds for the reference's synthetic benchmark
(Benchmarks/synthetic_benchmark/queries/*/*.sql on create_table.py's schema) at a small size on the CPU: the oracle
against SQLite running the query's aggregation step as SQL text, and against the product's row logic (host
emulation).  The GPU leg (tests/test_zz_gpu_refbench.py) runs the same units through the library."""
import os
import sys

import numpy as np
import pytest

from heavydb_amd import capi
from tests.cases import Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import refbench  # noqa: E402

N_ROWS = 30_000
CARD_CAP = 3_000     # x10k .. x10m are capped so that the perfect-hash tables stay small here


def small_table(oracle, n_rows=N_ROWS, card_cap=CARD_CAP):
    names, descs, gens = refbench.schema(card_cap)
    cols = [oracle.generate_column(n_rows, g[0], g[1], g[2], g[3], g[4], g[5]) for g in gens]
    cut = n_rows // 3
    return names, descs, [[c[:cut] for c in cols], [c[cut:] for c in cols]]


QUERIES = refbench.queries()


@pytest.mark.parametrize("name", list(QUERIES), ids=list(QUERIES))
def test_refbench_unit_oracle_sqlite_product(oracle, name):
    from tests.test_rowlogic_emu import _emu_execute
    from tests.test_sqlite_semantics import _check_case
    from tests.helpers import compare_buffers, qmd_equal
    names, descs, frags = small_table(oracle)
    ra, _ = refbench.build_unit(QUERIES[name], names, descs, N_ROWS)
    case = Case("refbench_" + name, ra, frags)
    assert _check_case(oracle, case, no_nulls_in_data=True) == "ok"
    plan = ra.to_plan()
    q, want, code = oracle.execute(plan, frags, n_threads=2)
    eq, got, ecode = _emu_execute(case, plan, None)
    assert code == 0 and ecode == 0
    qmd_equal(q, eq)
    compare_buffers(q, want, got, 1e-9)

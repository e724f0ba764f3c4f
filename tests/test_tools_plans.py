"""The plans tools/bool_filter_bench.py builds (filters that are BOOLEAN expressions: AND inside OR, NOT over a disjunction, a
division guarded by a short-circuit AND, a root that reads earlier expressions) on a small table: oracle vs the product's row
logic — so that the tool measures correct steps when it meets a GPU."""
import numpy as np

from heavydb_amd import capi
from heavydb_amd.executor import Expr, ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
from tests.cases import Case
from tests.helpers import compare_buffers


def test_bool_filter_bench_plans(oracle):
    from tests.test_rowlogic_emu import _emu_execute
    from tools.bool_filter_bench import shapes
    rng = np.random.default_rng(3)
    n = 5000
    cols = [rng.integers(0, 1000, n).astype(np.int32)] + [rng.integers(0, 1_000_000, n).astype(np.int32) for _ in range(4)]
    cols[3][::7] = 0                      # zero divisors: the guarded division must not raise
    cols[4][::11] = np.iinfo(np.int32).min   # NULLs in the nullable column
    descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 999))] + \
            [InputColDescriptor(capi.INT32, i == 4, ExpressionRange(True, 0, 999_999, i == 4)) for i in range(1, 5)]
    seen = {}
    for name, exprs, quals, reads in shapes(capi, Expr, Qual):
        xs = [e.with_range(ExpressionRange(True, 0, 1, True)) for e in exprs]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1)], quals, [0],
                                 exprs=xs, num_tuples=n)
        plan = ra.to_plan()
        frags = [[c[:n // 2] for c in cols], [c[n // 2:] for c in cols]]
        q, want, code = oracle.execute(plan, frags, n_threads=2)
        eq, ebuf, ecode = _emu_execute(Case(name, ra, frags), plan, None)
        assert code == 0 and ecode == 0, (name, code, ecode)
        compare_buffers(q, want, ebuf)
        seen[name] = oracle.row_count(q, want)
    assert set(seen) == {"plain", "and_in_or", "not_or", "guarded_div", "composed", "sum_gt", "col_lt_col", "affine"} and all(v > 100 for v in seen.values()), seen

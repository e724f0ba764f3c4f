"""The kernel FAMILIES (not just the row kernel) against SQLite at a million rows: partitioned
baseline-hash GROUP BY, LDS perfect-hash GROUP BY, the filtered scan, the one-to-one join probe with
SUM, and a one-to-many LEFT join — the reference's ExecuteTest method (HeavyDB result == SQLite
result) applied to the shapes BASELINE.json benchmarks."""
import math

import numpy as np
import pytest

from heavydb_amd import capi
from heavydb_amd.executor import ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
from tests.cases import Case
from tests.test_gpu_parity import _build_join, _fetch_result, _upload, torch_cuda  # noqa: F401
from tests.test_sqlite_semantics import _key, _load, _sql_for

pytestmark = pytest.mark.gpu
V = ExpressionRange
N = 1_000_000


def _shapes():
    rng = np.random.default_rng(2026)
    cut = [0, N // 2 + 16, N]          # two 16-byte-aligned fragments
    key = (rng.integers(0, 30_000, N) * 1000003 + 7).astype(np.int64)
    val = rng.random(N) * 1000.0
    fil = rng.integers(0, 2**31 - 1, N).astype(np.int32)
    k32 = rng.integers(0, 1000, N).astype(np.int32)
    v64 = rng.integers(-500_000, 500_001, N).astype(np.int64)
    dim_k = np.arange(50_000, dtype=np.int64)
    dim_w = rng.integers(-1000, 1001, 50_000).astype(np.int64)
    fk = rng.integers(-100, 60_000, N).astype(np.int64)
    dup_k = rng.integers(0, 20_000, 40_000).astype(np.int64)   # duplicates: one-to-many
    dup_w = rng.integers(0, 100, 40_000).astype(np.int64)

    def frags(cols):
        return [[c[cut[i]:cut[i + 1]] for c in cols] for i in range(2)]
    d = InputColDescriptor
    out = []
    out.append(("baseline_partitioned", 2, Case("s0", RelAlgExecutionUnit(
        [d(capi.INT64, False, V(True, 7, 29_999 * 1000003 + 7)), d(capi.DOUBLE, False, V(True, 0, 0, False, 0.0, 1000.0)),
         d(capi.INT32, False, V(True, 0, 2**31 - 1))],
        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1)], [Qual(2, capi.LT, 2**30)], [0],
        max_groups_buffer_entry_guess=65536), frags([key, val, fil]))))
    out.append(("perfect_lds", 0, Case("s1", RelAlgExecutionUnit(
        [d(capi.INT32, False, V(True, 0, 999)), d(capi.INT64, False, V(True, -500_000, 500_000))],
        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 1)], [], [0]), frags([k32, v64]))))
    out.append(("scan_count", 0, Case("s2", RelAlgExecutionUnit(
        [d(capi.INT32, False, V(True, 0, 2**31 - 1))], [TargetExpr(capi.COUNT)], [Qual(0, capi.LT, 2**30)]), frags([fil]))))
    out.append(("join_sum", 0, Case("s3", RelAlgExecutionUnit(
        [d(capi.INT64, False, V(True, -100, 59_999)), d(capi.INT64, False, V(True, -500_000, 500_000))],
        [TargetExpr(capi.SUM, 1), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1, 1)],
        inner_col_descs=[d(capi.INT64, False, V(True, 0, 49_999)), d(capi.INT64, False, V(True, -1000, 1000))],
        join_outer_col=0), frags([fk, v64]), [dim_k, dim_w], dim_k, capi.INT64, V(True, 0, 49_999))))
    out.append(("left_join_one_to_many_grouped", 0, Case("s4", RelAlgExecutionUnit(
        [d(capi.INT64, False, V(True, -100, 59_999)), d(capi.INT32, False, V(True, 0, 999))],
        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1, 1), TargetExpr(capi.COUNT, 0, 1)],
        [], [1], inner_col_descs=[d(capi.INT64, False, V(True, 0, 19_999)), d(capi.INT64, False, V(True, 0, 99))],
        join_outer_col=0, join_kind=capi.JOIN_LEFT), frags([fk, k32]), [dup_k, dup_w], dup_k, capi.INT64,
        V(True, 0, 19_999), False, 1)))
    return out


SHAPES = _shapes()


@pytest.mark.parametrize("si", range(len(SHAPES)), ids=[s[0] for s in SHAPES])
def test_kernel_families_agree_with_sqlite(torch_cuda, si):
    from heavydb_amd.executor import Executor
    name, variant, case = SHAPES[si]
    frag_t, inner_t = _upload(torch_cuda, case)
    hj, keep = _build_join(torch_cuda, case)
    case.ra.join_table = hj
    try:
        rs = Executor(0).executeWorkUnit(case.ra, _fetch_result(case, frag_t, inner_t), allow_retry=False,
                                         kernel_variant=variant)
        q = rs.getQueryMemDesc()
        iv, dv, nu = rs.fetch()
        kernel = rs.report.kernel_name.decode()
    finally:
        case.ra.join_table = None
    fp = [bool(q.target_is_fp[t]) for t in range(q.n_targets)]
    got = sorted((tuple(None if nu[r, t] else (float(dv[r, t]) if fp[t] else int(iv[r, t])) for t in range(q.n_targets))
                  for r in range(iv.shape[0])), key=_key)
    sql = _sql_for(case)
    want = sorted((tuple(float(v) if f and v is not None else v for v, f in zip(r, fp))
                   for r in _load(case).execute(sql).fetchall()), key=_key)
    assert len(got) == len(want), (name, kernel, sql, len(got), len(want))
    for w, g in zip(want, got):
        for t, (a, b) in enumerate(zip(w, g)):
            if a is None or b is None:
                assert a is None and b is None, (name, kernel, sql, t, w, g)
            elif fp[t]:
                assert math.isclose(a, b, rel_tol=1e-9, abs_tol=1e-9), (name, kernel, sql, t, w, g)
            else:
                assert a == b, (name, kernel, sql, t, w, g)

"""Payload probe (kernels_part.hip: k_join_payload -> k_part_scatter<DIRECT> -> k_part_probe ->
k_probe_finish): joins whose non-grouped targets read the inner side, one-to-many perfect tables (one
joined row per match) and LEFT joins, against the oracle's restatement of the reference's probe loops
(hash_join_idx, the one-to-many offsets | counts | payloads walk, the outer-join found flag:
JoinHashTableQueryRuntime.cpp:56-163, HashJoinRuntime.cpp:654-1110, IRCodegen.cpp buildJoinLoops).
Integer sums and counts: bit-exact."""
from __future__ import annotations

import numpy as np
import pytest

from heavydb_amd import capi
from tests.helpers import compare_buffers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.load_library()
    return torch


@pytest.mark.parametrize("sparse", [False, True], ids=["perfect", "keyed"])
@pytest.mark.parametrize("one_to_many", [False, True])
@pytest.mark.parametrize("left", [False, True])
@pytest.mark.parametrize("shape", ["all_targets", "inner_only", "hot_key", "inner_nulls", "no_inner_col"])
def test_payload_probe_matches_oracle(torch_cuda, oracle, one_to_many, left, shape, sparse, keyed_passes=0):
    from heavydb_amd.executor import (Executor, ExpressionRange, FetchResult, HashJoin, InputColDescriptor,
                                      RelAlgExecutionUnit, TargetExpr)
    torch = torch_cuda
    rng = np.random.default_rng(11 + 2 * int(one_to_many) + int(left))
    n_keys = 400_000
    keys = np.sort(rng.choice(np.arange(5000, 5000 + 2 * n_keys, dtype=np.int64), n_keys, replace=False))
    mul = 1000003 if sparse else 1   # sparse keys: the range is too wide for a perfect table -> keyed (baseline) table
    keys = keys * mul
    if one_to_many:   # 1-4 inner rows per key, shuffled
        dim = rng.permutation(np.repeat(keys, rng.integers(1, 5, n_keys)))
    else:
        dim = rng.permutation(keys)
    m = len(dim)
    w = rng.integers(-10**6, 10**6, m).astype(np.int64)
    w_nullable = shape == "inner_nulls"
    if w_nullable:
        w[rng.random(m) < 0.3] = -2**63
    dmin, dmax = int(dim.min()), int(dim.max())
    dk, dw = torch.from_numpy(dim).cuda(), torch.from_numpy(w).cuda()
    hj = HashJoin.getInstance(int(dk.data_ptr()), m, capi.INT64, ExpressionRange(True, dmin, dmax),
                              one_to_many=1 if one_to_many else 0)
    assert hj.info()["hash_type"] == (2 if one_to_many else 0) + (1 if sparse else 0)
    n = 3_000_000
    k = rng.integers(dmin // mul - 20000, dmax // mul + 20000, n).astype(np.int64) * mul   # about half of them match
    if shape == "hot_key":
        k[rng.random(n) < 0.4] = keys[777]
    v = rng.integers(-10**9, 10**9, n).astype(np.int64)
    v[rng.random(n) < 0.02] = -2**63   # skipped by the non-grouped SUM / COUNT(v)
    descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, dmin - 20000 * mul, dmax + 20000 * mul)),
             InputColDescriptor(capi.INT64, False, ExpressionRange(True, -10**9, 10**9))]
    inner = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, dmin, dmax)),
             InputColDescriptor(capi.INT64, w_nullable, ExpressionRange(True, -10**6, 10**6, w_nullable))]
    targets = {"all_targets": [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1), TargetExpr(capi.SUM, 1, 1),
                               TargetExpr(capi.COUNT, 1, 1)],
               "inner_only": [TargetExpr(capi.SUM, 1, 1)],
               "hot_key": [TargetExpr(capi.SUM, 1, 1), TargetExpr(capi.COUNT), TargetExpr(capi.COUNT, 1)],
               "inner_nulls": [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1, 1), TargetExpr(capi.COUNT, 1, 1),
                               TargetExpr(capi.SUM, 1)],
               "no_inner_col": [TargetExpr(capi.SUM, 1), TargetExpr(capi.COUNT)]}[shape]
    ra = RelAlgExecutionUnit(descs, targets, inner_col_descs=inner, join_outer_col=0, join_table=hj,
                             join_kind=capi.JOIN_LEFT if left else capi.JOIN_INNER)
    cuts = [0, n // 3 + 4, n]
    dev = [torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda()]
    fr = FetchResult([[int(t.data_ptr()) + cuts[i] * 8 for t in dev] for i in range(2)],
                     [cuts[i + 1] - cuts[i] for i in range(2)], [int(dk.data_ptr()), int(dw.data_ptr())], m,
                     keepalive=dev + [dk, dw])
    ex = Executor(0)
    rs = ex.executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=3, probe_keyed_passes=keyed_passes)
    assert rs.report.variant == 3 and rs.report.kernel_name.decode() == "k_part_scatter", \
        (rs.report.variant, rs.report.kernel_name)
    row = ex.executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=1)      # the row kernel / direct probe
    oj = oracle.OracleJoin(dim, capi.INT64, dmin, dmax, one_to_many=1 if one_to_many else 0)
    ra.join_table = None
    q, want, code = oracle.execute(ra.to_plan(), [[k[cuts[i]:cuts[i + 1]], v[cuts[i]:cuts[i + 1]]] for i in range(2)],
                                   [dim, w], oj, n_threads=2)
    ra.join_table = hj
    assert code == 0
    compare_buffers(q, want, rs.getStorage())
    assert np.array_equal(rs.getStorage(), row.getStorage())
    # a second call reuses the cached payload (and must not be confused by it)
    again = ex.executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=3, probe_keyed_passes=keyed_passes)
    assert np.array_equal(again.getStorage(), rs.getStorage())
    if shape == "all_targets" and not keyed_passes:
        # the inner column rewritten IN PLACE (same device pointer, new content): with a new inner_version the
        # cached per-key payload is rebuilt; mi355q_join_invalidate_payload does the same for callers that cannot
        # version their columns (a stale payload gave SUM(dim.w) of the old values)
        w2 = (w * 3 + 1).astype(np.int64)
        dw.copy_(torch.from_numpy(w2).cuda())
        torch.cuda.synchronize()
        ra.join_table = None
        q2, want2, code2 = oracle.execute(ra.to_plan(), [[k[cuts[i]:cuts[i + 1]], v[cuts[i]:cuts[i + 1]]] for i in range(2)],
                                          [dim, w2], oj, n_threads=2)
        ra.join_table = hj
        assert code2 == 0 and not np.array_equal(want2, want)
        fr.inner_version = 1
        fresh = ex.executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=3)
        compare_buffers(q2, want2, fresh.getStorage())
        info = hj.payload_info()
        assert info["bytes"] > 0 and info["inner_version"] == 1
        dw.copy_(torch.from_numpy(w).cuda())       # back to the first content under the SAME version ...
        torch.cuda.synchronize()
        hj.invalidate_payload()                    # ... so the cache has to be dropped explicitly
        back = ex.executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=3)
        compare_buffers(q, want, back.getStorage())


def test_payload_probe_sub_ranges_and_column_switch(torch_cuda, oracle):
    """A key range whose 16-byte-per-key slice does not fit LDS at 1024 partitions (two sub-ranges per
    partition), and two plans over the same join table that read different inner columns (the cached
    payload is rebuilt for the column in use)."""
    from heavydb_amd.executor import (Executor, ExpressionRange, FetchResult, HashJoin, InputColDescriptor,
                                      RelAlgExecutionUnit, TargetExpr)
    torch = torch_cuda
    rng = np.random.default_rng(99)
    m = 15_000_000                      # 14.6 K keys per partition x 16 B > 144 KB: R = 2
    dim = rng.permutation(np.arange(m, dtype=np.int64))
    w1 = rng.integers(-1000, 1000, m).astype(np.int64)
    w1[rng.random(m) < 0.1] = -2**63
    w2 = rng.integers(0, 50, m).astype(np.int64)
    dk, d1, d2 = (torch.from_numpy(a).cuda() for a in (dim, w1, w2))
    hj = HashJoin.getInstance(int(dk.data_ptr()), m, capi.INT64, ExpressionRange(True, 0, m - 1))
    n = 4_000_000
    k = rng.integers(-100, m + 100, n).astype(np.int64)
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, -100, m + 99)),
             InputColDescriptor(capi.INT64, False, ExpressionRange(True, -10**6, 10**6))]
    inner = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 0, m - 1)),
             InputColDescriptor(capi.INT64, True, ExpressionRange(True, -1000, 1000, True)),
             InputColDescriptor(capi.INT64, False, ExpressionRange(True, 0, 49))]
    dev = [torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda()]
    fr = FetchResult([[int(t.data_ptr()) for t in dev]], [n], [int(dk.data_ptr()), int(d1.data_ptr()), int(d2.data_ptr())],
                     m, keepalive=dev + [dk, d1, d2])
    ex = Executor(0)
    oj = oracle.OracleJoin(dim, capi.INT64, 0, m - 1)
    for col in (1, 2, 1):
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.SUM, col, 1), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1)],
                                 inner_col_descs=inner, join_outer_col=0, join_table=hj)
        rs = ex.executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=3)
        assert rs.report.variant == 3
        ra.join_table = None
        q, want, code = oracle.execute(ra.to_plan(), [[k, v]], [dim, w1, w2], oj, n_threads=2)
        assert code == 0
        compare_buffers(q, want, rs.getStorage())


@pytest.mark.parametrize("inner_nulls", [True, False])
@pytest.mark.parametrize("left", [False, True])
def test_payload_probe_l2_mode(torch_cuda, oracle, left, inner_nulls):
    """A key range too wide for LDS slices (32 M inner keys: 31 K keys per partition): the 16-byte-per-key
    payload slice of a partition stays in the XCD's L2 (k_part_probe_l2), inner column with NULLs."""
    from heavydb_amd.executor import (Executor, ExpressionRange, FetchResult, HashJoin, InputColDescriptor,
                                      RelAlgExecutionUnit, TargetExpr)
    torch = torch_cuda
    rng = np.random.default_rng(5)
    m = 32_000_000
    dim = np.arange(m, dtype=np.int64)
    dim[:1000] = rng.permutation(dim[:1000])
    w = rng.integers(-1000, 1000, m).astype(np.int64)
    if inner_nulls:        # 16-byte {sum, rows, non-NULL} entries; without NULLs the 8-byte value-or-absent entries
        w[::17] = -2**63
    dk, dw = torch.from_numpy(dim).cuda(), torch.from_numpy(w).cuda()
    hj = HashJoin.getInstance(int(dk.data_ptr()), m, capi.INT64, ExpressionRange(True, 0, m - 1))
    n = 6_000_000
    k = rng.integers(-1000, m + 1000, n).astype(np.int64)
    k[rng.random(n) < 0.2] = 12345                      # a hot key (heavy-hitter path -> spill kernel)
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, -1000, m + 999)),
             InputColDescriptor(capi.INT64, False, ExpressionRange(True, -10**6, 10**6))]
    inner = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 0, m - 1)),
             InputColDescriptor(capi.INT64, True, ExpressionRange(True, -1000, 1000, True))]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.SUM, 1), TargetExpr(capi.SUM, 1, 1), TargetExpr(capi.COUNT),
                                     TargetExpr(capi.COUNT, 1, 1)],
                             inner_col_descs=inner, join_outer_col=0, join_table=hj,
                             join_kind=capi.JOIN_LEFT if left else capi.JOIN_INNER)
    dev = [torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda()]
    fr = FetchResult([[int(t.data_ptr()) for t in dev]], [n], [int(dk.data_ptr()), int(dw.data_ptr())], m,
                     keepalive=dev + [dk, dw])
    rs = Executor(0).executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=3)
    assert rs.report.variant == 3
    oj = oracle.OracleJoin(dim, capi.INT64, 0, m - 1)
    ra.join_table = None
    q, want, code = oracle.execute(ra.to_plan(), [[k, v]], [dim, w], oj, n_threads=4)
    assert code == 0
    compare_buffers(q, want, rs.getStorage())


def test_keyed_probe_in_several_passes(torch_cuda, oracle, monkeypatch):
    """The keyed probe with its slot range walked in 3 passes per partition (what a table too large for one
    L2-resident slice per partition gets), forced on a small table."""
    test_payload_probe_matches_oracle(torch_cuda, oracle, True, True, "all_targets", True, keyed_passes=3)
    test_payload_probe_matches_oracle(torch_cuda, oracle, False, False, "hot_key", True, keyed_passes=3)


@pytest.mark.parametrize("sparse", [False, True], ids=["perfect", "keyed"])
@pytest.mark.parametrize("left", [False, True], ids=["inner", "left"])
def test_grouped_join_through_the_gather_route_at_16m_rows(torch_cuda, oracle, left, sparse):
    """SELECT f.g, COUNT(*), SUM(d.w), MAX(d.x), AVG(d.x), SUM(f.v) FROM f [LEFT] JOIN d ON f.k = d.k GROUP BY f.g over 16 M outer rows:
    the planner's own choice (api.cpp execute_join_gather: k_join_gather + the step without a join) against the oracle's join loop
    and against the row kernel (kernel_variant 1).  Keys that miss, NULL join keys, a nullable inner column."""
    from heavydb_amd.executor import (Executor, ExpressionRange, FetchResult, HashJoin, InputColDescriptor,
                                      RelAlgExecutionUnit, TargetExpr)
    torch = torch_cuda
    rng = np.random.default_rng(5 + 2 * int(left) + int(sparse))
    m, n = 1_000_000, 16_000_000
    mul = 1000003 if sparse else 1
    dim_k = rng.permutation(m).astype(np.int64) * mul
    dim_w = rng.integers(-1000, 1000, m).astype(np.int64)
    dim_x = rng.integers(-5000, 5000, m).astype(np.int32)
    dim_x[rng.random(m) < 0.1] = np.iinfo(np.int32).min
    fk = rng.integers(-m // 10, m + m // 10, n).astype(np.int64) * mul
    fk[rng.random(n) < 0.01] = np.iinfo(np.int64).min
    fg = rng.integers(0, 20_000, n).astype(np.int32)      # (a table beyond the row kernel's 64 KB LDS copy: the route's own territory)
    fv = rng.integers(-10**6, 10**6, n).astype(np.int64)
    kmin, kmax = int(dim_k.min()), int(dim_k.max())
    fdescs = [InputColDescriptor(capi.INT64, True, ExpressionRange(True, -(m // 10) * mul, (m + m // 10) * mul, True)),
              InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 19_999)),
              InputColDescriptor(capi.INT64, False, ExpressionRange(True, -10**6, 10**6))]
    idescs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, kmin, kmax)),
              InputColDescriptor(capi.INT64, False, ExpressionRange(True, -1000, 999)),
              InputColDescriptor(capi.INT32, True, ExpressionRange(True, -5000, 4999, True))]
    dk, dw, dx = (torch.from_numpy(a).cuda() for a in (dim_k, dim_w, dim_x))
    hj = HashJoin.getInstance(int(dk.data_ptr()), m, capi.INT64, ExpressionRange(True, kmin, kmax))
    assert hj.info()["hash_type"] == (1 if sparse else 0)
    ra = RelAlgExecutionUnit(fdescs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1, 1),
                                      TargetExpr(capi.MAX, 2, 1), TargetExpr(capi.AVG, 2, 1), TargetExpr(capi.SUM, 2)], [], [1],
                             inner_col_descs=idescs, join_outer_col=0, join_table=hj,
                             join_kind=capi.JOIN_LEFT if left else capi.JOIN_INNER)
    cut = n // 2 + 12
    dev = [torch.from_numpy(a).cuda() for a in (fk, fg, fv)]
    fr = FetchResult([[int(t.data_ptr()) + o * t.element_size() for t in dev] for o in (0, cut)], [cut, n - cut],
                     [int(dk.data_ptr()), int(dw.data_ptr()), int(dx.data_ptr())], m, keepalive=dev + [dk, dw, dx])
    ex = Executor(0)
    assert ex.explain(ra, [cut, n - cut]).startswith("k_join_gather")
    rs = ex.executeWorkUnit(ra, fr, allow_retry=False)
    row = ex.executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=1)
    assert row.report.kernel_name.decode() == "k_generic"
    print(f"grouped join 16 M rows ({'keyed' if sparse else 'perfect'}, {'LEFT' if left else 'INNER'}): gather route "
          f"{rs.report.total_ms:.2f} ms ({rs.report.kernel_name.decode()}), row kernel {row.report.total_ms:.2f} ms")
    oj = oracle.OracleJoin(dim_k, capi.INT64, kmin, kmax)
    ra.join_table = None
    q, want, code = oracle.execute(ra.to_plan(), [[fk[:cut], fg[:cut], fv[:cut]], [fk[cut:], fg[cut:], fv[cut:]]],
                                   [dim_k, dim_w, dim_x], oj, n_threads=16)
    ra.join_table = hj
    assert code == 0
    compare_buffers(q, want, rs.getStorage())
    compare_buffers(q, want, row.getStorage())

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

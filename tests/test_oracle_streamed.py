"""orc_execute_streamed (the oracle's BASELINE-size entry point: fragments generated inside the kernel
threads, multi-threaded ResultSetStorage::reduce for big baseline tables) against the oracle's
array-fed execute on the same generated columns."""
from __future__ import annotations

import numpy as np
import pytest

from heavydb_amd import capi
from heavydb_amd.executor import (ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr)
from tests.helpers import compare_buffers

SEED = 0xC0FFEE00


def _cols(orc, gens, n, off=0):
    return [orc.generate_column(n, g[0], g[1], g[2], g[3], g[4], g[5], g[6] if len(g) > 6 else 0, off) for g in gens]


@pytest.mark.parametrize("threads,reduce_threads", [(1, 1), (3, 1), (4, 4)])
def test_streamed_baseline_groupby_matches_array_fed(oracle, threads, reduce_threads):
    n, n_keys, frag = 700_003, 60_000, 100_000     # 120 000 entries: the multi-threaded reduce applies
    gens = [(capi.GEN_I64_MOD_MUL, SEED, n_keys, 1000003, 7, 0.0),
            (capi.GEN_F64_UNIT, SEED + 1, 0, 0, 0, 1000.0, 13),   # nullable value column
            (capi.GEN_I32_UNIFORM31, SEED + 2, 0, 0, 0, 0.0)]
    descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, (n_keys - 1) * 1000003 + 7)),
             InputColDescriptor(capi.DOUBLE, True, ExpressionRange(True, 0, 0, True, 0.0, 1000.0)),
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 2**31 - 1))]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1),
                                     TargetExpr(capi.MIN, 1)],
                             [Qual(2, capi.LT, 2**30)], [0], max_groups_buffer_entry_guess=2 * n_keys)
    plan = ra.to_plan()
    frags = [_cols(oracle, gens, min(frag, n - o), o) for o in range(0, n, frag)]
    q0, want, code0 = oracle.execute(plan, frags, n_threads=1)
    q1, got, code1, timing = oracle.execute_streamed(plan, gens, n, frag_rows=frag, block_rows=7_001,
                                                     n_threads=threads, reduce_threads=reduce_threads)
    assert code0 == 0 and code1 == 0
    assert q1.entry_count == 2 * n_keys
    compare_buffers(q0, want, got, 1e-12)
    assert set(timing) == {"init_s", "kernels_s", "generate_s", "reduce_s"}


def test_streamed_join_and_row_offset(oracle):
    n, m = 250_000, 5_000
    gens = [(capi.GEN_I64_MOD, SEED, m + 100, 0, 0, 0.0), (capi.GEN_I64_MOD, SEED + 1, 2001, -1000, 0, 0.0)]
    krange = ExpressionRange(True, 0, m - 1)
    descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 0, m + 99)),
             InputColDescriptor(capi.INT64, False, ExpressionRange(True, -1000, 1000))]
    dim_k = np.arange(m, dtype=np.int64)
    dim_w = oracle.generate_column(m, capi.GEN_I64_MOD, SEED + 100, 2001, -1000)
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.SUM, 1), TargetExpr(capi.SUM, 1, 1), TargetExpr(capi.COUNT)],
                             inner_col_descs=[InputColDescriptor(capi.INT64, False, krange),
                                              InputColDescriptor(capi.INT64, False, ExpressionRange(True, -1000, 1000))],
                             join_outer_col=0)
    plan = ra.to_plan()
    join = oracle.OracleJoin(dim_k, capi.INT64, 0, m - 1)
    off = 12_345
    frags = [_cols(oracle, gens, 50_000, off + o) for o in range(0, n, 50_000)]
    q0, want, c0 = oracle.execute(plan, frags, [dim_k, dim_w], join, n_threads=2)
    q1, got, c1, _ = oracle.execute_streamed(plan, gens, n, frag_rows=50_000, block_rows=4096, inner_cols=[dim_k, dim_w],
                                             join=join, n_threads=3, row_offset=off)
    assert c0 == 0 and c1 == 0
    compare_buffers(q0, want, got)
    assert int(got.reshape(-1)[2]) == int(np.isin(np.concatenate([f[0] for f in frags]), dim_k).sum())

#!/usr/bin/env python
"""Boundary-value fuzz: columns drawn from the extremes of their types (the inline NULL sentinels
included, in NOT NULL columns too), every aggregate kind, grouped and not, row-wise and columnar —
the product's row logic (host emulation) against the oracle, bit for bit.  64-bit values stay small
enough that no SUM wraps.  usage: boundary_fuzz.py <seed> <iterations>"""
import sys

import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heavydb_amd import capi
from heavydb_amd.executor import ExpressionRange as V, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
from oracle import oracle
from tests.cases import Case, col_range
from tests.helpers import compare_buffers, qmd_equal
from tests.test_rowlogic_emu import _emu_execute
NP = {capi.INT8: np.int8, capi.INT16: np.int16, capi.INT32: np.int32, capi.INT64: np.int64}


def run(seed, iters):
    rng = np.random.default_rng(seed)
    tally = {}
    for it in range(iters):
        n = int(rng.integers(1, 200))
        cols, descs = [], []
        key = rng.integers(0, 6, n).astype(np.int32)
        cols.append(key); descs.append(InputColDescriptor(capi.INT32, False, V(True, 0, 5)))
        for c in range(int(rng.integers(1, 4))):
            t = [capi.INT8, capi.INT16, capi.INT32, capi.INT64][int(rng.integers(0, 4))]
            info = np.iinfo(NP[t])
            nullable = bool(rng.integers(0, 2))
            if t == capi.INT64:  # keep 64-bit sums away from wrap-around (order-dependent in the reference too)
                pool = np.array(([info.min] if nullable else []) + [-2**50, -2**31 - 1, -2, -1, 0, 1, 2, 2**31, 2**50], dtype=np.int64)
            else:
                pool = np.array([info.min, info.min + 1, info.min + 2, -2, -1, 0, 1, 2, info.max - 2, info.max - 1, info.max], dtype=np.int64)
            a = pool[rng.integers(0, len(pool), n)].astype(NP[t])
            valid_range = bool(rng.integers(0, 2))
            r = col_range([a], t, nullable) if valid_range else V(False)
            cols.append(a); descs.append(InputColDescriptor(t, nullable, r))
        targets = []
        for _ in range(int(rng.integers(1, 5))):
            k = int(rng.integers(0, 7))
            col = int(rng.integers(1, len(cols)))
            if k == 0: targets.append(TargetExpr(capi.COUNT))
            elif k == 1: targets.append(TargetExpr(capi.COUNT, col))
            elif k == 6: targets.append(TargetExpr(capi.SUM_IF, col, cond=Qual(int(rng.integers(1, len(cols))), capi.GT, 0)))
            else: targets.append(TargetExpr([capi.SUM, capi.AVG, capi.MIN, capi.MAX][k - 2], col))
        grouped = bool(rng.integers(0, 3))
        if grouped and rng.integers(0, 2): targets = [TargetExpr(capi.PROJECT_KEY)] + targets
        quals = [Qual(int(rng.integers(1, len(cols))), [capi.LT, capi.GE, capi.NE, capi.EQ][int(rng.integers(0, 4))], int(rng.choice([-1, 0, 1, 127, -128, 2**31 - 1])))] if rng.integers(0, 2) else []
        ra = RelAlgExecutionUnit(descs, targets, quals, [0] if grouped else [], output_columnar_hint=int(rng.integers(0, 2)))
        cut = n // 2
        case = Case("b", ra, [[c[:cut] for c in cols], [c[cut:] for c in cols]])
        plan = ra.to_plan()
        try:
            q, want, code = oracle.execute(plan, case.frags, n_threads=2)
        except capi.Mi355qError:
            tally["rejected"] = tally.get("rejected", 0) + 1; continue
        eq, got, ecode = _emu_execute(case, plan, None)
        assert (code == 0) == (ecode == 0), (seed, it, code, ecode)
        if code: tally["err"] = tally.get("err", 0) + 1; continue
        qmd_equal(q, eq)
        if q.output_columnar:
            from tests.helpers import columnar_to_rows, rowwise_qmd
            compare_buffers(rowwise_qmd(q), columnar_to_rows(q, want), columnar_to_rows(q, got), 1e-9)
        else:
            compare_buffers(q, want, got, 1e-9)
        tally["ok"] = tally.get("ok", 0) + 1
    return tally


if __name__ == "__main__":
    print(sys.argv[1], run(int(sys.argv[1]), int(sys.argv[2])))

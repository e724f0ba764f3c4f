"""Random expression programs through the product's KERNEL evaluator (expr.h eval_expr_rows: typed handlers, stack in LDS)
against the oracle's eval_expression: every micro-op over every operand type / nullability the handlers are instantiated
for — and the combinations that have none (FLOAT, INT8 / INT16 operands) — in both places the evaluator runs:

    SELECT <expr> FROM t                          the Projection family's general member (expressions in its registers)
    SELECT MIN(<expr>), COUNT(<expr>) FROM t      the interpreter pass k_project (+ the per-tile column batch) ahead of a scan

on the host simulation of the library (the real kernels_generic.hip / kernels_proj.hip compiled for the CPU).  Values AND
errors (7 overflow, 1 division by zero) must agree.  MI355Q_FUZZ_SEED / MI355Q_FUZZ_ITERS give other programs
(tools/soak_fuzz.py style)."""
import os

import numpy as np
import pytest

from heavydb_amd import capi
from heavydb_amd.executor import Executor, Expr, ExpressionRange, FetchResult, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
from tests.helpers import compare_buffers, compare_rows, hostsim_lib, qmd_equal
from tests.test_projection import aligned

I8, I16, I32, I64, F64, F32 = capi.INT8, capi.INT16, capi.INT32, capi.INT64, capi.DOUBLE, capi.FLOAT
NP = {I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64, F64: np.float64, F32: np.float32}
NUL = {I8: -2**7, I16: -2**15, I32: -2**31, I64: -2**63}
# the table: (type, nullable)
COLS = [(I32, False), (I32, True), (I64, True), (I64, False), (F64, True), (F64, False), (I16, True), (F32, True), (I8, True)]
N = 2500


@pytest.fixture(scope="module")
def sim():
    lib = capi.load_library(hostsim_lib())
    saved = capi._lib
    capi._lib = lib
    yield lib
    capi._lib = saved


def _table(rng):
    cols = []
    for t, nullable in COLS:
        if t in (F64, F32):
            a = rng.choice([0.0, 1.0, -1.0, 2.5, -3.75, 100.0, -1e-3], N).astype(NP[t]) * rng.choice([1, 1, 3], N).astype(NP[t])
            if nullable:
                a[rng.random(N) < 0.15] = np.finfo(NP[t]).tiny
        else:
            info = np.iinfo(NP[t])
            small = rng.integers(-6, 7, N)
            edge = rng.choice([info.max, info.min + 1, info.max - 1, 0, 1, -1], N)
            a = np.where(rng.random(N) < 0.2, edge, small).astype(NP[t])
            if nullable:
                a[rng.random(N) < 0.15] = NUL[t]
        cols.append(a)
    return cols


class Gen:
    """random well-typed expressions (the typing rules of plan.cpp lower_exprs: both operands of a binary op have its type)"""

    def __init__(self, rng):
        self.rng = rng

    def leaf(self, t):
        r = self.rng
        idx = [i for i, (ct, _) in enumerate(COLS) if ct == t]
        if idx and r.random() < 0.7:
            return Expr.col(int(r.choice(idx)))
        if r.random() < 0.1:
            return Expr.null(t)
        if t in (F64, F32):
            return Expr.lit(t, float(r.choice([0.0, 1.0, -2.0, 0.5, 3.0])))
        return Expr.lit(t, int(r.choice([0, 1, -1, 2, 3, 7, -5])))

    def value(self, t, depth):
        r = self.rng
        if depth <= 0 or r.random() < 0.25:
            return self.leaf(t)
        k = r.random()
        if k < 0.45:
            ops = [capi.EX_ADD, capi.EX_SUB, capi.EX_MUL, capi.EX_DIV] + ([capi.EX_MOD] if t not in (F64, F32) else [])
            return self.value(t, depth - 1)._bin(int(r.choice(ops)), self.value(t, depth - 1), t)
        if k < 0.6:
            src = int(r.choice([I32, I64, F64, I16, F32, I8]))
            if src == t:
                return self.value(t, depth - 1).neg(t)
            if src in (F64, F32) and t not in (F64, F32):
                # a floating-point value cast to an integer type it does not fit is undefined in the reference (fptosi) and
                # differs between the host's and the device's conversion instruction: only table values (|v| <= 300) and
                # literals are cast, and not to INT8
                return self.leaf(src).cast(t) if t != I8 else self.leaf(t)
            return self.value(src, depth - 1).cast(t)
        if k < 0.7:
            return self.value(t, depth - 1).neg(t)
        if k < 0.9:
            return Expr.case(self.boolean(depth - 1), self.value(t, depth - 1), self.value(t, depth - 1), t)
        return self.leaf(t)

    def boolean(self, depth):
        r = self.rng
        k = r.random()
        if depth <= 0 or k < 0.45:
            t = int(r.choice([I32, I64, F64, I32, I64, I16, F32]))
            op = int(r.choice([capi.EX_EQ, capi.EX_NE, capi.EX_LT, capi.EX_LE, capi.EX_GT, capi.EX_GE]))
            return self.value(t, max(depth - 1, 0)).cmp(op, self.value(t, max(depth - 1, 0)))
        if k < 0.7:
            return self.boolean(depth - 1).logical(int(r.choice([capi.EX_AND, capi.EX_OR])), self.boolean(depth - 1), bool(r.random() < 0.3))
        if k < 0.8:
            return self.boolean(depth - 1).logical_not()
        if k < 0.95:
            return self.value(int(r.choice([I32, I64, F64, I16, F32])), depth - 1).is_null()
        return Expr.case(self.boolean(depth - 1), self.boolean(depth - 1), self.boolean(depth - 1), I8)


def _stack_ok(e):
    sp = deepest = 0
    for n in e.nodes:
        if n.op in (capi.EX_COL, capi.EX_LIT):
            sp += 1
        elif n.op == capi.EX_CASE:
            sp -= 2
        elif n.op not in (capi.EX_CAST, capi.EX_NOT, capi.EX_IS_NULL, capi.EX_UMINUS):
            sp -= 1
        deepest = max(deepest, sp)
    return deepest <= 8 and len(e.nodes) <= capi.MAX_EXPR_NODES


def _run(oracle, ra, frags, keep):
    plan = ra.to_plan()
    try:
        q, want, code = oracle.execute(plan, frags, n_threads=1)
    except capi.Mi355qError:
        return "rejected"
    fr = FetchResult([[a.ctypes.data for a in cols] for cols in keep], [len(cols[0]) for cols in keep], [], 0, 0, [keep])
    ex = Executor(0)
    if code:
        with pytest.raises(capi.Mi355qError) as ei:
            ex.executeWorkUnit(ra, fr, allow_retry=False)
        if ei.value.code != code:
            # rows with different errors: the oracle reports the first in row order, a kernel whichever row's atomic lands
            # first (as the reference's GPU kernels do) — the product's code must be one that some row raises
            codes = set()
            for f in frags:
                for o in range(0, len(f[0]), 50):
                    if oracle.execute(plan, [[c[o:o + 50] for c in f]], n_threads=1)[2]:   # (a piece reports its first error only)
                        for r in range(o, min(o + 50, len(f[0]))):
                            codes.add(oracle.execute(plan, [[c[r:r + 1] for c in f]], n_threads=1)[2])
            assert ei.value.code in codes - {0}, (ei.value.code, codes)
        return "error %d" % code
    rs = ex.executeWorkUnit(ra, fr, allow_retry=False)
    qmd_equal(q, rs.getQueryMemDesc())
    compare_buffers(q, want, rs.getStorage(), 0.0)
    compare_rows(q, oracle.fetch_rows(q, want), rs.fetch(), 0.0)
    return "ok"


SEED = int(os.environ.get("MI355Q_FUZZ_SEED", "2025"))
ITERS = int(os.environ.get("MI355Q_FUZZ_ITERS", "160"))


@pytest.mark.parametrize("chunk", range(4))
def test_random_expressions_through_the_kernel_evaluator(sim, oracle, chunk):
    rng = np.random.default_rng(SEED * 7919 + chunk)
    cols = _table(rng)
    frags = [[c[:N // 2 + 3] for c in cols], [c[N // 2 + 3:] for c in cols]]
    keep = [[aligned(a) for a in f] for f in frags]
    descs = [InputColDescriptor(t, nullable, ExpressionRange()) for t, nullable in COLS]
    gen = Gen(rng)
    seen = {}
    for it in range(ITERS // 4):
        depth = int(rng.integers(1, 4))
        if rng.random() < 0.4:
            e, t = gen.boolean(depth), I8
        else:
            t = int(rng.choice([I32, I64, F64, I32, I64, F32, I16]))
            e = gen.value(t, depth)
        if not _stack_ok(e):
            continue
        e = e.with_range(ExpressionRange())
        nc = len(descs)
        # (1) the Projection family: one entry per row, the expression's value in it (plus a plain column beside it)
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT, nc), TargetExpr(capi.PROJECT, 0)], exprs=[e], max_groups_buffer_entry_guess=N)
        try:
            r1 = _run(oracle, ra, frags, keep)
        except AssertionError:
            print("FAILING EXPRESSION:", [(n.op, n.type, n.arg, n.ilit, n.flit, n.null_lit) for n in e.nodes])
            raise
        # (2) the interpreter pass ahead of a non-grouped scan
        ra2 = RelAlgExecutionUnit(descs, [TargetExpr(capi.MIN, nc), TargetExpr(capi.MAX, nc), TargetExpr(capi.COUNT, nc)], exprs=[e])
        r2 = _run(oracle, ra2, frags, keep)
        # (3) a BOOLEAN as a filter: evaluated for every row
        if t == I8:
            from heavydb_amd.executor import Qual
            ra3 = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 0)], [Qual(nc, capi.EQ, 1)], exprs=[e])
            r3 = _run(oracle, ra3, frags, keep)
            seen[r3] = seen.get(r3, 0) + 1
            # (4) ... and as the filter of a Projection (the quals' expressions of a quad are evaluated together in pass A)
            ra4 = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT, 0), TargetExpr(capi.PROJECT, 4)], [Qual(nc, capi.EQ, 1)], exprs=[e],
                                      max_groups_buffer_entry_guess=N)
            r4 = _run(oracle, ra4, frags, keep)
            seen[r4] = seen.get(r4, 0) + 1
        for r in (r1, r2):
            seen[r] = seen.get(r, 0) + 1
    assert seen.get("ok", 0) > ITERS // 8, seen
    print(seen)

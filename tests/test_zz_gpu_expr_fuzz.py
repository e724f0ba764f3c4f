"""tests/test_expr_fuzz.py on the device: the same random expression programs through the compiled handlers of
expr.h eval_expr_rows (the Projection family's general member, the interpreter pass k_project, BOOLEANs as filters)
against the oracle.  MI355Q_FUZZ_SEED / MI355Q_FUZZ_ITERS as there."""
from __future__ import annotations

import numpy as np
import pytest

from heavydb_amd import capi
from tests import test_expr_fuzz as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.load_library()
    return torch


class _Dev:
    """a stand-in for the host arrays of F._run: .ctypes.data = the device address"""

    class _C:
        def __init__(self, p):
            self.data = p

    def __init__(self, t):
        self.t = t
        self.ctypes = _Dev._C(int(t.data_ptr()))

    def __len__(self):
        return int(self.t.numel())


@pytest.mark.parametrize("chunk", range(4))
def test_random_expressions_on_the_device(torch_cuda, oracle, chunk):
    torch = torch_cuda
    rng = np.random.default_rng(F.SEED * 7919 + chunk)
    cols = F._table(rng)
    frags = [[c[:F.N // 2 + 3] for c in cols], [c[F.N // 2 + 3:] for c in cols]]
    keep = [[_Dev(torch.from_numpy(np.ascontiguousarray(a)).cuda()) for a in f] for f in frags]
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    descs = [InputColDescriptor(t, nullable, ExpressionRange()) for t, nullable in F.COLS]
    gen = F.Gen(rng)
    seen = {}
    for it in range(F.ITERS // 4):
        depth = int(rng.integers(1, 4))
        if rng.random() < 0.4:
            e, t = gen.boolean(depth), F.I8
        else:
            t = int(rng.choice([F.I32, F.I64, F.F64, F.I32, F.I64, F.F32, F.I16]))
            e = gen.value(t, depth)
        if not F._stack_ok(e):
            continue
        e = e.with_range(ExpressionRange())
        nc = len(descs)
        runs = [RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT, nc), TargetExpr(capi.PROJECT, 0)], exprs=[e], max_groups_buffer_entry_guess=F.N),
                RelAlgExecutionUnit(descs, [TargetExpr(capi.MIN, nc), TargetExpr(capi.MAX, nc), TargetExpr(capi.COUNT, nc)], exprs=[e])]
        if t == F.I8:
            runs.append(RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 0)], [Qual(nc, capi.EQ, 1)], exprs=[e]))
            runs.append(RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT, 0), TargetExpr(capi.PROJECT, 4)], [Qual(nc, capi.EQ, 1)], exprs=[e],
                                            max_groups_buffer_entry_guess=F.N))
        for ra in runs:
            try:
                r = F._run(oracle, ra, frags, keep)
            except AssertionError:
                print("FAILING EXPRESSION:", [(n.op, n.type, n.arg, n.ilit, n.flit, n.null_lit) for n in e.nodes])
                raise
            seen[r] = seen.get(r, 0) + 1
    assert seen.get("ok", 0) > F.ITERS // 8, seen

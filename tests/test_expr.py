"""Projected expressions (mi355q_expr): the cast / + - * micro-ops against the REFERENCE'S OWN runtime functions.

tests/golden/ref_expr_vectors.json was produced by oracle/gen_golden_expr.py from
cast_<from>_to_<to>_nullable and {add,sub,mul}_<type>_nullable[_lhs|_rhs] of QueryEngine/RuntimeFunctions.cpp
(compiled unmodified into oracle/_ref).  Both evaluators must reproduce every vector bit for bit:
  * the oracle's      oracle/oracle.cpp eval_expression      (orc_eval_expr)
  * the product's     heavydb_amd/csrc/expr.h eval_expr      (through the host emulation here; the device leg is
                      tests/test_gpu_parity.py::test_hip_expressions_match_reference_functions)
The overflow rule is not a runtime function in the reference (llvm.s{add,sub,mul}.with.overflow,
ArithmeticIR.cpp:840-909; codegenCastBetweenIntTypesOverflowChecks, CastIR.cpp:497-553) and is held against
exact integer arithmetic.  Whole steps with expressions are part of the case matrix (tests/cases.py expr_*):
oracle vs product row logic, vs SQLite, and vs the HIP library on the device."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from heavydb_amd import capi
from heavydb_amd.executor import Expr, ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
from tests.helpers import emu_lib

NP = {capi.INT8: np.int8, capi.INT16: np.int16, capi.INT32: np.int32, capi.INT64: np.int64,
      capi.DOUBLE: np.float64, capi.FLOAT: np.float32}
INTS = [capi.INT8, capi.INT16, capi.INT32, capi.INT64]
INT_NULL = {capi.INT8: -2**7, capi.INT16: -2**15, capi.INT32: -2**31, capi.INT64: -2**63}
INT_MAX = {t: -v - 1 for t, v in INT_NULL.items()}


def _vectors():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_expr_vectors.json")
    with open(path) as f:
        return json.load(f)


def _col_of(t, pattern):
    """a one-row column of type t holding the 64-bit pattern (integers sign-extended, fp as bits)"""
    if t == capi.DOUBLE:
        return np.array([struct.unpack("<d", struct.pack("<q", pattern))[0]], dtype=np.float64)
    if t == capi.FLOAT:
        return np.array([struct.unpack("<f", struct.pack("<I", pattern & 0xffffffff))[0]], dtype=np.float32)
    return np.array([pattern], dtype=NP[t])


def _plan(descs, expr):
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT)], exprs=[expr.with_range(ExpressionRange())])
    return ra.to_plan()


def _eval_both(oracle, plan, cols):
    """(oracle pattern, product pattern, oracle code, product code) of expression 0 on row 0"""
    ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    ol = oracle.lib()
    ol.orc_eval_expr.restype = C.c_int32
    ol.orc_eval_expr.argtypes = [C.POINTER(capi.Plan), C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    ob, ot, on = C.c_int64(), C.c_int32(), C.c_int32()
    oc = ol.orc_eval_expr(C.byref(plan), 0, ptrs, 0, C.byref(ob), C.byref(ot), C.byref(on))
    el = emu_lib()
    el.emu_eval_expr.restype = C.c_int32
    el.emu_eval_expr.argtypes = [C.POINTER(capi.Plan), C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int32)]
    eb, et = C.c_int64(), C.c_int32()
    ec = el.emu_eval_expr(C.byref(plan), 0, ptrs, 0, C.byref(eb), C.byref(et))
    return ob.value, eb.value, oc, ec


def _same_pattern(t, a, b):
    if t == capi.FLOAT:
        return (a & 0xffffffff) == (b & 0xffffffff)
    return a == b


def test_casts_match_the_reference_runtime_functions(oracle):
    for v in _vectors()["cast"]:
        f, t = v["from"], v["to"]
        plan = _plan([InputColDescriptor(f, True)], Expr.col(0).cast(t))
        ob, eb, oc, ec = _eval_both(oracle, plan, [_col_of(f, v["in"])])
        assert oc == 0 and ec == 0, v
        assert _same_pattern(t, ob, v["out"]), ("oracle", v, ob)
        assert _same_pattern(t, eb, v["out"]), ("product", v, eb)


def test_arithmetic_matches_the_reference_runtime_functions(oracle):
    for v in _vectors()["arith"]:
        t, sfx = v["type"], v["suffix"]
        descs = [InputColDescriptor(t, sfx in ("_nullable", "_nullable_lhs")),
                 InputColDescriptor(t, sfx in ("_nullable", "_nullable_rhs"))]
        e = Expr.col(0)._bin(v["op"], Expr.col(1), t)
        ob, eb, oc, ec = _eval_both(oracle, _plan(descs, e), [_col_of(t, v["a"]), _col_of(t, v["b"])])
        assert oc == 0 and ec == 0, v
        assert _same_pattern(t, ob, v["out"]), ("oracle", v, ob)
        assert _same_pattern(t, eb, v["out"]), ("product", v, eb)


def test_vectors_still_match_the_reference_live(oracle):
    """When oracle/_ref is present (build container), a sample of the vectors is re-derived from the reference's
    functions themselves: the committed file has not drifted."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built here")
    ref = C.CDLL(oracle.REF_LIB)
    ref.cast_int32_t_to_double_nullable.restype = C.c_double
    ref.cast_int32_t_to_double_nullable.argtypes = [C.c_int32, C.c_int32, C.c_double]
    n = 0
    for v in _vectors()["cast"]:
        if v["from"] == capi.INT32 and v["to"] == capi.DOUBLE:
            r = ref.cast_int32_t_to_double_nullable(v["in"], -2**31, float(np.finfo(np.float64).tiny))
            assert struct.unpack("<q", struct.pack("<d", r))[0] == v["out"]
            n += 1
    assert n > 5
    ref.mul_int64_t_nullable_lhs.restype = C.c_int64
    ref.mul_int64_t_nullable_lhs.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    for v in _vectors()["arith"]:
        if v["type"] == capi.INT64 and v["op"] == capi.EX_MUL and v["suffix"] == "_nullable_lhs":
            assert ref.mul_int64_t_nullable_lhs(v["a"], v["b"], -2**63) == v["out"]


def test_integer_overflow_rule_against_exact_arithmetic(oracle):
    """s{add,sub,mul}.with.overflow at the operand type's width: error 7 exactly when the exact result leaves the
    type (a NULL operand skips the check); otherwise the exact result."""
    rng = np.random.default_rng(77)
    for t in INTS:
        lo, hi = INT_NULL[t], INT_MAX[t]
        edge = [lo, lo + 1, -1, 0, 1, 2, hi - 1, hi, hi // 2 + 1, lo // 2 - 1]
        vals = edge + [int(x) for x in rng.integers(lo, hi, 12, endpoint=True)]
        for op, f in ((capi.EX_ADD, lambda a, b: a + b), (capi.EX_SUB, lambda a, b: a - b), (capi.EX_MUL, lambda a, b: a * b)):
            for nullable in (False, True):
                descs = [InputColDescriptor(t, nullable), InputColDescriptor(t, nullable)]
                plan = _plan(descs, Expr.col(0)._bin(op, Expr.col(1), t))
                for a in vals:
                    for b in vals[::2]:
                        ob, eb, oc, ec = _eval_both(oracle, plan, [_col_of(t, a), _col_of(t, b)])
                        if nullable and (a == lo or b == lo):
                            assert oc == 0 and ec == 0 and ob == lo and eb == lo, (t, op, a, b)
                            continue
                        exact = f(a, b)
                        if lo <= exact <= hi:
                            assert oc == 0 and ec == 0 and ob == exact and eb == exact, (t, op, a, b, ob, eb)
                        else:
                            assert oc == 7 and ec == 7, (t, op, a, b, oc, ec)


def test_division_rules(oracle):
    """codegenDiv / codegenMod (ArithmeticIR.cpp:431-560, :731-760; the checks are generated IR, not runtime functions, so they
    are held to the rule itself): DIV by zero is error 1 unless an operand may be NULL and one of them IS the NULL pattern
    (then the check is skipped and the nullable function answers); MOD tests the divisor FIRST, whatever the NULLs; floating
    point: a divisor that is not "ordered and != 0" (0.0, -0.0, NaN) is error 1.  Quotients truncate towards zero, remainders
    take the dividend's sign."""
    for t in INTS:
        lo, hi = INT_NULL[t], INT_MAX[t]
        vals = [lo, lo + 1, -7, -1, 0, 1, 3, hi]
        for op in (capi.EX_DIV, capi.EX_MOD):
            for ln, rn in ((False, False), (True, False), (False, True), (True, True)):
                plan = _plan([InputColDescriptor(t, ln), InputColDescriptor(t, rn)], Expr.col(0)._bin(op, Expr.col(1), t))
                for a in vals:
                    for b in vals:
                        if not (ln or rn) and (a == lo or b == lo):
                            continue   # NOT NULL columns never hold the pattern in the reference's own tests; INT_MIN / -1 traps there
                        if a == lo and b == -1 and not ln:
                            continue   # INT_MIN / -1 as values: SIGFPE in the reference
                        ob, eb, oc, ec = _eval_both(oracle, plan, [_col_of(t, a), _col_of(t, b)])
                        skip = op == capi.EX_DIV and (ln or rn) and (a == lo or b == lo)
                        if b == 0 and not skip:
                            assert oc == 1 and ec == 1, (t, op, ln, rn, a, b, oc, ec)
                            continue
                        if b == 0:   # behind the skip with a real INT_MIN dividend (only rhs nullable): undefined there
                            if not (ln and a == lo):
                                continue
                        assert oc == 0 and ec == 0, (t, op, ln, rn, a, b, oc, ec)
                        if (ln and a == lo) or (rn and b == lo):
                            assert ob == lo and eb == lo, (t, op, ln, rn, a, b, ob, eb)
                        else:
                            q = abs(a) // abs(b) * (1 if (a < 0) == (b < 0) else -1)
                            want = q if op == capi.EX_DIV else a - q * b
                            assert ob == want and eb == want, (t, op, ln, rn, a, b, ob, eb, want)
    for t in (capi.DOUBLE, capi.FLOAT):
        plan = _plan([InputColDescriptor(t, True), InputColDescriptor(t, False)], Expr.col(0).div(Expr.col(1), t))
        dt = np.float64 if t == capi.DOUBLE else np.float32
        for a, b, code in ((6.0, 1.5, 0), (1.0, 0.0, 1), (1.0, -0.0, 1), (1.0, float("nan"), 1), (0.0, 3.0, 0)):
            ob, eb, oc, ec = _eval_both(oracle, plan, [np.array([a], dtype=dt), np.array([b], dtype=dt)])
            assert oc == code and ec == code, (t, a, b, oc, ec)
            if code == 0:
                assert _same_pattern(t, ob, eb)
        # a NULL dividend skips the zero check and stays NULL
        ob, eb, oc, ec = _eval_both(oracle, plan, [np.array([np.finfo(dt).tiny], dtype=dt), np.array([0.0], dtype=dt)])
        assert oc == 0 and ec == 0 and _same_pattern(t, ob, eb)


def test_narrowing_cast_rule(oracle):
    """codegenCastBetweenIntTypesOverflowChecks (CastIR.cpp:497-553): error when v > max(to) or v <= min(to) —
    the target's minimum (its NULL sentinel) is refused too; a NULL operand passes and becomes the target's NULL."""
    for f in INTS:
        for t in INTS:
            if INT_MAX[t] >= INT_MAX[f]:
                continue
            for nullable in (False, True):
                plan = _plan([InputColDescriptor(f, nullable)], Expr.col(0).cast(t))
                for v in (INT_MAX[t], INT_MAX[t] + 1, INT_NULL[t], INT_NULL[t] + 1, INT_NULL[t] - 1, 0, -5, INT_NULL[f]):
                    ob, eb, oc, ec = _eval_both(oracle, plan, [_col_of(f, v)])
                    if nullable and v == INT_NULL[f]:
                        assert oc == 0 and ec == 0 and ob == INT_NULL[t] and eb == INT_NULL[t]
                    elif v > INT_MAX[t] or v <= INT_NULL[t]:
                        assert oc == 7 and ec == 7, (f, t, v, oc, ec)
                    else:
                        assert oc == 0 and ec == 0 and ob == v and eb == v


def test_invalid_programs_are_refused(oracle):
    """plan-time validation (plan.cpp lower_exprs and the oracle's lower_plan): operand types that differ from
    the node's, an unbalanced stack, a column out of range, an expression used as a join key."""
    descs = [InputColDescriptor(capi.INT32, False), InputColDescriptor(capi.INT64, False)]
    bad = [Expr.col(0).add(Expr.col(1), capi.INT64),      # int32 + int64 without the analyzer's cast
           Expr.col(0).add(Expr.lit(capi.INT32, 1), capi.INT64),
           Expr(Expr.col(0).nodes + Expr.col(1).nodes),    # two values left on the stack
           Expr(Expr.col(0).cast(capi.INT64).nodes[1:]),   # cast of nothing
           Expr.col(5)]
    el = emu_lib()
    for e in bad:
        plan = _plan(descs, e)
        q = capi.QMD()
        assert el.emu_qmd_init(C.byref(plan), C.byref(q)) == capi.ERR_INVALID_PLAN
        with pytest.raises(Exception):
            oracle.qmd_init(plan)
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT)], exprs=[Expr.col(1).add(Expr.lit(capi.INT64, 1), capi.INT64)],
                             inner_col_descs=[InputColDescriptor(capi.INT64, False)], join_outer_col=2)
    q = capi.QMD()
    assert el.emu_qmd_init(C.byref(ra.to_plan()), C.byref(q)) == capi.ERR_INVALID_PLAN


# ---- comparisons of two values and CASE (round 4) ---------------------------------------------------------------------------
def test_comparisons_match_the_reference_runtime_functions(oracle):
    """{eq,ne,lt,le,gt,ge}_<type>_nullable[_lhs|_rhs] of RuntimeFunctions.cpp:73-149, 7 128 vectors incl. NaN / inf operands
    and NULL patterns under every nullability of the operands: 1 / 0 / the BOOLEAN NULL, bit for bit"""
    vs = _vectors()["cmp"]
    assert len(vs) > 7000
    for v in vs:
        t, sfx = v["type"], v["suffix"]
        descs = [InputColDescriptor(t, sfx in ("_nullable", "_nullable_lhs")),
                 InputColDescriptor(t, sfx in ("_nullable", "_nullable_rhs"))]
        e = Expr.col(0).cmp(v["op"], Expr.col(1))
        ob, eb, oc, ec = _eval_both(oracle, _plan(descs, e), [_col_of(t, v["a"]), _col_of(t, v["b"])])
        assert oc == 0 and ec == 0, v
        assert ob == v["out"], ("oracle", v, ob)
        assert eb == v["out"], ("product", v, eb)


def test_comparison_vectors_still_match_the_reference_live(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built here")
    ref = C.CDLL(oracle.REF_LIB)
    ref.le_double_nullable_rhs.restype = C.c_int8
    ref.le_double_nullable_rhs.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int8]
    n = 0
    for v in _vectors()["cmp"]:
        if v["type"] == capi.DOUBLE and v["op"] == capi.EX_LE and v["suffix"] == "_nullable_rhs":
            a, b = (struct.unpack("<d", struct.pack("<q", v[k]))[0] for k in ("a", "b"))
            assert ref.le_double_nullable_rhs(a, b, float(np.finfo(np.float64).tiny), -128) == v["out"]
            n += 1
    assert n > 20


def _case_eval(oracle, descs, e, cols):
    ob, eb, oc, ec = _eval_both(oracle, _plan(descs, e), cols)
    assert oc == ec, (oc, ec)
    if oc == 0:
        assert ob == eb, (ob, eb)
    return ob, oc


def test_case_rules(oracle):
    """codegenCase (CaseIR.cpp:67-140): THEN where the condition is TRUE, a NULL condition is not; the branches are lazy —
    a check in the branch that is not taken does not fire; an error in the condition or in the TAKEN branch does."""
    I64, NUL = capi.INT64, -2**63
    d2 = [InputColDescriptor(I64, True), InputColDescriptor(I64, True)]
    a, b = Expr.col(0), Expr.col(1)
    # CASE WHEN b <> 0 THEN a / b ELSE -1 END: the guard of every safe division
    guard = Expr.case(b.cmp(capi.EX_NE, Expr.lit(I64, 0)), a.div(b, I64), Expr.lit(I64, -1), I64)
    col = lambda v: np.array([v], dtype=np.int64)
    assert _case_eval(oracle, d2, guard, [col(84), col(2)]) == (42, 0)
    assert _case_eval(oracle, d2, guard, [col(84), col(0)]) == (-1, 0)          # no DIV_BY_ZERO: the THEN branch does not run
    assert _case_eval(oracle, d2, guard, [col(84), col(NUL)]) == (-1, 0)        # NULL <> 0 is NULL: not TRUE
    assert _case_eval(oracle, d2, guard, [col(NUL), col(2)]) == (NUL, 0)        # NULL / 2
    # the same division unguarded raises error 1
    assert _case_eval(oracle, d2, a.div(b, I64), [col(84), col(0)])[1] == capi.ERR_DIV_BY_ZERO
    # an error in the TAKEN branch counts; in the condition too
    over = Expr.case(a.cmp(capi.EX_GT, b), a.mul(a, I64), b, I64)
    assert _case_eval(oracle, d2, over, [col(2**62), col(5)])[1] == 7
    assert _case_eval(oracle, d2, over, [col(3), col(2**62)]) == (2**62, 0)
    cond_err = Expr.case(a.div(b, I64).cmp(capi.EX_EQ, Expr.lit(I64, 1)), a, b, I64)
    assert _case_eval(oracle, d2, cond_err, [col(3), col(0)])[1] == capi.ERR_DIV_BY_ZERO
    # no ELSE: the NULL constant; a chain of WHENs nests in the ELSE position
    no_else = Expr.case(a.cmp(capi.EX_LT, b), a, Expr.null(I64), I64)
    assert _case_eval(oracle, d2, no_else, [col(1), col(2)]) == (1, 0)
    assert _case_eval(oracle, d2, no_else, [col(2), col(1)]) == (NUL, 0)
    d1 = [InputColDescriptor(capi.INT32, True)]
    x = Expr.col(0)
    chain = Expr.case(x.cmp(capi.EX_LT, Expr.lit(capi.INT32, 10)), Expr.lit(capi.INT32, 1),
                      Expr.case(x.cmp(capi.EX_LT, Expr.lit(capi.INT32, 100)), Expr.lit(capi.INT32, 2), Expr.lit(capi.INT32, 3), capi.INT32),
                      capi.INT32)
    c32 = lambda v: np.array([v], dtype=np.int32)
    assert [_case_eval(oracle, d1, chain, [c32(v)])[0] for v in (5, 50, 500, -2**31)] == [1, 2, 3, 3]
    # floating point: DOUBLE branches, FLOAT comparison
    dd = [InputColDescriptor(capi.DOUBLE, True), InputColDescriptor(capi.DOUBLE, False)]
    fmax = Expr.case(Expr.col(0).cmp(capi.EX_GE, Expr.col(1)), Expr.col(0), Expr.col(1), capi.DOUBLE)
    pat = lambda v: struct.unpack("<q", struct.pack("<d", v))[0]
    f64 = lambda v: np.array([v], dtype=np.float64)
    assert _case_eval(oracle, dd, fmax, [f64(2.5), f64(1.5)]) == (pat(2.5), 0)
    assert _case_eval(oracle, dd, fmax, [f64(float(np.finfo(np.float64).tiny)), f64(1.5)]) == (pat(1.5), 0)   # NULL >= x: not TRUE
    assert _case_eval(oracle, dd, fmax, [f64(float("nan")), f64(1.5)]) == (pat(1.5), 0)


def test_logic_and_unary_minus_match_the_reference_runtime_functions(oracle):
    """logical_not / logical_and / logical_or over nullable BOOLEANs and uminus_<type>_nullable (RuntimeFunctions.cpp:247-258,
    :331-358): both evaluators, bit for bit.  BOOLEAN operands are nullable INT8 columns holding 1 / 0 / -128."""
    vec = _vectors()
    b8 = [InputColDescriptor(capi.INT8, True), InputColDescriptor(capi.INT8, True)]
    for v in vec["logic"]:
        e = Expr.col(0).logical_not() if v["op"] == capi.EX_NOT else Expr.col(0).logical(v["op"], Expr.col(1))
        ob, eb, oc, ec = _eval_both(oracle, _plan(b8, e), [_col_of(capi.INT8, v["a"]), _col_of(capi.INT8, v["b"])])
        assert (oc, ec, ob, eb) == (0, 0, v["out"], v["out"]), (v, ob, eb)
    assert len(vec["logic"]) == 21
    n = 0
    for v in vec["uminus"]:
        t = v["type"]
        ob, eb, oc, ec = _eval_both(oracle, _plan([InputColDescriptor(t, True)], Expr.col(0).neg(t)), [_col_of(t, v["in"])])
        assert oc == 0 and ec == 0, v
        assert _same_pattern(t, ob, v["out"]) and _same_pattern(t, eb, v["out"]), (v, hex(ob), hex(eb))
        n += 1
    assert n > 80


def test_logic_rules(oracle):
    """What the generated IR adds around those functions (LogicalIR.cpp:197-432, ArithmeticIR.cpp:787-838; no runtime function
    to run): NOT NULL operands go through toBool; the short-circuit form lets the first operand decide (NULL included) and then
    never enters the second one's block; IS NULL of a NOT NULL operand is constant false without evaluating it; -x of the type's
    minimum as a VALUE is error 7."""
    i8 = lambda v: np.array([v], dtype=np.int8)
    i32 = lambda v: np.array([v], dtype=np.int32)
    NUL = -128
    # toBool(lhs) op toBool(rhs) / CreateNot(toBool) on NOT NULL operands
    nn = [InputColDescriptor(capi.INT8, False), InputColDescriptor(capi.INT8, False)]
    for a in (0, 1):
        assert _eval_both(oracle, _plan(nn, Expr.col(0).logical_not()), [i8(a), i8(0)]) == (1 - a, 1 - a, 0, 0)
        for b in (0, 1):
            for op, want in ((capi.EX_AND, a & b), (capi.EX_OR, a | b)):
                for sc in (False, True):
                    assert _eval_both(oracle, _plan(nn, Expr.col(0).logical(op, Expr.col(1), sc)), [i8(a), i8(b)]) == (want, want, 0, 0)
    # one nullable operand makes the BinOper nullable: three-valued in the plain form ...
    mix = [InputColDescriptor(capi.INT8, True), InputColDescriptor(capi.INT8, False)]
    assert _eval_both(oracle, _plan(mix, Expr.col(0).logical(capi.EX_AND, Expr.col(1))), [i8(NUL), i8(0)])[:2] == (0, 0)
    assert _eval_both(oracle, _plan(mix, Expr.col(0).logical(capi.EX_OR, Expr.col(1))), [i8(NUL), i8(1)])[:2] == (1, 1)
    # ... and first-operand-decides in the short-circuit form: NULL AND FALSE = NULL there (the phi's nullcheck_fail input)
    assert _eval_both(oracle, _plan(mix, Expr.col(0).logical(capi.EX_AND, Expr.col(1), True)), [i8(NUL), i8(0)])[:2] == (NUL, NUL)
    assert _eval_both(oracle, _plan(mix, Expr.col(0).logical(capi.EX_OR, Expr.col(1), True)), [i8(NUL), i8(1)])[:2] == (NUL, NUL)
    rmix = [InputColDescriptor(capi.INT8, False), InputColDescriptor(capi.INT8, True)]
    assert _eval_both(oracle, _plan(rmix, Expr.col(0).logical(capi.EX_AND, Expr.col(1), True)), [i8(1), i8(NUL)])[:2] == (NUL, NUL)
    assert _eval_both(oracle, _plan(rmix, Expr.col(0).logical(capi.EX_AND, Expr.col(1), True)), [i8(0), i8(NUL)])[:2] == (0, 0)
    assert _eval_both(oracle, _plan(rmix, Expr.col(0).logical(capi.EX_OR, Expr.col(1), True)), [i8(1), i8(NUL)])[:2] == (1, 1)
    # b <> 0 AND a / b > 1: the division's zero check exists only where the first operand lets the second run
    d = [InputColDescriptor(capi.INT32, False), InputColDescriptor(capi.INT32, False)]
    safe = Expr.col(1).cmp(capi.EX_NE, Expr.lit(capi.INT32, 0))
    unsafe = Expr.col(0).div(Expr.col(1), capi.INT32).cmp(capi.EX_GT, Expr.lit(capi.INT32, 1))
    guarded = safe.logical(capi.EX_AND, unsafe, True)
    assert _eval_both(oracle, _plan(d, guarded), [i32(10), i32(0)]) == (0, 0, 0, 0)
    assert _eval_both(oracle, _plan(d, guarded), [i32(10), i32(3)]) == (1, 1, 0, 0)
    assert _eval_both(oracle, _plan(d, guarded), [i32(10), i32(20)]) == (0, 0, 0, 0)
    plain = safe.logical(capi.EX_AND, unsafe)                 # (the plain form evaluates both: error 1)
    assert _eval_both(oracle, _plan(d, plain), [i32(10), i32(0)])[2:] == (capi.ERR_DIV_BY_ZERO, capi.ERR_DIV_BY_ZERO)
    either = Expr.col(1).cmp(capi.EX_EQ, Expr.lit(capi.INT32, 0)).logical(capi.EX_OR, unsafe, True)
    assert _eval_both(oracle, _plan(d, either), [i32(10), i32(0)]) == (1, 1, 0, 0)
    assert _eval_both(oracle, _plan(d, either), [i32(10), i32(3)]) == (1, 1, 0, 0)
    # an error in the FIRST operand ends the step in both forms
    first_bad = unsafe.logical(capi.EX_OR, safe, True)
    assert _eval_both(oracle, _plan(d, first_bad), [i32(10), i32(0)])[2:] == (capi.ERR_DIV_BY_ZERO, capi.ERR_DIV_BY_ZERO)
    # NOT over a floating-point comparison is not the opposite comparison: a NaN makes `d < 1.5` FALSE and its negation TRUE
    # (fcmp olt + CreateNot), while `d >= 1.5` is FALSE too — which is why the binding keeps such a NOT as an expression
    dd = [InputColDescriptor(capi.DOUBLE, False)]
    nan = [np.array([float("nan")])]
    assert _eval_both(oracle, _plan(dd, Expr.col(0).cmp(capi.EX_LT, Expr.lit(capi.DOUBLE, 1.5)).logical_not()), nan) == (1, 1, 0, 0)
    assert _eval_both(oracle, _plan(dd, Expr.col(0).cmp(capi.EX_GE, Expr.lit(capi.DOUBLE, 1.5))), nan) == (0, 0, 0, 0)
    # IS NULL: the comparison with the inline NULL for a nullable operand (the NULL literal included) ...
    for t, null_v, some in ((capi.INT32, -2**31, 5), (capi.INT64, -2**63, -9), (capi.INT8, -128, 0)):
        dn = [InputColDescriptor(t, True)]
        assert _eval_both(oracle, _plan(dn, Expr.col(0).is_null()), [_col_of(t, null_v)]) == (1, 1, 0, 0)
        assert _eval_both(oracle, _plan(dn, Expr.col(0).is_null()), [_col_of(t, some)]) == (0, 0, 0, 0)
        assert _eval_both(oracle, _plan(dn, Expr.col(0).is_null().logical_not()), [_col_of(t, null_v)]) == (0, 0, 0, 0)   # IS NOT NULL
        # ... constant false for a NOT NULL one, whatever pattern it holds
        assert _eval_both(oracle, _plan([InputColDescriptor(t, False)], Expr.col(0).is_null()), [_col_of(t, null_v)]) == (0, 0, 0, 0)
    tiny = struct.unpack("<q", struct.pack("<d", float(np.finfo(np.float64).tiny)))[0]
    assert _eval_both(oracle, _plan([InputColDescriptor(capi.DOUBLE, True)], Expr.col(0).is_null()), [_col_of(capi.DOUBLE, tiny)]) == (1, 1, 0, 0)
    assert _eval_both(oracle, _plan([InputColDescriptor(capi.DOUBLE, True)], Expr.col(0).is_null()),
                      [np.array([float("nan")])]) == (0, 0, 0, 0)
    assert _eval_both(oracle, _plan([InputColDescriptor(capi.INT32, False)], Expr.null(capi.INT32).is_null()), [i32(0)]) == (1, 1, 0, 0)
    # (a NOT NULL operand is not evaluated: its overflow cannot fire; a nullable one is, and it does)
    big = [InputColDescriptor(capi.INT32, False), InputColDescriptor(capi.INT32, True)]
    assert _eval_both(oracle, _plan(big, Expr.col(0).add(Expr.col(0), capi.INT32).is_null()), [i32(2**31 - 1), i32(1)]) == (0, 0, 0, 0)
    assert _eval_both(oracle, _plan(big, Expr.col(0).add(Expr.col(1), capi.INT32).is_null()), [i32(2**31 - 1), i32(1)])[2:] == (7, 7)
    assert _eval_both(oracle, _plan(big, Expr.col(0).add(Expr.col(1), capi.INT32).is_null()), [i32(2**31 - 1), i32(-2**31)]) == (1, 1, 0, 0)
    # unary minus: the type's minimum is NULL for a nullable operand and an overflow for a NOT NULL one
    for t in INTS:
        lo = INT_NULL[t]
        assert _eval_both(oracle, _plan([InputColDescriptor(t, True)], Expr.col(0).neg(t)), [_col_of(t, lo)]) == (lo, lo, 0, 0)
        assert _eval_both(oracle, _plan([InputColDescriptor(t, False)], Expr.col(0).neg(t)), [_col_of(t, lo)])[2:] == (7, 7)
        assert _eval_both(oracle, _plan([InputColDescriptor(t, False)], Expr.col(0).neg(t)), [_col_of(t, lo + 1)]) == (-lo - 1, -lo - 1, 0, 0)
        assert _eval_both(oracle, _plan([InputColDescriptor(t, False)], Expr.col(0).neg(t)), [_col_of(t, INT_MAX[t])]) == (lo + 1, lo + 1, 0, 0)
    nd = _eval_both(oracle, _plan([InputColDescriptor(capi.DOUBLE, False)], Expr.col(0).neg(capi.DOUBLE)), [_col_of(capi.DOUBLE, tiny)])
    assert nd[0] == nd[1] == struct.unpack("<q", struct.pack("<d", -float(np.finfo(np.float64).tiny)))[0] and nd[2:] == (0, 0)


def test_null_with_and_or_literals_of_the_reference(oracle):
    """Select.NullWithAndOr (Tests/ExecuteTest.cpp:1793-1868): `CAST(NULL AS BOOLEAN) AND val`, `val AND NULL`, `NULL OR val`,
    `val OR NULL` over table_bool_test's three rows (val = true, false, NULL) — the twelve ASSERT_EQ literals, BOOLEAN_NULL_SENTINEL
    = the INT8 NULL.  Both evaluators."""
    NUL = -128
    d = [InputColDescriptor(capi.INT8, True)]
    val = {1: 1, 2: 0, 3: NUL}     # id -> val
    null_b = Expr.null(capi.INT8)
    want = {("NULL AND val", 1): NUL, ("NULL AND val", 2): 0, ("NULL AND val", 3): NUL,
            ("val AND NULL", 1): NUL, ("val AND NULL", 2): 0, ("val AND NULL", 3): NUL,
            ("NULL OR val", 1): 1, ("NULL OR val", 2): NUL, ("NULL OR val", 3): NUL,
            ("val OR NULL", 1): 1, ("val OR NULL", 2): NUL, ("val OR NULL", 3): NUL}
    progs = {"NULL AND val": null_b.logical(capi.EX_AND, Expr.col(0)), "val AND NULL": Expr.col(0).logical(capi.EX_AND, null_b),
             "NULL OR val": null_b.logical(capi.EX_OR, Expr.col(0)), "val OR NULL": Expr.col(0).logical(capi.EX_OR, null_b)}
    for (name, row), w in want.items():
        ob, eb, oc, ec = _eval_both(oracle, _plan(d, progs[name]), [np.array([val[row]], dtype=np.int8)])
        assert (oc, ec, ob, eb) == (0, 0, w, w), (name, row, ob, eb, w)


def test_invalid_logic_programs_are_refused(oracle):
    from heavydb_amd.executor import ExprNode
    d = [InputColDescriptor(capi.INT32, True), InputColDescriptor(capi.INT8, True)]
    el = emu_lib()
    el.emu_eval_expr.restype = C.c_int32
    bad = [Expr.col(0).logical_not(),                                               # operand is not a BOOLEAN
           Expr.col(1).logical(capi.EX_AND, Expr.col(0)),                          # one operand is not
           Expr(Expr.col(1).nodes + [ExprNode(capi.EX_NOT, capi.INT32)]),          # result type must be INT8
           Expr(Expr.col(1).nodes + Expr.col(1).nodes + [ExprNode(capi.EX_OR, capi.INT8, null_lit=2)]),   # reserved: 0 / 1
           Expr(Expr.col(0).nodes + [ExprNode(capi.EX_IS_NULL, capi.INT32)]),      # result type must be INT8
           Expr.col(0).neg(capi.INT64),                                            # the node's type is the operand's
           Expr([ExprNode(capi.EX_NOT, capi.INT8)])]                               # empty stack
    for e in bad:
        plan = _plan(d, e)
        ptrs = (C.c_void_p * 2)(0, 0)
        eb, et = C.c_int64(), C.c_int32()
        assert el.emu_eval_expr(C.byref(plan), 0, ptrs, 0, C.byref(eb), C.byref(et)) == capi.ERR_INVALID_PLAN, [n.op for n in e.nodes]


def test_invalid_comparison_and_case_programs_are_refused(oracle):
    d = [InputColDescriptor(capi.INT32, True), InputColDescriptor(capi.INT64, True)]
    el = emu_lib()
    el.emu_eval_expr.restype = C.c_int32
    bad = [Expr.col(0).cmp(capi.EX_LT, Expr.col(1)),                                      # operands of two types
           Expr([*Expr.col(0).nodes, *Expr.col(0).nodes, capi and __import__("heavydb_amd.executor", fromlist=["ExprNode"]).ExprNode(capi.EX_LT, capi.INT32)]),  # result type must be INT8
           Expr.case(Expr.col(0), Expr.col(1), Expr.col(1), capi.INT64),                  # condition is not a BOOLEAN
           Expr.case(Expr.col(0).cmp(capi.EX_LT, Expr.col(0)), Expr.col(0), Expr.col(1), capi.INT64)]   # branches of two types
    for e in bad:
        plan = _plan(d, e)
        ptrs = (C.c_void_p * 2)(0, 0)
        eb, et = C.c_int64(), C.c_int32()
        assert el.emu_eval_expr(C.byref(plan), 0, ptrs, 0, C.byref(eb), C.byref(et)) == capi.ERR_INVALID_PLAN


def _random_bool(rng, descs, depth, big=True):
    """a random BOOLEAN: a comparison, IS NULL, NOT, AND / OR in both forms"""
    ints = [capi.INT8, capi.INT16, capi.INT32, capi.INT64]
    choice = int(rng.integers(0, 10)) if depth > 0 else 0
    if choice <= 4:
        ct = int(rng.choice(ints + [capi.DOUBLE, capi.FLOAT]))
        x, y = _random_expr(rng, descs, ct, depth - 1, big), _random_expr(rng, descs, ct, 0, big)
        if not (x and y):
            return None
        return x.cmp(int(rng.choice([capi.EX_EQ, capi.EX_NE, capi.EX_LT, capi.EX_LE, capi.EX_GT, capi.EX_GE])), y)
    if choice == 5:
        x = _random_expr(rng, descs, int(rng.choice(ints + [capi.DOUBLE, capi.FLOAT])), depth - 1, big)
        return x.is_null() if x else None
    if choice == 6:
        x = _random_bool(rng, descs, depth - 1, big)
        return x.logical_not() if x else None
    a, b = _random_bool(rng, descs, depth - 1, big), _random_bool(rng, descs, depth - 2, big)
    if not (a and b):
        return None
    return a.logical(int(rng.choice([capi.EX_AND, capi.EX_OR])), b, bool(rng.integers(0, 2)))


def _random_expr(rng, descs, want_type, depth, big=True):
    """a random well-typed expression of `want_type` over the columns (every micro-op; depth-limited so that the postfix
    program stays within 12 nodes and the 8-deep stack), or None when none was found"""
    ints = [capi.INT8, capi.INT16, capi.INT32, capi.INT64]
    cols_of = [i for i, d in enumerate(descs) if d.type == want_type]
    choice = int(rng.integers(0, 10))
    if depth <= 0 or choice <= 1:
        if cols_of and rng.integers(0, 4):
            return Expr.col(int(rng.choice(cols_of)))
        if want_type in (capi.DOUBLE, capi.FLOAT):
            return Expr.lit(want_type, float(rng.choice([0.0, 1.5, -2.25, 100.0, 1e6])))
        # (big=False: literals whose SUMs over a few thousand rows stay far from 2^63 — a partial sum that wraps onto the
        # NULL sentinel is dropped by the reduce rule of partial tables, in the reference too, and where the partials end
        # differs between the oracle's threads and the device's workgroups)
        hi = {capi.INT8: 100, capi.INT16: 30000, capi.INT32: 2**31 - 1 if big else 10**6, capi.INT64: 2**62 if big else 10**9}[want_type]
        if rng.integers(0, 12) == 0:
            return Expr.null(want_type)
        return Expr.lit(want_type, int(rng.choice([0, 1, -1, 2, 7, -13, hi, -hi])))
    if choice <= 3:      # cast from another type
        # (floating point -> integer is fptosi: undefined outside the target's range in the reference too — only to BIGINT here,
        # and the tables hold no value beyond 1e9)
        srcs = [t for t in ints + ([capi.DOUBLE, capi.FLOAT] if want_type not in ints or want_type == capi.INT64 else []) if t != want_type]
        src = int(rng.choice(srcs))
        e = _random_expr(rng, descs, src, depth - 1, big)
        return e.cast(want_type) if e else None
    if choice == 4 and rng.integers(0, 2):      # unary minus
        e = _random_expr(rng, descs, want_type, depth - 1, big)
        return e.neg(want_type) if e else None
    if choice <= 6:      # arithmetic
        op = int(rng.choice([capi.EX_ADD, capi.EX_SUB, capi.EX_MUL, capi.EX_DIV] + ([capi.EX_MOD] if want_type in ints else [])))
        a, b = _random_expr(rng, descs, want_type, depth - 1, big), _random_expr(rng, descs, want_type, depth - 2, big)
        return a._bin(op, b, want_type) if a and b else None
    if want_type == capi.INT8 and choice == 7:   # a BOOLEAN as the value itself
        return _random_bool(rng, descs, depth - 1, big)
    # CASE WHEN <boolean> THEN .. ELSE .. END
    c = _random_bool(rng, descs, depth - 1 if rng.integers(0, 3) == 0 else 1, big)
    t, e = _random_expr(rng, descs, want_type, depth - 2, big), _random_expr(rng, descs, want_type, depth - 2, big)
    if not (c and t and e):
        return None
    return Expr.case(c, t, e, want_type)


def _stack_depth(e):
    sp = mx = 0
    for n in e.nodes:
        if n.op in (capi.EX_COL, capi.EX_LIT):
            sp += 1
        elif n.op == capi.EX_CASE:
            sp -= 2
        elif n.op not in (capi.EX_CAST, capi.EX_NOT, capi.EX_IS_NULL, capi.EX_UMINUS):
            sp -= 1
        mx = max(mx, sp)
    return mx


def test_random_expression_programs_agree(oracle):
    """Random well-typed programs over every micro-op (casts, + - * / %, comparisons, nested CASE, the NULL literal) on rows
    with NULLs, zeros and extreme values: the oracle (recursive, lazy, from the root) and the product's evaluator (flat stack
    machine with a per-value error) must agree on the value AND on the error code, row by row."""
    import os
    rng = np.random.default_rng(int(os.environ.get("MI355Q_FUZZ_SEED", "20260923")))
    iters = int(os.environ.get("MI355Q_FUZZ_ITERS", "400"))
    types = [capi.INT8, capi.INT16, capi.INT32, capi.INT64, capi.DOUBLE, capi.FLOAT]
    n_rows = 64
    ol, el = oracle.lib(), emu_lib()
    ol.orc_eval_expr.restype = C.c_int32
    ol.orc_eval_expr.argtypes = [C.POINTER(capi.Plan), C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    el.emu_eval_expr.restype = C.c_int32
    el.emu_eval_expr.argtypes = [C.POINTER(capi.Plan), C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    done = errs = cases = 0
    for it in range(iters):
        descs, cols = [], []
        for t in types:
            for nullable in (True, False):
                if t in (capi.DOUBLE, capi.FLOAT):
                    v = rng.choice([0.0, -0.0, 1.0, -1.5, 3.25, 1e9, -1e-3, 2.5], n_rows).astype(NP[t])
                    if nullable:
                        v[rng.random(n_rows) < 0.25] = NP[t](np.finfo(NP[t]).tiny)
                else:
                    lo, hi = INT_NULL[t], INT_MAX[t]
                    v = rng.choice([0, 1, -1, 2, 5, -7, hi, hi - 1, lo + 1], n_rows).astype(NP[t])
                    if nullable:
                        v[rng.random(n_rows) < 0.25] = lo
                descs.append(InputColDescriptor(t, nullable))
                cols.append(np.ascontiguousarray(v))
        if rng.integers(0, 3) == 0:   # a BOOLEAN as the program itself: NOT / AND / OR (both forms) / IS NULL over comparisons
            e = _random_bool(rng, descs, int(rng.integers(1, 4)))
        else:
            e = _random_expr(rng, descs, int(rng.choice(types)), int(rng.integers(1, 4)))
        if e is None or len(e.nodes) > capi.MAX_EXPR_NODES or _stack_depth(e) > capi.MAX_EXPR_STACK:
            continue
        plan = _plan(descs, e)
        ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        rt = e.result(descs)[0]
        cases += 1
        for pos in range(n_rows):
            ob, ot, on, eb, et = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32()
            oc = ol.orc_eval_expr(C.byref(plan), 0, ptrs, pos, C.byref(ob), C.byref(ot), C.byref(on))
            ec = el.emu_eval_expr(C.byref(plan), 0, ptrs, pos, C.byref(eb), C.byref(et))
            assert oc == ec, (it, pos, oc, ec, [(n.op, n.type, n.arg) for n in e.nodes])
            if oc:
                errs += 1
                continue
            assert ot.value == et.value == rt, (it, ot.value, et.value, rt)
            same = _same_pattern(rt, ob.value, eb.value)
            if not same and rt in (capi.DOUBLE, capi.FLOAT):   # NaN payloads may differ
                f = (lambda p: struct.unpack("<d", struct.pack("<q", p))[0]) if rt == capi.DOUBLE else \
                    (lambda p: struct.unpack("<f", struct.pack("<I", p & 0xffffffff))[0])
                same = np.isnan(f(ob.value)) and np.isnan(f(eb.value))
            assert same, (it, pos, hex(ob.value), hex(eb.value), [(n.op, n.type, n.arg, n.ilit, n.flit) for n in e.nodes])
            done += 1
    assert cases > iters // 3 and done > 5000 and errs > 50, (cases, done, errs)

"""Generate tests/golden/ref_agg_vectors.json from the REFERENCE'S OWN aggregate and comparison runtime functions
(QueryEngine/RuntimeFunctions.cpp compiled unmodified into oracle/_ref/libref_runtime.so, oracle/Makefile).

Run in the build container only (needs /root/reference):  python oracle/gen_golden_agg.py
The committed JSON is what travels.  tests/test_oracle_golden.py::test_aggregates_match_reference_functions pins
oracle/oracle.cpp's row function to it on the CPU, tests/test_gpu_parity.py::test_hip_aggregates_match_reference_functions
pins the HIP library to it on the device.

Each "agg" case is one grouped step  SELECT AGG(v) FROM t GROUP BY k  with a single group (k = 0 everywhere) over a
short column v: the expected slot value is what the reference's own agg_* function — the one TargetExprBuilder's
codegenAggregate names for that target (TargetExprBuilder.cpp:600-760: base name, _double for floating-point
arguments, _skip_val with the argument type's inline NULL when the target skips NULLs) — leaves in a slot that started
at the descriptor's init value after being called once per row, in row order.
Each "cmp" case is  SELECT COUNT(*) FROM t WHERE v <op> literal : the expected count is the number of rows for which
the reference's <op>_<type>_nullable_lhs / plain comparison (DEF_CMP_NULLABLE_LHS, RuntimeFunctions.cpp:85-96) is > 0
(toBool).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heavydb_amd import capi  # noqa: E402
from heavydb_amd.executor import ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr  # noqa: E402
from oracle import oracle as orc  # noqa: E402

NP = {capi.INT8: np.int8, capi.INT16: np.int16, capi.INT32: np.int32, capi.INT64: np.int64, capi.DOUBLE: np.float64}
TNAME = {capi.INT8: "int8_t", capi.INT16: "int16_t", capi.INT32: "int32_t", capi.INT64: "int64_t", capi.DOUBLE: "double"}
INT_NULL = {capi.INT8: -2**7, capi.INT16: -2**15, capi.INT32: -2**31, capi.INT64: -2**63}
NULL_DOUBLE = float(np.finfo(np.float64).tiny)
AGG_NAME = {capi.COUNT: "count", capi.SUM: "sum", capi.MIN: "min", capi.MAX: "max"}


def dbl_bits(x: float) -> int:
    return struct.unpack("<q", struct.pack("<d", x))[0]


def step_unit(t, nullable, vals, agg):
    """GROUP BY k (always 0) -> perfect hash with one entry; the target under test owns the last slot."""
    nn = [v for v in vals if not (v == (NULL_DOUBLE if t == capi.DOUBLE else INT_NULL[t]) and nullable)]
    if t == capi.DOUBLE:
        rng_v = ExpressionRange(True, 0, 0, nullable, min(nn) if nn else 0.0, max(nn) if nn else 0.0)
    else:
        rng_v = ExpressionRange(True, int(min(nn)) if nn else 0, int(max(nn)) if nn else 0, nullable)
    descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 0)), InputColDescriptor(t, nullable, rng_v)]
    return RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT), TargetExpr(agg, 1)], [], [0], bigint_count=True)


def main():
    orc.build()
    ref = C.CDLL(orc.REF_LIB)
    rng = np.random.default_rng(2026)
    out = {"source": "QueryEngine/RuntimeFunctions.cpp of the reference, compiled unmodified (oracle/ref_shim.cpp)", "agg": [], "cmp": []}

    # ---- aggregates
    def ref_fold(agg, t, skip, init, vals):
        slot = C.c_int64(init)
        p = C.byref(slot)
        fp = t == capi.DOUBLE
        name = "agg_" + AGG_NAME[agg] + ("_double" if fp else "") + ("_skip_val" if skip else "")
        fn = getattr(ref, name)
        fn.restype = C.c_uint64 if agg == capi.COUNT else (C.c_int64 if agg == capi.SUM and not fp else None)
        # the argument as the call site passes it (TargetExprBuilder.cpp:486-495, 553-569): sign-extended to the
        # slot's 64 bits; for SUM / COUNT (not "domain range equivalent") convertNullIfAny first turns the argument
        # type's NULL into the NULL of the aggregate's own type (BIGINT here) and that is the skip value, while
        # MIN / MAX keep the argument type's NULL
        arg_null = NULL_DOUBLE if fp else INT_NULL[t]
        widen = skip and not fp and agg in (capi.SUM, capi.COUNT)
        null = NULL_DOUBLE if fp else (INT_NULL[capi.INT64] if widen else INT_NULL[t])
        for v in vals:
            if fp:
                fn.argtypes = [C.c_void_p, C.c_double] + ([C.c_double] if skip else [])
                fn(p, C.c_double(v), *([C.c_double(null)] if skip else []))
            else:
                arg = null if (widen and int(v) == arg_null) else int(v)
                fn.argtypes = [C.c_void_p, C.c_int64] + ([C.c_int64] if skip else [])
                fn(p, C.c_int64(arg), *([C.c_int64(null)] if skip else []))
        return name, slot.value

    for t in (capi.INT64, capi.INT32, capi.INT16, capi.INT8, capi.DOUBLE):
        for nullable in (False, True):
            for agg in (capi.COUNT, capi.SUM, capi.MIN, capi.MAX):
                for variant in range(4):
                    n = [1, 7, 40, 40][variant]
                    if t == capi.DOUBLE:
                        vals = [float(x) for x in (rng.random(n) - 0.5) * [1.0, 1e3, 1e-3, 1e12][variant]]
                        if variant == 3:
                            vals[3], vals[9] = -0.0, 0.0
                    else:
                        lo, hi = np.iinfo(NP[t]).min + 1, np.iinfo(NP[t]).max
                        span = [(lo, hi), (-50, 50), (lo // 4, hi // 4), (hi - 60, hi)][variant]
                        if t == capi.INT64:  # keep SUM inside int64
                            span = [(-2**40, 2**40), (-50, 50), (-2**60, 2**60), (2**62 - 60, 2**62)][variant]
                        vals = [int(x) for x in rng.integers(span[0], span[1], n, dtype=np.int64)]
                    if nullable:
                        null = NULL_DOUBLE if t == capi.DOUBLE else INT_NULL[t]
                        for i in range(0, n, 3):
                            vals[i] = null
                        if variant == 1:
                            vals = [null] * n   # a group whose argument is NULL in every row
                    ra = step_unit(t, nullable, vals, agg)
                    plan = ra.to_plan()
                    q = orc.qmd_init(plan)
                    slot = q.target_slot[1]
                    skip = bool(q.target_skip_null[1])
                    init = int(q.init_vals[slot])
                    name, want = ref_fold(agg, t, skip, init, vals)
                    out["agg"].append({"agg": AGG_NAME[agg], "type": TNAME[t], "nullable": nullable, "ref_function": name,
                                       "values": [dbl_bits(v) for v in vals] if t == capi.DOUBLE else vals,
                                       "values_are_double_bits": t == capi.DOUBLE, "slot": slot, "init": init, "want": want})

    # ---- FLOAT arguments: SUM / MIN / MAX run agg_*_float on the LOW FOUR BYTES of the 8-byte slot
    # (takes_float_argument -> agg_chosen_bytes = sizeof(float), TargetExprBuilder.cpp:477-483, 519-522); COUNT(float)
    # widens the value to double and counts with agg_count_double[_skip_val] ((double)NULL_FLOAT as the skip value)
    NULL_FLOAT = float(np.finfo(np.float32).tiny)
    for nullable in (False, True):
        for agg in (capi.COUNT, capi.SUM, capi.MIN, capi.MAX):
            for variant in range(3):
                n = [1, 9, 40][variant]
                vals = [float(np.float32(x)) for x in (rng.random(n) - 0.5) * [1.0, 1e3, 1e-3][variant]]
                if nullable:
                    for i in range(0, n, 3):
                        vals[i] = NULL_FLOAT
                    if variant == 1:
                        vals = [NULL_FLOAT] * n
                nn = [v for v in vals if not (nullable and v == NULL_FLOAT)]
                descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 0)),
                         InputColDescriptor(capi.FLOAT, nullable, ExpressionRange(True, 0, 0, nullable, min(nn) if nn else 0.0, max(nn) if nn else 0.0))]
                ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT), TargetExpr(agg, 1)], [], [0], bigint_count=True)
                q = orc.qmd_init(ra.to_plan())
                slot, skip, init = q.target_slot[1], bool(q.target_skip_null[1]), int(q.init_vals[q.target_slot[1]])
                cell = C.c_int64(init)
                if agg == capi.COUNT:
                    name = "agg_count_double" + ("_skip_val" if skip else "")
                    fn = getattr(ref, name)
                    fn.restype = C.c_uint64
                    fn.argtypes = [C.c_void_p, C.c_double] + ([C.c_double] if skip else [])
                    for v in vals:
                        fn(C.byref(cell), C.c_double(float(np.float32(v))), *([C.c_double(float(np.float32(NULL_FLOAT)))] if skip else []))
                else:
                    name = "agg_" + AGG_NAME[agg] + "_float" + ("_skip_val" if skip else "")
                    fn = getattr(ref, name)
                    fn.restype = None
                    fn.argtypes = [C.c_void_p, C.c_float] + ([C.c_float] if skip else [])
                    for v in vals:
                        fn(C.byref(cell), C.c_float(v), *([C.c_float(NULL_FLOAT)] if skip else []))
                out["agg"].append({"agg": AGG_NAME[agg], "type": "float", "nullable": nullable, "ref_function": name,
                                   "values": [int(np.float32(v).view(np.int32)) for v in vals], "values_are_float_bits": True,
                                   "values_are_double_bits": False, "slot": slot, "init": init, "want": cell.value})

    # ---- conditional aggregates: COUNT_IF(c < 3) and SUM_IF(v, c < 3) over a nullable condition column c.
    # cond = lt_int64_t_nullable_lhs(c, 3, NULL_BIGINT, -128) in {1, 0, -128}.  COUNT_IF: the (nullable BOOLEAN)
    # argument goes through convertNullIfAny -> NULL_BIGINT and agg_count_if_skip_val; SUM_IF: the condition becomes
    # (cond == 1) (codegenConditionalAggregateCondValSelector, WindowFunctionIR.cpp:1610-1637) and the value goes to
    # agg_sum_if[_skip_val] / agg_sum_if_double[_skip_val]
    lt = ref.lt_int64_t_nullable_lhs
    lt.restype = C.c_int8
    lt.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int8]
    NULL_BIG = INT_NULL[capi.INT64]
    out["cond"] = []
    for vt in (capi.INT64, capi.DOUBLE):
        for v_nullable in (False, True):
            for c_nullable in (False, True):
                n = 48
                cvals = [int(x) for x in rng.integers(-5, 9, n)]
                if c_nullable:
                    for i in range(1, n, 4):
                        cvals[i] = NULL_BIG
                if vt == capi.DOUBLE:
                    vvals = [float(x) for x in (rng.random(n) - 0.5) * 100.0]
                    vnull = NULL_DOUBLE
                else:
                    vvals = [int(x) for x in rng.integers(-1000, 1000, n)]
                    vnull = NULL_BIG
                if v_nullable:
                    for i in range(0, n, 5):
                        vvals[i] = vnull
                conds = [int(lt(c, 3, NULL_BIG, -128)) if c_nullable else int(c < 3) for c in cvals]
                nnv = [v for v in vvals if not (v_nullable and v == vnull)]
                nnc = [c for c in cvals if not (c_nullable and c == NULL_BIG)]
                descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 0)),
                         InputColDescriptor(vt, v_nullable, ExpressionRange(True, 0, 0, v_nullable, min(nnv), max(nnv)) if vt == capi.DOUBLE
                                            else ExpressionRange(True, int(min(nnv)), int(max(nnv)), v_nullable)),
                         InputColDescriptor(capi.INT64, c_nullable, ExpressionRange(True, min(nnc), max(nnc), c_nullable))]
                ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT_IF, cond=Qual(2, capi.LT, 3)),
                                                 TargetExpr(capi.SUM_IF, 1, cond=Qual(2, capi.LT, 3))], [], [0], bigint_count=True)
                q = orc.qmd_init(ra.to_plan())
                # COUNT_IF
                cell = C.c_int64(int(q.init_vals[q.target_slot[0]]))
                if c_nullable:
                    f = ref.agg_count_if_skip_val
                    f.restype = C.c_uint64
                    f.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
                    for c in conds:
                        f(C.byref(cell), NULL_BIG if c == -128 else c, NULL_BIG)
                    cname = "agg_count_if_skip_val"
                else:
                    f = ref.agg_count_if
                    f.restype = C.c_uint64
                    f.argtypes = [C.c_void_p, C.c_int64]
                    for c in conds:
                        f(C.byref(cell), c)
                    cname = "agg_count_if"
                want_count = cell.value
                # SUM_IF
                cell = C.c_int64(int(q.init_vals[q.target_slot[1]]))
                skip = bool(q.target_skip_null[1])
                sname = "agg_sum_if" + ("_double" if vt == capi.DOUBLE else "") + ("_skip_val" if skip else "")
                f = getattr(ref, sname)
                f.restype = None if vt == capi.DOUBLE else C.c_int64
                vt_c = C.c_double if vt == capi.DOUBLE else C.c_int64
                f.argtypes = [C.c_void_p, vt_c] + ([vt_c] if skip else []) + [C.c_int8]
                for v, c in zip(vvals, conds):
                    f(C.byref(cell), vt_c(v), *([vt_c(vnull)] if skip else []), C.c_int8(1 if c == 1 else 0))
                out["cond"].append({"value_type": TNAME[vt], "value_nullable": v_nullable, "cond_nullable": c_nullable,
                                    "ref_functions": [cname, sname],
                                    "values": [dbl_bits(v) for v in vvals] if vt == capi.DOUBLE else vvals,
                                    "values_are_double_bits": vt == capi.DOUBLE, "cond_values": cvals,
                                    "slots": [q.target_slot[0], q.target_slot[1]],
                                    "init": [int(q.init_vals[q.target_slot[0]]), int(q.init_vals[q.target_slot[1]])],
                                    "want": [want_count, cell.value]})

    # ---- multi-column perfect hash: get_matching_group_value_perfect_hash (RuntimeFunctions.cpp:2077-2091) writes the
    # key columns of entry `hashed_index` on first touch and returns the slots behind them
    gm = ref.get_matching_group_value_perfect_hash
    gm.restype = C.c_void_p
    gm.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    out["perfect_multi"] = []
    for key_count, rq, entries in ((2, 4, 12), (3, 5, 20), (4, 6, 9)):
        buf = np.full(entries * rq, 2**63 - 1, dtype=np.int64)
        for e in range(entries):
            buf[e * rq + key_count:(e + 1) * rq] = 0
        calls = []
        for _ in range(3 * entries):
            h = int(rng.integers(0, entries))
            key = np.array([h * 7 + g for g in range(key_count)], dtype=np.int64)   # one key tuple per entry
            p = gm(buf.ctypes.data, h, key.ctypes.data, key_count, rq)
            calls.append({"hashed_index": h, "key": key.tolist(), "returned_quad_offset": (p - buf.ctypes.data) // 8})
        out["perfect_multi"].append({"key_count": key_count, "row_size_quad": rq, "entry_count": entries, "calls": calls,
                                     "final_buffer": buf.tolist()})

    # ---- comparisons: DEF_CMP_NULLABLE_LHS (a nullable column against a literal) / the plain operator
    OPS = {"eq": capi.EQ, "ne": capi.NE, "lt": capi.LT, "gt": capi.GT, "le": capi.LE, "ge": capi.GE}
    for t in (capi.INT64, capi.INT32, capi.INT16, capi.INT8, capi.DOUBLE):
        for nullable in (False, True):
            for opn, opc in OPS.items():
                n = 64
                if t == capi.DOUBLE:
                    vals = [float(x) for x in np.round((rng.random(n) - 0.5) * 20.0)]
                    lit = 3.0
                    null = NULL_DOUBLE
                else:
                    vals = [int(x) for x in rng.integers(-10, 11, n)]
                    lit = 3
                    null = INT_NULL[t]
                if nullable:
                    for i in range(0, n, 5):
                        vals[i] = null
                passed = 0
                if nullable:
                    fn = getattr(ref, f"{opn}_{TNAME[t]}_nullable_lhs")
                    fn.restype = C.c_int8
                    ct = C.c_double if t == capi.DOUBLE else {capi.INT8: C.c_int8, capi.INT16: C.c_int16, capi.INT32: C.c_int32, capi.INT64: C.c_int64}[t]
                    nt = C.c_double if t == capi.DOUBLE else C.c_int64
                    fn.argtypes = [ct, ct, nt, C.c_int8]
                    for v in vals:
                        passed += int(fn(ct(v), ct(lit), nt(null), C.c_int8(-128)) > 0)
                    fname = f"{opn}_{TNAME[t]}_nullable_lhs"
                else:
                    import operator
                    f = {"eq": operator.eq, "ne": operator.ne, "lt": operator.lt, "gt": operator.gt, "le": operator.le, "ge": operator.ge}[opn]
                    passed = sum(int(f(v, lit)) for v in vals)
                    fname = "plain C++ operator (NOT NULL operands: codegenCmp emits icmp / fcmp)"
                out["cmp"].append({"op": opn, "op_code": opc, "type": TNAME[t], "nullable": nullable, "ref_function": fname,
                                   "values": [dbl_bits(v) for v in vals] if t == capi.DOUBLE else vals,
                                   "values_are_double_bits": t == capi.DOUBLE, "literal": lit, "want_count": passed})
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_agg_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["agg"]), "aggregate cases,", len(out["cond"]), "conditional cases,", len(out["cmp"]), "comparison cases")


if __name__ == "__main__":
    main()

"""Generate tests/golden/ref_expr_vectors.json from the REFERENCE'S OWN cast / arithmetic runtime functions
(QueryEngine/RuntimeFunctions.cpp compiled unmodified into oracle/_ref/libref_runtime.so, oracle/Makefile).

Run in the build container only (needs /root/reference):  python oracle/gen_golden_expr.py
The committed JSON is what travels.  tests/test_expr.py pins the oracle's expression evaluator
(oracle/oracle.cpp eval_expression) and the product's (heavydb_amd/csrc/expr.h, through the host emulation
on the CPU and through the projection kernel on the device) to it.

  "cast"   cast_<from>_to_<to>_nullable(operand, from_null, to_null)        RuntimeFunctions.cpp:262-330
           (integer <-> integer, integer -> float / double, float <-> double, float / double -> integer with
           DEF_ROUND_NULLABLE's rounding) — what codegenCast emits for a nullable operand (CastIR.cpp:424-653);
           the NOT NULL forms (sext / sitofp / fpext / fptrunc / round + fptosi) are the same functions on
           values that are not the sentinel
  "arith"  {add,sub,mul}_<type>_nullable[_lhs|_rhs](lhs, rhs, null)          RuntimeFunctions.cpp:46-71,118-131
           for int8_t .. int64_t, float, double — what codegenArith emits with a nullable operand
           (ArithmeticIR.cpp:187-429).  Integer operand pairs are chosen so that the exact result fits the
           type: the overflow check itself is an LLVM intrinsic (s{add,sub,mul}.with.overflow,
           ArithmeticIR.cpp:840-909), not a runtime function, and is tested against exact integer arithmetic.
  "cmp"    {eq,ne,lt,le,gt,ge}_<type>_nullable[_lhs|_rhs](lhs, rhs, null, null_bool)  RuntimeFunctions.cpp:73-107,132-149
           for int8_t .. int64_t, float, double (NaN operands included) — what codegenCmp emits for two values with a
           nullable operand (CompareIR.cpp:230-330); `out` is the int8 result: 1 / 0 / -128
  "logic"  logical_not(operand, null_bool) / logical_and(lhs, rhs, null_bool) / logical_or(..)   RuntimeFunctions.cpp:331-358
           over {1, 0, -128}: what codegenLogical emits for a nullable BOOLEAN operand (LogicalIR.cpp:299-379)
  "uminus" uminus_<type>_nullable(operand, null)                                RuntimeFunctions.cpp:247-258
           for int8_t .. int64_t, float, double — what codegenUMinus emits for a nullable operand (ArithmeticIR.cpp:787-838);
           the type's minimum IS the NULL (a NOT NULL operand holding it is an overflow error before any function)
Values travel as 64-bit patterns: integers sign-extended, double bits, float bits in the low word.
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
from oracle import oracle as orc  # noqa: E402

TNAME = {capi.INT8: "int8_t", capi.INT16: "int16_t", capi.INT32: "int32_t", capi.INT64: "int64_t",
         capi.DOUBLE: "double", capi.FLOAT: "float"}
CT = {capi.INT8: C.c_int8, capi.INT16: C.c_int16, capi.INT32: C.c_int32, capi.INT64: C.c_int64,
      capi.DOUBLE: C.c_double, capi.FLOAT: C.c_float}
INTS = [capi.INT8, capi.INT16, capi.INT32, capi.INT64]
INT_NULL = {capi.INT8: -2**7, capi.INT16: -2**15, capi.INT32: -2**31, capi.INT64: -2**63}
INT_MAX = {capi.INT8: 2**7 - 1, capi.INT16: 2**15 - 1, capi.INT32: 2**31 - 1, capi.INT64: 2**63 - 1}
NULL_DOUBLE = float(np.finfo(np.float64).tiny)
NULL_FLOAT = float(np.finfo(np.float32).tiny)


def null_of(t):
    return NULL_DOUBLE if t == capi.DOUBLE else NULL_FLOAT if t == capi.FLOAT else INT_NULL[t]


def bits(t, v) -> int:
    if t == capi.DOUBLE:
        return struct.unpack("<q", struct.pack("<d", v))[0]
    if t == capi.FLOAT:
        return struct.unpack("<I", struct.pack("<f", v))[0]
    return int(v)


def int_samples(t, rng):
    lo, hi = INT_NULL[t], INT_MAX[t]
    base = [0, 1, -1, 2, -7, 100, -100, hi, hi - 1, lo, lo + 1, lo + 2]
    base += [int(x) for x in rng.integers(max(lo, -10**6), min(hi, 10**6), 6)]
    return sorted({v for v in base if lo <= v <= hi})


def fp_samples(t, rng):
    base = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 0.49999, 123456.789, -98765.4321, 1e-3,
            null_of(t), 3.0e9 if t == capi.DOUBLE else 3.0e6]
    base += [float(x) for x in rng.normal(0, 1000, 6)]
    if t == capi.FLOAT:
        base = [float(np.float32(x)) for x in base]
    return base


def main():
    orc.build()
    ref = C.CDLL(orc.REF_LIB)
    rng = np.random.default_rng(2027)
    out = {"source": "QueryEngine/RuntimeFunctions.cpp of the reference, compiled unmodified (oracle/ref_shim.cpp)",
           "cast": [], "arith": [], "cmp": []}
    # ---- casts
    pairs = [(a, b) for a in INTS for b in INTS if a != b]
    pairs += [(a, b) for a in INTS for b in (capi.FLOAT, capi.DOUBLE)]
    pairs += [(capi.FLOAT, capi.DOUBLE), (capi.DOUBLE, capi.FLOAT)]
    pairs += [(a, b) for a in (capi.FLOAT, capi.DOUBLE) for b in INTS]
    for f, t in pairs:
        fn = getattr(ref, f"cast_{TNAME[f]}_to_{TNAME[t]}_nullable")
        fn.restype = CT[t]
        fn.argtypes = [CT[f], CT[f], CT[t]]
        vals = int_samples(f, rng) if f in INTS else fp_samples(f, rng)
        for v in vals:
            if f in INTS and t in INTS and INT_MAX[t] < INT_MAX[f] and v != INT_NULL[f] and not (INT_NULL[t] < v <= INT_MAX[t]):
                continue  # a narrowing cast of this value is an overflow error before the function is reached
            if f not in INTS and t in INTS and v != null_of(f) and not (INT_NULL[t] + 1 < v < INT_MAX[t] - 1):
                continue  # fptosi out of range: undefined in the reference as well
            r = fn(v, null_of(f), null_of(t))
            out["cast"].append({"from": f, "to": t, "in": bits(f, v), "out": bits(t, r)})
    # ---- arithmetic
    ops = {"add": capi.EX_ADD, "sub": capi.EX_SUB, "mul": capi.EX_MUL, "div": capi.EX_DIV, "mod": capi.EX_MOD}
    for t in INTS + [capi.FLOAT, capi.DOUBLE]:
        vals = int_samples(t, rng) if t in INTS else fp_samples(t, rng)
        for name, op in ops.items():
            if name == "mod" and t not in INTS:
                continue
            for suffix in ("_nullable", "_nullable_lhs", "_nullable_rhs"):
                fn = getattr(ref, f"{name}_{TNAME[t]}{suffix}")
                fn.restype = CT[t]
                # the NULL argument is int64_t for the integer forms (DEF_BINARY_NULLABLE_ALL_OPS(type, int64_t))
                fn.argtypes = [CT[t], CT[t], C.c_int64 if t in INTS else CT[t]]
                a_list = [null_of(t)] + [v for v in vals[::2] if v != null_of(t)]
                b_list = [null_of(t)] + [v for v in vals[1::4] if v != null_of(t)]
                for a in a_list:
                    for b in b_list:
                        a_null = suffix in ("_nullable", "_nullable_lhs") and a == null_of(t)
                        b_null = suffix in ("_nullable", "_nullable_rhs") and b == null_of(t)
                        if name in ("div", "mod"):
                            # what the generated code lets reach the function: a non-zero divisor (the zero check sits in
                            # front — skipped, for DIV, behind a NULL pattern, where the function returns NULL or would trap)
                            if name == "mod" and b == 0:
                                continue  # codegenMod tests the divisor first, whatever the NULLs
                            if (b == 0 or b == 0.0) and not (a_null or b_null):
                                continue
                            if t in INTS and (b == 0 or (a == INT_NULL[t] and b == -1)) and not ((suffix != "_nullable_rhs" and a_null) or b_null):
                                continue  # INT_MIN / 0 and INT_MIN / -1: SIGFPE in the reference
                        elif t in INTS and not (a_null or b_null):
                            exact = a + b if name == "add" else a - b if name == "sub" else a * b
                            if not (INT_NULL[t] <= exact <= INT_MAX[t]):
                                continue  # signed overflow: the check fires first
                        r = fn(a, b, null_of(t))
                        out["arith"].append({"op": op, "type": t, "suffix": suffix, "a": bits(t, a), "b": bits(t, b),
                                             "out": bits(t, r)})
    # ---- comparisons of two values
    cmps = {"eq": capi.EX_EQ, "ne": capi.EX_NE, "lt": capi.EX_LT, "le": capi.EX_LE, "gt": capi.EX_GT, "ge": capi.EX_GE}
    for t in INTS + [capi.FLOAT, capi.DOUBLE]:
        vals = int_samples(t, rng) if t in INTS else fp_samples(t, rng) + [float("nan"), float("inf"), -float("inf")]
        pick = vals[::3] + [null_of(t)]
        for name, op in cmps.items():
            for suffix in ("_nullable", "_nullable_lhs", "_nullable_rhs"):
                fn = getattr(ref, f"{name}_{TNAME[t]}{suffix}")
                fn.restype = C.c_int8
                fn.argtypes = [CT[t], CT[t], C.c_int64 if t in INTS else CT[t], C.c_int8]
                for a in pick:
                    for b in pick:
                        r = fn(a, b, null_of(t), -128)
                        out["cmp"].append({"op": op, "type": t, "suffix": suffix, "a": bits(t, a), "b": bits(t, b), "out": int(r)})
    # ---- NOT / AND / OR over nullable BOOLEANs
    out["logic"] = []
    ref.logical_not.restype = C.c_int8
    ref.logical_not.argtypes = [C.c_int8, C.c_int8]
    for a in (1, 0, -128):
        out["logic"].append({"op": capi.EX_NOT, "a": a, "b": 0, "out": int(ref.logical_not(a, -128))})
    for name, op in (("logical_and", capi.EX_AND), ("logical_or", capi.EX_OR)):
        fn = getattr(ref, name)
        fn.restype = C.c_int8
        fn.argtypes = [C.c_int8, C.c_int8, C.c_int8]
        for a in (1, 0, -128):
            for b in (1, 0, -128):
                out["logic"].append({"op": op, "a": a, "b": b, "out": int(fn(a, b, -128))})
    # ---- unary minus
    out["uminus"] = []
    for t in INTS + [capi.FLOAT, capi.DOUBLE]:
        fn = getattr(ref, f"uminus_{TNAME[t]}_nullable")
        fn.restype = CT[t]
        fn.argtypes = [CT[t], CT[t]]
        vals = int_samples(t, rng) if t in INTS else fp_samples(t, rng) + [float("inf"), -float("inf")]
        for v in vals:
            out["uminus"].append({"type": t, "in": bits(t, v), "out": bits(t, fn(v, null_of(t)))})
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "ref_expr_vectors.json")
    with open(path, "w") as fjs:
        json.dump(out, fjs, separators=(",", ":"))
    print(f"wrote {path}: {len(out['cast'])} cast + {len(out['arith'])} arithmetic + {len(out['cmp'])} comparison + "
          f"{len(out['logic'])} logic + {len(out['uminus'])} unary-minus vectors")


if __name__ == "__main__":
    main()

"""Generate tests/golden/ref_layout_vectors.json from the REFERENCE'S OWN layout code
(oracle/_ref/libref_layout.so = QueryEngine/OutputBufferInitialization.cpp, Descriptors/ColSlotContext.cpp,
Shared/TargetInfo.cpp, QueryEngine/CalciteDeserializerUtils.cpp compiled unmodified; oracle/ref_layout_shim.cpp
says what is glue and why QueryMemoryDescriptor.cpp itself cannot be built here).

Run in the build container only (needs /root/reference):  python oracle/gen_golden_layout.py
The committed JSON is what travels.  tests/test_ref_layout.py pins oracle/oracle.cpp's qmd_init and the product's
heavydb_amd/csrc/plan.cpp (through the host emulation) to it:

  per target   TargetInfo.sql_type / agg_arg_type / skip_null_val / is_agg     (get_target_info_impl)
  per slot     logical and padded size; slot count; getCompactByteWidth; getAllSlotsAlignedPaddedSize
  per slot     init_agg_val_vec(target_exprs, quals, descriptor)                (a7)

For every random plan the QueryDescriptionType, keyless flag, columnar flag and the slot size
pick_target_compact_width chose are taken from the oracle's descriptor and handed to the reference code as inputs
(those decisions live in GroupByAndAggregate.cpp / QueryMemoryDescriptor.cpp, which need LLVM headers to compile);
everything listed above is then computed by the reference.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heavydb_amd import capi  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from heavydb_amd.executor import RelAlgExecutionUnit, TargetExpr  # noqa: E402
from tests.test_plan_fuzz import _random_col, _random_plan  # noqa: E402

# SQLTypes (Shared/sqltypes.h:65-95)
kBOOLEAN, kINT, kSMALLINT, kFLOAT, kDOUBLE, kBIGINT, kDATE, kTINYINT = 1, 6, 7, 8, 9, 12, 14, 22
SQL_OF = {capi.INT8: kTINYINT, capi.INT16: kSMALLINT, capi.INT32: kINT, capi.INT64: kBIGINT,
          capi.DOUBLE: kDOUBLE, capi.FLOAT: kFLOAT}


def sql_type_of(desc) -> int:
    """SQLTypes of the ColumnVar a column descriptor stands for (its logical type: what the decoders hand the
    row function).  Dictionary ids are 32-bit integers on this path; DATE in days decodes to 8-byte kDATE."""
    if desc.encoding == capi.ENC_FIXED:
        return SQL_OF[desc.logical_type]
    if desc.encoding == capi.ENC_DICT:
        return kINT
    if desc.encoding == capi.ENC_DATE_IN_DAYS:
        return kDATE
    return SQL_OF[desc.type]


class RefTarget(C.Structure):
    _fields_ = [("agg", C.c_int32), ("arg_type", C.c_int32), ("arg_notnull", C.c_int32), ("arg_col", C.c_int32),
                ("reserved0", C.c_int32), ("reserved1", C.c_int32), ("key_index", C.c_int32), ("reserved", C.c_int32)]


class RefQual(C.Structure):
    _fields_ = [("col", C.c_int32), ("col_type", C.c_int32), ("negated", C.c_int32), ("reserved", C.c_int32)]


class RefOut(C.Structure):
    _fields_ = [("n_slots", C.c_int32), ("n_init", C.c_int32), ("compact_width", C.c_int32),
                ("aligned_padded_size", C.c_int32),
                ("ti_sql_type", C.c_int32 * 32), ("ti_sql_notnull", C.c_int32 * 32), ("ti_arg_type", C.c_int32 * 32),
                ("ti_skip_null", C.c_int32 * 32), ("ti_is_agg", C.c_int32 * 32),
                ("slot_logical", C.c_int32 * 64), ("slot_padded", C.c_int32 * 64), ("init_vals", C.c_int64 * 64)]


def unit_to_json(ra) -> dict:
    d = dataclasses.asdict(ra)
    for k in ("join_table", "inner_col_descs", "exprs", "join_outer_col", "join_kind"):
        d.pop(k, None)
    return d


def ref_inputs(ra, qmd, rng):
    """(targets, quals) in the shim's vocabulary.  `col IS NOT NULL` reaches the reference in either of the two
    spellings constrained_not_null recognises (chosen at random); `col IS NULL` as the bare UOper it must ignore;
    comparison quals are BinOpers it skips and are left out."""
    targets = []
    for t in ra.target_exprs:
        if t.agg == capi.PROJECT_KEY:
            g = max(t.col, 0)
            c = ra.groupby_exprs[g]
            d = ra.input_col_descs[c]
            targets.append(RefTarget(-1, sql_type_of(d), int(not d.nullable), c, 0, 0, g, 0))
        elif t.agg == capi.COUNT_IF:
            # the argument is the condition itself, a boolean BinOper that is nullable when its column operand is
            # (only its SQLTypeInfo matters here; a boolean ColumnVar that no qual names stands in for it)
            d = ra.input_col_descs[t.cond.col]
            targets.append(RefTarget(t.agg, kBOOLEAN, int(not d.nullable), 1000 + t.cond.col, 0, 0, -1, 0))
        elif t.col < 0:
            targets.append(RefTarget(t.agg, -1, 1, -1, 0, 0, -1, 0))
        else:
            d = ra.input_col_descs[t.col]
            targets.append(RefTarget(t.agg, sql_type_of(d), int(not d.nullable), t.col, 0, 0, -1, 0))
    quals = []
    for q in ra.simple_quals:
        if q.op == capi.IS_NOT_NULL:
            quals.append(RefQual(q.col, sql_type_of(ra.input_col_descs[q.col]), int(rng.integers(0, 2)), 0))
        elif q.op == capi.IS_NULL:
            quals.append(RefQual(q.col, sql_type_of(ra.input_col_descs[q.col]), 2, 0))
    return targets, quals


def run_reference(ref, ra, qmd, rng) -> dict | None:
    targets, quals = ref_inputs(ra, qmd, rng)
    ta = (RefTarget * max(len(targets), 1))(*targets)
    qa = (RefQual * max(len(quals), 1))(*quals)
    out = RefOut()
    baseline = qmd.desc_type == capi.GROUP_BY_BASELINE_HASH
    rc = ref.ref_layout_run(ta, len(targets), qa, len(quals), qmd.desc_type, qmd.group_col_count, qmd.keyless,
                            qmd.output_columnar, int(ra.bigint_count), qmd.slot_width, int(baseline), C.byref(out))
    if rc != 0:
        return None
    n, s = len(targets), out.n_slots
    return {"n_slots": s, "compact_width": out.compact_width, "aligned_padded_size": out.aligned_padded_size,
            "ti_sql_type": list(out.ti_sql_type[:n]), "ti_sql_notnull": list(out.ti_sql_notnull[:n]),
            "ti_arg_type": list(out.ti_arg_type[:n]), "ti_skip_null": list(out.ti_skip_null[:n]),
            "ti_is_agg": list(out.ti_is_agg[:n]),
            "slot_logical": list(out.slot_logical[:s]), "slot_padded": list(out.slot_padded[:s]),
            "init_vals": [str(v) for v in out.init_vals[:out.n_init]]}


def narrow_plan(rng):
    """The shapes pick_target_compact_width narrows: one group column, COUNT(*) and key projections only."""
    descs = [_random_col(rng) for _ in range(3)]
    ints = [i for i, d in enumerate(descs) if d.type not in (capi.DOUBLE, capi.FLOAT)]
    if not ints:
        return None
    targets = [TargetExpr(capi.COUNT) if rng.integers(0, 3) else TargetExpr(capi.PROJECT_KEY, 0)
               for _ in range(int(rng.integers(1, 4)))]
    return RelAlgExecutionUnit(descs, targets, [], [int(rng.choice(ints))],
                               max_groups_buffer_entry_guess=int(rng.choice([0, 1000, 16384])),
                               bigint_count=bool(rng.integers(0, 6) == 0),
                               num_tuples=int(rng.choice([0, 10 ** 6, 2 ** 32 - 1, 2 ** 32])))


def main():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_layout.so"))
    rng = np.random.default_rng(20260922)
    cases, seen = [], {}
    tries = 0
    while len(cases) < 720 and tries < 20000:
        tries += 1
        ra = _random_plan(rng) if len(cases) < 600 else narrow_plan(rng)
        if ra is None:
            continue
        if rng.integers(0, 5) == 0:
            ra.output_columnar_hint = capi.OUTPUT_COLUMNAR
        plan = ra.to_plan()
        qmd = capi.QMD()
        if orc.lib().orc_qmd_init(C.byref(plan), C.byref(qmd)) != 0:
            continue
        if len(ra.target_exprs) > 32:
            continue
        exp = run_reference(ref, ra, qmd, rng)
        if exp is None:
            continue
        key = (qmd.desc_type, qmd.keyless, qmd.slot_width, qmd.output_columnar)
        seen[key] = seen.get(key, 0) + 1
        cases.append({"unit": unit_to_json(ra),
                      "given": {"desc_type": qmd.desc_type, "keyless": qmd.keyless, "slot_width": qmd.slot_width,
                                "output_columnar": qmd.output_columnar, "group_col_count": qmd.group_col_count},
                      "ref": exp})
    path = os.path.join(ROOT, "tests", "golden", "ref_layout_vectors.json")
    with open(path, "w") as f:
        json.dump({"generator": "oracle/gen_golden_layout.py", "cases": cases}, f,
                  separators=(",", ":"))
    print(f"{len(cases)} cases ({tries} plans tried), kinds {sorted(seen.items())}")
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

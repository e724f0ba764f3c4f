#!/usr/bin/env python
"""Known-answer vectors for the PROJECTION runtime, produced by the reference's OWN functions compiled in place
(oracle/_ref/libref_runtime.so: QueryEngine/GroupByRuntime.cpp get_scan_output_slot :242-255, get_columnar_scan_output_offset
:257-269, and the agg_id family of RuntimeFunctions.cpp the targets are written with) -> tests/golden/ref_projection_vectors.json.

A trace = a sequence of (old_total_matched, offset_in_fragment, target values) as the row function of a projection step
issues them (GroupByAndAggregate.cpp:1080-1101, :1255-1275; TargetExprBuilder.cpp:330-560), the returned slot / offset of
every call (-1: the buffer is full — the row function then answers -pos) and the final buffer image.

    python oracle/gen_golden_projection.py        (needs /root/reference; run in the build container)"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

EMPTY = 2**63 - 1


def main():
    orc.build()
    ref = orc.ref_lib()
    assert ref is not None, "oracle/_ref not built (no /root/reference?)"
    ref.get_scan_output_slot.restype = C.c_void_p
    ref.get_scan_output_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int64, C.c_uint32]
    ref.get_columnar_scan_output_offset.restype = C.c_int32
    ref.get_columnar_scan_output_offset.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int64]
    for name, ct in (("agg_id", C.c_int64), ("agg_id_int32", C.c_int32), ("agg_id_int16", C.c_int16), ("agg_id_int8", C.c_int8)):
        getattr(ref, name).restype = None
        getattr(ref, name).argtypes = [C.c_void_p, ct]
    ref.agg_id_double.restype = None
    ref.agg_id_double.argtypes = [C.c_void_p, C.c_double]
    ref.agg_id_float.restype = None
    ref.agg_id_float.argtypes = [C.c_void_p, C.c_float]
    rng = np.random.default_rng(20250923)
    out = {"source": "heavyai/heavydb GroupByRuntime.cpp / RuntimeFunctions.cpp compiled in place (oracle/ref_shim.cpp)"}

    # ---- row-wise: entry = [offset | 8-byte slots]; integers sign-extended to the slot, doubles as their bits
    rowwise = []
    for entry_count, n_slots, n_calls in ((8, 2, 11), (5, 3, 5), (1, 1, 3), (16, 1, 9)):
        rq = 1 + n_slots
        buf = np.zeros(entry_count * rq, dtype=np.int64)
        buf.reshape(entry_count, rq)[:, 0] = EMPTY
        calls = []
        for old in range(n_calls):
            pos = int(rng.integers(0, 2**40))
            ivals = [int(x) for x in rng.integers(-2**62, 2**62, n_slots)]
            dval = float(rng.normal())
            p = ref.get_scan_output_slot(buf.ctypes.data, entry_count, old, pos, rq)
            slot = -1 if not p else (p - buf.ctypes.data) // 8
            if p:
                for s in range(n_slots):
                    if s == 1:
                        ref.agg_id_double(p + 8 * s, dval)
                    else:
                        ref.agg_id(p + 8 * s, ivals[s])
            calls.append({"old_total_matched": old, "offset_in_fragment": pos, "ivals": ivals, "dval": dval, "slot_quad": int(slot)})
        rowwise.append({"entry_count": entry_count, "row_size_quad": rq, "calls": calls, "final": [int(x) for x in buf]})
    out["rowwise"] = rowwise

    # ---- columnar: the key column [entry_count] int64, then one column per target of its logical width
    columnar = []
    for entry_count, n_calls in ((6, 9), (3, 3), (12, 7)):
        keys = np.full(entry_count, EMPTY, dtype=np.int64)
        c8 = np.zeros(entry_count, dtype=np.int8)
        c16 = np.zeros(entry_count, dtype=np.int16)
        c32 = np.zeros(entry_count, dtype=np.int32)
        cf = np.zeros(entry_count, dtype=np.float32)
        calls = []
        for old in range(n_calls):
            pos = int(rng.integers(0, 2**33))
            v8, v16, v32 = int(rng.integers(-128, 128)), int(rng.integers(-2**15, 2**15)), int(rng.integers(-2**31, 2**31))
            vf = float(np.float32(rng.normal()))
            off = ref.get_columnar_scan_output_offset(keys.ctypes.data, entry_count, old, pos)
            if off >= 0:
                ref.agg_id_int8(c8.ctypes.data + off, v8)
                ref.agg_id_int16(c16.ctypes.data + 2 * off, v16)
                ref.agg_id_int32(c32.ctypes.data + 4 * off, v32)
                ref.agg_id_float(cf.ctypes.data + 4 * off, vf)
            calls.append({"old_total_matched": old, "offset_in_fragment": pos, "v8": v8, "v16": v16, "v32": v32, "vf": vf, "offset": int(off)})
        columnar.append({"entry_count": entry_count, "calls": calls, "keys": [int(x) for x in keys], "c8": [int(x) for x in c8],
                         "c16": [int(x) for x in c16], "c32": [int(x) for x in c32], "cf_bits": [int(x) for x in cf.view(np.int32)]})
    out["columnar"] = columnar

    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_projection_vectors.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()

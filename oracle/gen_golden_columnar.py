#!/usr/bin/env python
"""Generates tests/golden/ref_columnar_vectors.json from the REFERENCE's own columnar runtime
functions (QueryEngine/GroupByRuntime.cpp get_group_value_columnar_slot :84-105,
get_columnar_group_bin_offset :227-239, compiled in place into oracle/_ref by oracle/Makefile).
TEST INFRASTRUCTURE; run in the build container (needs /root/reference), the JSON is committed."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

EMPTY64 = 2**63 - 1


def main():
    ref = orc.ref_lib()
    assert ref is not None, "oracle/_ref not built (no /root/reference?)"
    ref.get_group_value_columnar_slot.restype = C.c_int32
    ref.get_group_value_columnar_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    ref.get_columnar_group_bin_offset.restype = C.c_uint32
    ref.get_columnar_group_bin_offset.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    rng = np.random.default_rng(20260923)
    out = {"source": "heavyai/heavydb GroupByRuntime.cpp compiled in place (oracle/ref_shim.cpp)"}

    def baseline(entry_count, key_count, keys):
        buf = np.full(entry_count * key_count, EMPTY64, dtype=np.int64)   # the prepended key columns
        bins = []
        for k in keys:
            kb = np.array(k, dtype=np.int64)
            bins.append(int(ref.get_group_value_columnar_slot(buf.ctypes.data, entry_count, kb.ctypes.data,
                                                              key_count, 8)))
        return {"entry_count": entry_count, "key_count": key_count, "keys": [[int(x) for x in k] for k in keys],
                "bins": bins, "final_key_columns": [int(x) for x in buf]}

    traces = [baseline(8, 1, [[10], [20], [30], [10], [40], [50], [20], [60]]),
              baseline(8, 1, [[k] for k in range(1, 10)])]                    # the ninth key finds no bin: -1
    keys = [[int(a) * 1000003 + 7, int(b)] for a, b in zip(rng.integers(0, 30, 150), rng.integers(-3, 3, 150))]
    traces.append(baseline(211, 2, keys))
    keys3 = [[int(a), int(b), int(c)] for a, b, c in zip(rng.integers(-2**40, 2**40, 64), rng.integers(0, 4, 64),
                                                           rng.integers(0, 2, 64))]
    traces.append(baseline(97, 3, keys3 + keys3[:20]))
    out["columnar_baseline_traces"] = traces

    def perfect(min_key, bucket, n, keys):
        col = np.full(n, EMPTY64, dtype=np.int64)
        bins = [int(ref.get_columnar_group_bin_offset(col.ctypes.data, k, min_key, bucket)) for k in keys]
        return {"min_key": min_key, "bucket": bucket, "entries": n, "keys": [int(k) for k in keys], "bins": bins,
                "final_key_column": [int(x) for x in col]}

    out["columnar_perfect_traces"] = [
        perfect(100, 0, 5, [102, 100, 104, 102]),
        perfect(-7, 0, 32, [int(x) for x in rng.integers(-7, 25, 80)]),
        perfect(18000 * 86400, 86400, 20, [int(d) * 86400 for d in rng.integers(18000, 18020, 50)])]
    path = os.path.join(ROOT, "tests", "golden", "ref_columnar_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()

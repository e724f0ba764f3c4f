#!/usr/bin/env python
"""Runs the reference's OWN join-build code (oracle/_ref/libref_join.so = QueryEngine/JoinHashTable/Runtime/
HashJoinRuntime.cpp compiled in place, see oracle/ref_join_shim.cpp) over random inner key columns and writes
tests/golden/ref_join_build_vectors.json: for every case the inputs and the hash join buffer the reference filled
(perfect OneToOne / OneToMany, keyed OneToOne / OneToMany with 1 - 3 key components of 4 or 8 bytes, NULL keys,
duplicates).  TEST INFRASTRUCTURE: tests/test_ref_join_build.py holds the oracle's join build (oracle.cpp) to these
buffers byte for byte — both fill single-threaded in row order, so even the slot positions of the keyed tables agree.

    python oracle/gen_golden_join.py            # needs /root/reference (build container only)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NP = {4: np.int32, 8: np.int64}


def ref_lib():
    path = os.path.join(ROOT, "oracle", "_ref", "libref_join.so")
    lib = C.CDLL(path)
    lib.ref_join_perfect.restype = C.c_int32
    lib.ref_join_perfect.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    lib.ref_join_keyed.restype = C.c_int32
    lib.ref_join_keyed.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]
    return lib


def ref_perfect(lib, keys, elem_sz, lo, hi, one_to_many):
    k = np.ascontiguousarray(keys, dtype=NP[elem_sz])
    entries = hi - lo + 1
    out = np.zeros(entries if not one_to_many else 2 * entries + len(k), dtype=np.int32)
    err = lib.ref_join_perfect(k.ctypes.data, elem_sz, len(k), lo, hi, int(one_to_many), out.ctypes.data)
    return err, out


def ref_keyed(lib, cols, elem_szs, width, entries, one_to_many):
    ks = [np.ascontiguousarray(c, dtype=NP[e]) for c, e in zip(cols, elem_szs)]
    n = len(ks[0])
    ptrs = (C.c_void_p * len(ks))(*[k.ctypes.data for k in ks])
    szs = (C.c_int32 * len(ks))(*elem_szs)
    nbytes = entries * len(ks) * width + (entries * 8 + n * 4 if one_to_many else entries * width)
    out = np.zeros(nbytes, dtype=np.uint8)
    err = lib.ref_join_keyed(ptrs, szs, len(ks), n, width, entries, int(one_to_many), out.ctypes.data)
    return err, out


def cases(rng):
    out = []
    # ---- perfect hash: dense-ish key ranges, with and without duplicates and NULLs
    for i in range(40):
        elem = int(rng.choice([4, 8]))
        lo = int(rng.integers(-50, 50))
        span = int(rng.integers(1, 200))
        n = int(rng.integers(0, 150))
        dup = bool(rng.integers(0, 2))
        if dup:
            keys = rng.integers(lo, lo + span, n)
        else:
            keys = rng.permutation(np.arange(lo, lo + span))[:n]
        keys = keys.astype(NP[elem])
        if len(keys) and rng.integers(0, 3) == 0:   # NULL keys are kept out of the table
            keys[rng.integers(0, len(keys), max(1, len(keys) // 7))] = np.iinfo(NP[elem]).min
        out.append(dict(kind="perfect", elem_sz=elem, min=lo, max=lo + span - 1, keys=[int(x) for x in keys], one_to_many=int(dup)))
    # ---- keyed (baseline) tables: sparse keys, 1 - 3 components
    for i in range(40):
        nk = int(rng.integers(1, 4))
        width = int(rng.choice([4, 8]))
        n = int(rng.integers(1, 120))
        dup = bool(rng.integers(0, 2))
        cols, szs = [], []
        for k in range(nk):
            elem = width if width == 4 else int(rng.choice([4, 8]))
            pool = rng.integers(-10**6 if elem == 4 else -10**12, 10**6 if elem == 4 else 10**12, max(2, n // (3 if dup else 1) + 1))
            col = rng.choice(pool, n).astype(NP[elem])
            if not dup and k == 0:
                col = (np.arange(n) * 977 + int(rng.integers(0, 1000))).astype(NP[elem])   # distinct first component
                rng.shuffle(col)
            if rng.integers(0, 4) == 0:
                col[rng.integers(0, n, max(1, n // 9))] = np.iinfo(NP[elem]).min
            cols.append([int(x) for x in col])
            szs.append(elem)
        # key component width as the reference derives it (BaselineJoinHashTable::getKeyComponentWidth: 8 bytes as soon as
        # one key column is 8 bytes wide, else 4)
        width = 8 if 8 in szs else 4
        entries = 2 * max(n, 1)
        out.append(dict(kind="keyed", elem_sz=szs, width=width, entries=entries, cols=cols, one_to_many=int(dup)))
    return out


def run_case(lib, c):
    if c["kind"] == "perfect":
        err, buf = ref_perfect(lib, c["keys"], c["elem_sz"], c["min"], c["max"], c["one_to_many"])
        return err, [int(x) for x in buf]
    err, buf = ref_keyed(lib, c["cols"], c["elem_sz"], c["width"], c["entries"], c["one_to_many"])
    return err, [int(x) for x in buf]


def main():
    lib = ref_lib()
    rng = np.random.default_rng(20260923)
    vec = []
    for c in cases(rng):
        err, buf = run_case(lib, c)
        c["err"] = int(err)
        c["buffer"] = buf
        vec.append(c)
    path = os.path.join(ROOT, "tests", "golden", "ref_join_build_vectors.json")
    with open(path, "w") as f:
        json.dump(dict(source="QueryEngine/JoinHashTable/Runtime/HashJoinRuntime.cpp compiled in place (oracle/ref_join_shim.cpp)",
                       cases=vec), f, separators=(",", ":"))
    print(f"{len(vec)} cases -> {path} ({os.path.getsize(path) // 1024} KB)")


if __name__ == "__main__":
    main()

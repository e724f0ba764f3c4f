"""Generate tests/golden/ref_vectors.json from the REFERENCE'S OWN runtime functions
(oracle/_ref/libref_runtime.so, compiled in place from /root/reference by oracle/Makefile).

Run in the build container only (needs /root/reference):  python oracle/gen_golden.py
The committed JSON is what travels; tests/test_oracle_golden.py pins oracle/oracle.cpp to it.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

EMPTY64 = 2**63 - 1
EMPTY32 = 2**31 - 1


def main():
    orc.build()
    ref = orc.ref_lib()
    assert ref is not None, "oracle/_ref not built (no /root/reference?)"
    rng = np.random.default_rng(20260922)
    out = {"source": "heavyai/heavydb reference sources compiled in place (oracle/ref_shim.cpp)"}

    # ---- Murmur vectors
    hashes = []
    ints64 = [0, 1, 7, 42, 1000, 1000010, 9999999 * 1000003 + 7, -1, 2**40, -(2**63), 2**63 - 1]
    ints64 += [int(x) for x in rng.integers(-2**62, 2**62, 24)]
    for v in ints64:
        b = struct.pack("<q", v)
        hashes.append({"hex": b.hex(), "m3": ref.MurmurHash3(b, 8, 0), "m1": ref.MurmurHash1(b, 8, 0)})
    for v in [0, 1, 7, 999, -1, 2**31 - 1, -(2**31)] + [int(x) for x in rng.integers(-2**31, 2**31, 16)]:
        b = struct.pack("<i", v)
        hashes.append({"hex": b.hex(), "m3": ref.MurmurHash3(b, 4, 0), "m1": ref.MurmurHash1(b, 4, 0)})
    b = struct.pack("<qq", 3, 5)
    hashes.append({"hex": b.hex(), "m3": ref.MurmurHash3(b, 16, 0), "m1": ref.MurmurHash1(b, 16, 0)})
    for n in range(1, 20):  # tails of every length, non-zero seeds
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        seed = int(rng.integers(0, 2**32))
        hashes.append({"hex": b.hex(), "seed": seed, "m3": ref.MurmurHash3(b, n, seed),
                       "m1": ref.MurmurHash1(b, n, seed)})
    out["hashes"] = hashes

    # ---- baseline group-by traces: get_group_value (GroupByRuntime.cpp:25-48)
    def baseline_trace(entry_count, key_width, keys, row_quad=2):
        buf = np.zeros(entry_count * row_quad, dtype=np.int64)
        for e in range(entry_count):
            if key_width == 8:
                buf[e * row_quad] = EMPTY64
            else:
                buf[e * row_quad:e * row_quad + 1].view(np.int32)[0] = EMPTY32
        landed = []
        for k in keys:
            if key_width == 8:
                kb = np.array([k], dtype=np.int64)
            else:
                kb = np.array([k, 0], dtype=np.int32)
            p = ref.get_group_value(buf.ctypes.data, entry_count, kb.ctypes.data, 1, key_width,
                                    row_quad)
            if not p:
                landed.append(-1)
                continue
            quad = (p - buf.ctypes.data) // 8
            landed.append(int(quad))          # quad index of the first agg slot
            buf[quad] += 1                    # COUNT
        return {"entry_count": entry_count, "key_width": key_width, "row_quad": row_quad,
                "keys": [int(k) for k in keys], "slot_quads": landed,
                "final": [int(x) for x in buf]}

    traces = [baseline_trace(8, 8, [10, 20, 30, 10, 40, 50, 20, 60])]
    traces.append(baseline_trace(8, 8, [1, 2, 3, 4, 5, 6, 7, 8, 9]))  # overflow -> NULL
    keys = [int(x) * 1000003 + 7 for x in rng.integers(0, 40, 200)]
    traces.append(baseline_trace(97, 8, keys, row_quad=4))
    keys32 = [int(x) for x in rng.integers(-1000, 1000, 300)]
    traces.append(baseline_trace(1024, 4, keys32, row_quad=3))
    out["baseline_traces"] = traces

    # ---- perfect group-by trace: get_group_value_fast (GroupByRuntime.cpp:208-223)
    def perfect_trace(min_key, n, keys, row_quad=2):
        buf = np.zeros(n * row_quad, dtype=np.int64)
        buf[0::row_quad] = EMPTY64
        for k in keys:
            p = ref.get_group_value_fast(buf.ctypes.data, k, min_key, 0, row_quad)
            quad = (p - buf.ctypes.data) // 8
            buf[quad] += k  # SUM(key)
        return {"min_key": min_key, "entries": n, "row_quad": row_quad,
                "keys": [int(k) for k in keys], "final": [int(x) for x in buf]}

    out["perfect_traces"] = [perfect_trace(100, 5, [102, 100, 104, 102]),
                             perfect_trace(-7, 32, [int(x) for x in rng.integers(-7, 25, 100)], 3)]

    # ---- bucketed perfect trace (DATE keys, bucket 86400): get_group_value_fast with a bucket
    def perfect_bucket_trace(min_key, bucket, n, keys, row_quad=2):
        buf = np.zeros(n * row_quad, dtype=np.int64)
        buf[0::row_quad] = EMPTY64
        for k in keys:
            p = ref.get_group_value_fast(buf.ctypes.data, k, min_key, bucket, row_quad)
            buf[(p - buf.ctypes.data) // 8] += 1
        return {"min_key": min_key, "bucket": bucket, "entries": n, "row_quad": row_quad,
                "keys": [int(k) for k in keys], "final": [int(x) for x in buf]}

    days = [int(x) for x in rng.integers(18000, 18020, 60)]
    out["perfect_bucket_traces"] = [perfect_bucket_trace(18000 * 86400, 86400, 20, [d * 86400 for d in days])]

    # ---- multi-column baseline traces: get_group_value with key_count components
    def multi_trace(entry_count, key_width, key_count, keys, row_quad):
        dt = np.int64 if key_width == 8 else np.int32
        buf = np.zeros(entry_count * row_quad, dtype=np.int64)
        kq = (key_count * key_width + 7) // 8
        for e in range(entry_count):
            kv = buf[e * row_quad:e * row_quad + kq].view(dt)
            kv[:key_count] = EMPTY64 if key_width == 8 else EMPTY32
        landed = []
        for k in keys:
            kb = np.zeros(kq * (8 // key_width), dtype=dt)
            kb[:key_count] = k
            p = ref.get_group_value(buf.ctypes.data, entry_count, kb.ctypes.data, key_count, key_width,
                                    row_quad)
            if not p:
                landed.append(-1)
                continue
            quad = (p - buf.ctypes.data) // 8
            landed.append(int(quad))
            buf[quad] += 1
        return {"entry_count": entry_count, "key_width": key_width, "key_count": key_count,
                "row_quad": row_quad, "keys": [[int(x) for x in k] for k in keys],
                "slot_quads": landed, "final": [int(x) for x in buf]}

    mt = []
    k2 = [[int(a) * 1000003 + 7, int(b)] for a, b in zip(rng.integers(0, 12, 150), rng.integers(-3, 3, 150))]
    mt.append(multi_trace(131, 8, 2, k2, 4))
    k3 = [[int(a), int(b), int(c)] for a, b, c in zip(rng.integers(-40, 40, 200), rng.integers(0, 3, 200),
                                                       rng.integers(1000, 1004, 200))]
    mt.append(multi_trace(257, 4, 3, k3, 3))       # 12 key bytes + 4 padding
    mt.append(multi_trace(16, 4, 2, [[i, -i] for i in range(20)], 2))  # overflow -> NULL
    out["multi_baseline_traces"] = mt

    # ---- encoded-column decoders (DecodersImpl.h:57-85 unsigned, :130-139 small date)
    u8 = np.array([0, 1, 127, 128, 200, 254, 255], dtype=np.uint8)
    u16 = np.array([0, 1, 32767, 32768, 65534, 65535], dtype=np.uint16)
    out["unsigned_decode"] = [
        {"width": 1, "hex": u8.tobytes().hex(),
         "decoded": [int(ref.fixed_width_unsigned_decode(u8.ctypes.data, 1, i)) for i in range(len(u8))]},
        {"width": 2, "hex": u16.tobytes().hex(),
         "decoded": [int(ref.fixed_width_unsigned_decode(u16.ctypes.data, 2, i)) for i in range(len(u16))]}]
    d32 = np.array([0, 1, -1, 18000, -(2**31), 2**31 - 1], dtype=np.int32)
    d16 = np.array([0, 1, -1, 18000, -(2**15), 2**15 - 1], dtype=np.int16)
    out["small_date_decode"] = [
        {"width": 4, "hex": d32.tobytes().hex(),
         "decoded": [int(ref.fixed_width_small_date_decode(d32.ctypes.data, 4, -(2**31), -(2**63), i))
                     for i in range(len(d32))]},
        {"width": 2, "hex": d16.tobytes().hex(),
         "decoded": [int(ref.fixed_width_small_date_decode(d16.ctypes.data, 2, -(2**15), -(2**63), i))
                     for i in range(len(d16))]}]

    # ---- perfect join probe: hash_join_idx (GroupByRuntime.cpp:287-297)
    table = np.full(5, -1, dtype=np.int32)
    for row_id, k in enumerate([3, 1, 4]):
        table[k - 1] = row_id
    probes = [1, 2, 3, 4, 5, 0, 9, -5]
    out["perfect_join"] = {
        "min": 1, "max": 5, "table": [int(x) for x in table], "probes": probes,
        "idx": [int(ref.hash_join_idx(table.ctypes.data, k, 1, 5)) for k in probes]}

    # ---- keyed join probe: baseline_hash_join_idx_64 (JoinHashTableQueryRuntime.cpp:56-94)
    def keyed_join(entry_count, dim_keys, probes):
        tab = np.zeros((entry_count, 2), dtype=np.int64)
        tab[:, 0] = EMPTY64
        tab[:, 1] = -1
        slots = []
        for row_id, k in enumerate(dim_keys):
            kb = struct.pack("<q", k)
            h = ref.MurmurHash1(kb, 8, 0) % entry_count
            while tab[h, 0] != EMPTY64:
                h = (h + 1) % entry_count
            tab[h] = (k, row_id)
            slots.append(int(h))
        res = []
        for k in probes:
            kb = np.array([k], dtype=np.int64)
            res.append(int(ref.baseline_hash_join_idx_64(tab.ctypes.data, kb.ctypes.data, 8,
                                                         entry_count)))
        return {"entry_count": entry_count, "dim_keys": [int(k) for k in dim_keys],
                "slots": slots, "table": [int(x) for x in tab.reshape(-1)],
                "probes": [int(k) for k in probes], "idx": res}

    kj = [keyed_join(8, [1000010, 2000013, 3000016, 7], [1000010, 7, 3000016, 8, 2000013, 0])]
    dk = [int(x) * 1000003 for x in rng.permutation(500)[:200]]
    pr = dk[:50] + [int(x) for x in rng.integers(0, 10**9, 50)]
    kj.append(keyed_join(400, dk, pr))
    kj.append(keyed_join(4, [5, 6, 7, 9], [5, 9, 11]))  # full table: -1 (kNoMatch) on miss
    out["keyed_join"] = kj

    # ---- composite keyed tables: baseline_hash_join_idx_{32,64} with several key components
    # and get_composite_key_index_{32,64} (JoinHashTableQueryRuntime.cpp:35-94,140-163).  The
    # tables are laid down here with the reference's MurmurHash1 + linear probing in row order.
    def composite(entry_count, width, dim_rows, probes, with_payload):
        dt = np.int32 if width == 4 else np.int64
        empty = EMPTY32 if width == 4 else EMPTY64
        kc = len(dim_rows[0])
        stride = kc + (1 if with_payload else 0)
        tab = np.zeros((entry_count, stride), dtype=dt)
        tab[:, :kc] = empty
        if with_payload:
            tab[:, kc] = -1
        for row_id, k in enumerate(dim_rows):
            kb = np.array(k, dtype=dt)
            h = ref.MurmurHash1(kb.ctypes.data, kc * width, 0) % entry_count
            while tab[h, 0] != empty and not (tab[h, :kc] == kb).all():
                h = (h + 1) % entry_count
            tab[h, :kc] = kb
            if with_payload:
                tab[h, kc] = row_id
        res = []
        for k in probes:
            kb = np.array(k, dtype=dt)
            if with_payload:
                f = ref.baseline_hash_join_idx_32 if width == 4 else ref.baseline_hash_join_idx_64
                res.append(int(f(tab.ctypes.data, kb.ctypes.data, kc * width, entry_count)))
            else:
                f = ref.get_composite_key_index_32 if width == 4 else ref.get_composite_key_index_64
                res.append(int(f(kb.ctypes.data, kc, tab.ctypes.data, entry_count)))
        return {"entry_count": entry_count, "width": width, "key_count": kc, "with_payload": with_payload,
                "dim_rows": [[int(x) for x in k] for k in dim_rows], "table": [int(x) for x in tab.reshape(-1)],
                "probes": [[int(x) for x in k] for k in probes], "idx": res}

    ck = []
    rows2 = [[int(a), int(b)] for a, b in zip(rng.permutation(300)[:120], rng.integers(-5, 5, 120))]
    pr2 = rows2[:40] + [[int(a), int(b)] for a, b in zip(rng.integers(0, 300, 40), rng.integers(-5, 5, 40))]
    ck.append(composite(240, 4, rows2, pr2, True))
    rows3 = [[int(a) * 1000003, int(b), int(c)] for a, b, c in zip(rng.permutation(200)[:90], rng.integers(0, 3, 90),
                                                                    rng.integers(-2**40, 2**40, 90))]
    pr3 = rows3[:30] + [[r[0], r[1], r[2] + 1] for r in rows3[:20]]
    ck.append(composite(180, 8, rows3, pr3, True))
    dup = [[int(a), int(b)] for a, b in zip(rng.integers(0, 30, 150), rng.integers(0, 3, 150))]  # repeats
    ck.append(composite(300, 4, dup, dup[:40] + [[99, 99], [0, 7]], False))
    ck.append(composite(300, 8, [[d[0] * 10**10, d[1]] for d in dup], [[d[0] * 10**10, d[1]] for d in dup[:40]] +
                        [[5, 5]], False))
    ck.append(composite(6, 4, [[1, 1], [3, 3], [0, 0]], [[1, 1], [3, 3], [0, 0], [2, 2]], True))  # hash_joins.rst
    out["composite_keyed"] = ck

    # ---- decoders: fixed_width_int_decode / fixed_width_double_decode (DecodersImpl.h:27-55,121)
    dec = []
    for width, dt in [(1, np.int8), (2, np.int16), (4, np.int32), (8, np.int64)]:
        info = np.iinfo(dt)
        vals = np.array([0, 1, -1, info.min, info.max, 37, -99], dtype=dt)
        dec.append({"width": width, "hex": vals.tobytes().hex(),
                    "decoded": [int(ref.fixed_width_int_decode(vals.ctypes.data, width, i))
                                for i in range(len(vals))]})
    out["int_decode"] = dec
    dv = np.array([0.0, -1.5, 2.2250738585072014e-308, 1e300, 999.999], dtype=np.float64)
    out["double_decode"] = {"hex": dv.tobytes().hex(),
                            "decoded": [float(ref.fixed_width_double_decode(dv.ctypes.data, i))
                                        for i in range(len(dv))]}

    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests",
                       "golden", "ref_vectors.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()

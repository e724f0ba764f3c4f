"""Pin oracle/oracle.cpp to the reference: golden vectors generated from the reference's own
sources (tests/golden/ref_vectors.json, oracle/gen_golden.py), the SURVEY (c') literals, and
— when oracle/_ref is present — the reference functions themselves, live."""
import ctypes as C
import struct

import numpy as np
import pytest

EMPTY64 = 2**63 - 1
EMPTY32 = 2**31 - 1


def test_murmur_golden(oracle, golden):
    for h in golden["hashes"]:
        b = bytes.fromhex(h["hex"])
        seed = h.get("seed", 0)
        assert oracle.murmur3(b, seed) == h["m3"], h
        assert oracle.murmur1(b, seed) == h["m1"], h


def test_murmur_survey_literals(oracle):
    # SURVEY.md section 8(c'): values produced from the reference sources
    lit = {0: (1669671676, 533107803), 1: (1392991556, 2619614180), 7: (4157363267, 3494347355),
           42: (1871679806, 575224328), 1000: (2827443468, 3092276255),
           1000010: (3537811470, 3303879907), 10000029000004: (4167831984, 3923777440),
           -1: (1651860712, 1905253887), 2**40: (2851483426, 2855848196)}
    for k, (m3, m1) in lit.items():
        b = struct.pack("<q", k)
        assert oracle.murmur3(b) == m3 and oracle.murmur1(b) == m1
    assert oracle.murmur3(struct.pack("<i", 0)) == 593689054  # canonical MurmurHash3_x86_32
    assert oracle.murmur3(struct.pack("<qq", 3, 5)) == 2388078947
    assert oracle.murmur1(struct.pack("<qq", 3, 5)) == 2049968836


def _baseline_run(oracle, tr):
    n, kw, rq = tr["entry_count"], tr["key_width"], tr["row_quad"]
    buf = np.zeros(n * rq, dtype=np.int64)
    for e in range(n):
        if kw == 8:
            buf[e * rq] = EMPTY64
        else:
            buf[e * rq:e * rq + 1].view(np.int32)[0] = EMPTY32
    landed = []
    for k in tr["keys"]:
        s = oracle.lib().orc_get_group_value_slot(buf.ctypes.data, n, k, kw, rq)
        landed.append(int(s))
        if s >= 0:
            buf[s] += 1
    return landed, [int(x) for x in buf]


def test_baseline_groupby_traces(oracle, golden):
    for tr in golden["baseline_traces"]:
        landed, final = _baseline_run(oracle, tr)
        assert landed == tr["slot_quads"]
        assert final == tr["final"]
    # SURVEY (c') literal: keys 10,20,30,10,40,50,20,60 in 8 entries
    tr = golden["baseline_traces"][0]
    assert [q // 2 for q in tr["slot_quads"]] == [4, 3, 5, 4, 6, 0, 3, 7]


def test_perfect_groupby_traces(oracle, golden):
    for tr in golden["perfect_traces"]:
        rq = tr["row_quad"]
        buf = np.zeros(tr["entries"] * rq, dtype=np.int64)
        buf[0::rq] = EMPTY64
        for k in tr["keys"]:
            s = oracle.lib().orc_get_group_value_fast_slot(buf.ctypes.data, k, tr["min_key"], rq)
            buf[s] += k
        assert [int(x) for x in buf] == tr["final"]


def test_perfect_bucketed_traces(oracle, golden):
    """get_group_value_fast with a bucket (DATE keys): reference-generated trace."""
    for tr in golden["perfect_bucket_traces"]:
        rq = tr["row_quad"]
        buf = np.zeros(tr["entries"] * rq, dtype=np.int64)
        buf[0::rq] = EMPTY64
        for k in tr["keys"]:
            s = oracle.lib().orc_get_group_value_fast_bucket_slot(buf.ctypes.data, k, tr["min_key"],
                                                                  tr["bucket"], rq)
            buf[s] += 1
        assert [int(x) for x in buf] == tr["final"]


def test_multi_column_baseline_traces(oracle, golden):
    """get_group_value over 2- and 3-component keys of 8 and 4 bytes (incl. the padded 12-byte
    key and a full table), slot by slot against the reference's own function."""
    for tr in golden["multi_baseline_traces"]:
        n, kw, kc, rq = tr["entry_count"], tr["key_width"], tr["key_count"], tr["row_quad"]
        dt = np.int64 if kw == 8 else np.int32
        kq = (kc * kw + 7) // 8
        buf = np.zeros(n * rq, dtype=np.int64)
        for e in range(n):
            buf[e * rq:e * rq + kq].view(dt)[:kc] = EMPTY64 if kw == 8 else EMPTY32
        landed = []
        for k in tr["keys"]:
            kb = np.array(k, dtype=dt)
            s = oracle.lib().orc_get_group_value_n_slot(buf.ctypes.data, n, kb.ctypes.data, kc, kw, rq)
            landed.append(int(s))
            if s >= 0:
                buf[s] += 1
        assert landed == tr["slot_quads"]
        assert [int(x) for x in buf] == tr["final"]


def test_encoded_decoders(oracle, golden):
    """fixed_width_unsigned_decode / fixed_width_small_date_decode vectors from the reference
    == the oracle's column fetch for ENC_DICT (NOT NULL: no sentinel widening) and
    ENC_DATE_IN_DAYS columns."""
    import ctypes as C
    from heavydb_amd import capi
    tmap = {1: capi.INT8, 2: capi.INT16, 4: capi.INT32}
    for d in golden["unsigned_decode"]:
        raw = np.frombuffer(bytes.fromhex(d["hex"]), dtype=np.uint8).copy()
        cd = capi.ColDesc(tmap[d["width"]], 0, capi.ENC_DICT, 0)
        got = [int(oracle.lib().orc_decode_col(C.byref(cd), raw.ctypes.data, i)) for i in range(len(d["decoded"]))]
        assert got == d["decoded"]
        # nullable: the all-ones id is NULL and widens to NULL_INT
        cdn = capi.ColDesc(tmap[d["width"]], 1, capi.ENC_DICT, 0)
        gotn = [int(oracle.lib().orc_decode_col(C.byref(cdn), raw.ctypes.data, i)) for i in range(len(d["decoded"]))]
        top = 255 if d["width"] == 1 else 65535
        assert gotn == [(-2**31 if v == top else v) for v in d["decoded"]]
    for d in golden["small_date_decode"]:
        raw = np.frombuffer(bytes.fromhex(d["hex"]), dtype=np.uint8).copy()
        cd = capi.ColDesc(tmap[d["width"]], 1, capi.ENC_DATE_IN_DAYS, 0)
        got = [int(oracle.lib().orc_decode_col(C.byref(cd), raw.ctypes.data, i)) for i in range(len(d["decoded"]))]
        assert got == d["decoded"]
    # ENC_FIXED: sign-extending load (pinned by int_decode above) + NULL widening
    v = np.array([5, -7, -(2**15), 2**15 - 1], dtype=np.int16)
    cd = capi.ColDesc(capi.INT16, 1, capi.ENC_FIXED, capi.INT64)
    assert [int(oracle.lib().orc_decode_col(C.byref(cd), v.ctypes.data, i)) for i in range(4)] == \
        [5, -7, -(2**63), 2**15 - 1]
    cd = capi.ColDesc(capi.INT16, 0, capi.ENC_FIXED, capi.INT32)
    assert [int(oracle.lib().orc_decode_col(C.byref(cd), v.ctypes.data, i)) for i in range(4)] == \
        [5, -7, -(2**15), 2**15 - 1]


def test_perfect_join_probe(oracle, golden):
    pj = golden["perfect_join"]
    j = oracle.OracleJoin(np.array([3, 1, 4], dtype=np.int64), 4, pj["min"], pj["max"])
    assert j.info()["hash_type"] == 0
    assert [int(x) for x in j.buffer()] == pj["table"]
    assert [j.probe(k) for k in pj["probes"]] == pj["idx"]


def test_keyed_join_build_and_probe(oracle, golden):
    for kj in golden["keyed_join"][:2]:
        keys = np.array(kj["dim_keys"], dtype=np.int64)
        # the oracle sizes keyed tables at 2 x rows like the reference (2 x NDV)
        j = oracle.OracleJoin(keys, 4, 0, -1, prefer_baseline=True)
        assert j.info() == {"hash_type": 1, "entry_count": kj["entry_count"]}
        assert j.shape()["component_width"] == 8
        tab = j.buffer()
        assert [int(x) for x in tab.reshape(-1)] == kj["table"]  # serial insert order == ref
        assert [j.probe(k) for k in kj["probes"]] == kj["idx"]
    # the two distinct "no match" values of the reference: -2 at an empty slot
    assert golden["keyed_join"][0]["idx"][3] == -2


def test_composite_keyed_tables(oracle, golden):
    """Keyed join tables with several key components, 4- and 8-byte: the oracle's build lays the
    keys (and one-to-one payloads) down exactly where the reference's MurmurHash1 probing puts
    them, baseline_hash_join_idx_{32,64} / get_composite_key_index_{32,64} agree on every probe,
    and a one-to-many table returns each key's rows.  Includes the 6-entry table of
    docs/source/execution/hash_joins.rst."""
    from heavydb_amd import capi
    for ck in golden["composite_keyed"]:
        kc, w, n = ck["key_count"], ck["width"], ck["entry_count"]
        rows = np.array(ck["dim_rows"], dtype=np.int64)
        t = capi.INT32 if w == 4 else capi.INT64
        j = oracle.OracleJoin([rows[:, i] for i in range(kc)], [t] * kc, 0, -1, prefer_baseline=True,
                              one_to_many=0 if ck["with_payload"] else 2, keyed_entry_count=n)
        assert j.info() == {"hash_type": 1 if ck["with_payload"] else 3, "entry_count": n}
        assert j.shape()["key_components"] == kc and j.shape()["component_width"] == w
        stride = kc + (1 if ck["with_payload"] else 0)
        dt = np.int32 if w == 4 else np.int64
        tab = j.raw()[:n * stride * w].view(dt)
        assert [int(x) for x in tab] == ck["table"]
        for probe, idx in zip(ck["probes"], ck["idx"]):
            m = j.matches(probe)
            want_rows = [i for i, r in enumerate(ck["dim_rows"]) if r == probe]
            if ck["with_payload"]:
                assert m == ([idx] if idx >= 0 else []) and m == want_rows
            else:
                assert (idx >= 0) == bool(m) and m == want_rows  # serial build: row order
    # the literal of hash_joins.rst: keys land in slots 1, 2, 3 of 6
    doc = golden["composite_keyed"][-1]
    assert doc["table"][3:12] == [1, 1, 0, 3, 3, 1, 0, 0, 2]


def test_one_to_many_perfect_doc_example(oracle):
    """hash_joins.rst 'One-To-Many JoinHashTable Example': table2 = (0, 1, 3, 3) ->
    | offsets 0 1 * 2 | counts 1 1 * 2 | payloads 0 1 2 3 |; and the keyed variant
    | keys * (1,1) (3,3) (0,0) * * | offsets * 0 1 3 * * | counts * 1 2 1 * * | payloads 1 2 3 0 |."""
    from heavydb_amd import capi
    b = np.array([0, 1, 3, 3], dtype=np.int32)
    j = oracle.OracleJoin(b, capi.INT32, 0, 3, one_to_many=1)  # OneToOne fails on the duplicate -> rebuilt
    assert j.info() == {"hash_type": 2, "entry_count": 4}
    assert [int(x) for x in j.raw().view(np.int32)] == [0, 1, -1, 2, 1, 1, 0, 2, 0, 1, 2, 3]
    assert j.matches(3) == [2, 3] and j.matches(2) == [] and j.matches(7) == []
    with pytest.raises(capi.Mi355qError) as ei:
        oracle.OracleJoin(b, capi.INT32, 0, 3, one_to_many=0)
    assert ei.value.code == capi.ERR_JOIN_NOT_ONE_TO_ONE
    # 'One-To-One BaselineJoinHashTable Example': table2 = (0, 1, 3), key (b, b) ->
    # | keys * (1,1,1) (3,3,2) (0,0,0) * * |   (payload = row id interleaved with the key)
    b3 = np.array([0, 1, 3], dtype=np.int32)
    j = oracle.OracleJoin([b3, b3], [capi.INT32, capi.INT32], 0, -1, one_to_many=0, keyed_entry_count=6)
    assert j.info() == {"hash_type": 1, "entry_count": 6}
    E = EMPTY32
    assert [int(x) for x in j.raw().view(np.int32)] == [E, E, -1, 1, 1, 1, 3, 3, 2, 0, 0, 0, E, E, -1, E, E, -1]
    j = oracle.OracleJoin([b, b], [capi.INT32, capi.INT32], 0, -1, one_to_many=1, keyed_entry_count=6)
    assert j.info() == {"hash_type": 3, "entry_count": 6}
    raw = j.raw().view(np.int32)
    E = EMPTY32
    assert [int(x) for x in raw[:12]] == [E, E, 1, 1, 3, 3, 0, 0, E, E, E, E]
    assert [int(x) for x in raw[12:18]] == [-1, 0, 1, 3, -1, -1]
    assert [int(x) for x in raw[18:24]] == [0, 1, 2, 1, 0, 0]
    assert [int(x) for x in raw[24:28]] == [1, 2, 3, 0]


def test_decoders(oracle, golden):
    # decode through a non-grouped MIN/MAX/SUM over each width == reference decode
    from heavydb_amd import capi
    from heavydb_amd.executor import InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    tmap = {1: capi.INT8, 2: capi.INT16, 4: capi.INT32, 8: capi.INT64}
    for d in golden["int_decode"]:
        t = tmap[d["width"]]
        vals = np.frombuffer(bytes.fromhex(d["hex"]), dtype=oracle.NP_DTYPE[t])
        for i, want in enumerate(d["decoded"]):
            ra = RelAlgExecutionUnit([InputColDescriptor(t)],
                                     [TargetExpr(capi.MAX, 0)])
            q, buf, code = oracle.execute(ra.to_plan(), [[vals[i:i + 1]]])
            assert code == 0
            if want == oracle.NP_DTYPE[t](np.iinfo(oracle.NP_DTYPE[t]).min):
                continue  # equals the NULL sentinel: skipped by the non-grouped skip_val rule
            assert int(buf[0, 0]) == want


@pytest.mark.skipif(__import__("oracle.oracle", fromlist=["x"]).ref_lib() is None,
                    reason="oracle/_ref not built")
def test_live_against_reference_functions(oracle):
    """Fuzz the restatement against the reference's own compiled functions."""
    ref = oracle.ref_lib()
    rng = np.random.default_rng(7)
    for _ in range(300):
        n = int(rng.integers(1, 33))
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        seed = int(rng.integers(0, 2**32))
        assert oracle.murmur3(b, seed) == ref.MurmurHash3(b, n, seed)
        assert oracle.murmur1(b, seed) == ref.MurmurHash1(b, n, seed)
    for kw, rq, n in [(8, 4, 61), (4, 2, 128), (8, 2, 7)]:
        ref_buf = np.zeros(n * rq, dtype=np.int64)
        for e in range(n):
            if kw == 8:
                ref_buf[e * rq] = EMPTY64
            else:
                ref_buf[e * rq:e * rq + 1].view(np.int32)[0] = EMPTY32
        my_buf = ref_buf.copy()
        for k in rng.integers(-50, 50, 400):
            k = int(k)
            kb = np.array([k], dtype=np.int64) if kw == 8 else np.array([k, 0], dtype=np.int32)
            p = ref.get_group_value(ref_buf.ctypes.data, n, kb.ctypes.data, 1, kw, rq)
            s = oracle.lib().orc_get_group_value_slot(my_buf.ctypes.data, n, k, kw, rq)
            if not p:
                assert s == -1
                continue
            quad = (p - ref_buf.ctypes.data) // 8
            assert quad == s
            ref_buf[quad] += 1
            my_buf[s] += 1
        assert (ref_buf == my_buf).all()


def test_probe_invariant_checker(oracle):
    """tests.helpers.check_probe_invariant (used by the GPU parity tests to prove a product
    buffer is a valid image of the reference's linear probing) accepts the oracle's own
    get_group_value table and rejects a table with a displaced key; its numpy MurmurHash3
    agrees with the oracle's (reference-pinned) one."""
    from heavydb_amd import capi
    from heavydb_amd.executor import (ExpressionRange, InputColDescriptor, RelAlgExecutionUnit,
                                      TargetExpr)
    from tests.helpers import check_probe_invariant, murmur3_u64
    rng = np.random.default_rng(5)
    keys = (rng.integers(0, 3000, 20000) * 1000003 + 7).astype(np.int64)
    for k in keys[:64]:
        assert int(murmur3_u64(np.array([k]))[0]) == oracle.murmur3(struct.pack("<q", int(k)))
    ra = RelAlgExecutionUnit(
        [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, 2999 * 1000003 + 7))],
        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT)], [], [0],
        max_groups_buffer_entry_guess=4200)  # ~71 % fill: long probe chains
    q, buf, code = oracle.execute(ra.to_plan(), [[keys]], n_threads=1)
    assert code == 0
    check_probe_invariant(q, buf)
    rq = q.row_size // 8
    rows = buf.reshape(-1, rq).copy()
    live = np.nonzero(rows[:, 0] != EMPTY64)[0]
    empty = np.nonzero(rows[:, 0] == EMPTY64)[0]
    # move one key far away from its probe chain: the reference's probe would not find it
    src = live[len(live) // 2]
    home = int(murmur3_u64(rows[src:src + 1, 0])[0] % rows.shape[0])
    dst = next(int(e) for e in empty if (int(e) - home) % rows.shape[0] > (int(empty[np.searchsorted(empty, home) % len(empty)]) - home) % rows.shape[0])
    rows[dst] = rows[src]
    rows[src, 0] = EMPTY64
    with pytest.raises(AssertionError):
        check_probe_invariant(q, rows.reshape(-1))


def test_rows_to_arrow(oracle):
    """The Arrow export of the host mirror (ArrowResultSetConverter analogue): values and NULLs of
    every target survive the conversion (fed here with the oracle's rows of a nullable case)."""
    from heavydb_amd.executor import rows_to_arrow
    from tests import cases as cases_mod
    case = next(c for c in cases_mod.build_cases() if c.name == "baseline_nullable_args")
    q, buf, code = oracle.execute(case.ra.to_plan(), case.frags)
    assert code == 0
    ival, dval, nul = oracle.fetch_rows(q, buf)
    tab = rows_to_arrow(q, ival, dval, nul, names=["key", "sum", "avg", "min", "cnt"])
    assert tab.num_rows == ival.shape[0] and tab.column_names == ["key", "sum", "avg", "min", "cnt"]
    for t, name in enumerate(tab.column_names):
        col = tab.column(name).to_pylist()
        for r in (0, 1, len(col) // 2, len(col) - 1):
            want = None if nul[r, t] else (float(dval[r, t]) if q.target_is_fp[t] else int(ival[r, t]))
            assert col[r] == want
    assert tab.column("sum").null_count == int(nul[:, 1].sum())


def test_columnar_runtime_functions_match_reference_vectors(oracle):
    """get_group_value_columnar_slot / get_columnar_group_bin_offset restated in oracle.cpp against
    traces produced by the reference's own functions (oracle/gen_golden_columnar.py ->
    tests/golden/ref_columnar_vectors.json), and against the reference live when oracle/_ref is here."""
    import ctypes as C
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_columnar_vectors.json")
    g = json.load(open(path))
    EMPTY = 2**63 - 1
    ref = oracle.ref_lib()
    if ref is not None:
        ref.get_group_value_columnar_slot.restype = C.c_int32
        ref.get_group_value_columnar_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        ref.get_columnar_group_bin_offset.restype = C.c_uint32
        ref.get_columnar_group_bin_offset.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    lib = oracle.lib()
    assert len(g["columnar_baseline_traces"]) >= 4
    for tr in g["columnar_baseline_traces"]:
        ec, kc = tr["entry_count"], tr["key_count"]
        buf = np.full(ec * kc, EMPTY, dtype=np.int64)
        live = np.full(ec * kc, EMPTY, dtype=np.int64)
        for k, want in zip(tr["keys"], tr["bins"]):
            kb = np.array(k, dtype=np.int64)
            assert lib.orc_get_group_value_columnar_slot(buf.ctypes.data, ec, kb.ctypes.data, kc) == want
            if ref is not None:
                assert ref.get_group_value_columnar_slot(live.ctypes.data, ec, kb.ctypes.data, kc, 8) == want
        assert buf.tolist() == tr["final_key_columns"]
    for tr in g["columnar_perfect_traces"]:
        col = np.full(tr["entries"], EMPTY, dtype=np.int64)
        for k, want in zip(tr["keys"], tr["bins"]):
            assert lib.orc_get_columnar_group_bin_offset(col.ctypes.data, k, tr["min_key"], tr["bucket"]) == want
        assert col.tolist() == tr["final_key_column"]
    # the same bins through a whole columnar step: a baseline GROUP BY over the first trace's keys
    from heavydb_amd import capi
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    tr = g["columnar_baseline_traces"][0]
    keys = np.array([k[0] for k in tr["keys"]], dtype=np.int64)
    ra = RelAlgExecutionUnit([InputColDescriptor(capi.INT64, False, ExpressionRange(False))], [TargetExpr(capi.COUNT)],
                             groupby_exprs=[0], max_groups_buffer_entry_guess=tr["entry_count"],
                             output_columnar_hint=capi.OUTPUT_COLUMNAR, bigint_count=True)
    q, buf, code = oracle.execute(ra.to_plan(), [[keys]])
    assert code == 0 and q.output_columnar and q.entry_count == 8
    assert buf[:8].tolist() == tr["final_key_columns"]
    counts = buf[8:16]
    for k, b in zip(tr["keys"], tr["bins"]):
        assert counts[b] == sum(1 for kk in tr["keys"] if kk == k)


# ---- aggregates and comparisons against vectors produced by the reference's own RuntimeFunctions.cpp
# (tests/golden/ref_agg_vectors.json, oracle/gen_golden_agg.py)
_TYPE = {"int8_t": ("INT8", np.int8), "int16_t": ("INT16", np.int16), "int32_t": ("INT32", np.int32),
         "int64_t": ("INT64", np.int64), "double": ("DOUBLE", np.float64), "float": ("FLOAT", np.float32)}


def _agg_vectors():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_agg_vectors.json")) as f:
        return json.load(f)


def agg_case_unit(case):
    """The step of one "agg" vector: SELECT COUNT(*), AGG(v) FROM t GROUP BY k with k = 0 in every row."""
    from heavydb_amd import capi
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    tname, npt = _TYPE[case["type"]]
    t = getattr(capi, tname)
    if case["values_are_double_bits"]:
        v = np.array(case["values"], dtype=np.int64).view(np.float64)
    elif case.get("values_are_float_bits"):
        v = np.array(case["values"], dtype=np.int32).view(np.float32)
    else:
        v = np.array(case["values"], dtype=npt)
    null = np.finfo(npt).tiny if t in (capi.DOUBLE, capi.FLOAT) else np.iinfo(npt).min
    nn = v[v != null] if case["nullable"] else v
    if t in (capi.DOUBLE, capi.FLOAT):
        rng_v = ExpressionRange(True, 0, 0, case["nullable"], float(nn.min()) if nn.size else 0.0, float(nn.max()) if nn.size else 0.0)
    else:
        rng_v = ExpressionRange(True, int(nn.min()) if nn.size else 0, int(nn.max()) if nn.size else 0, case["nullable"])
    descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 0)), InputColDescriptor(t, case["nullable"], rng_v)]
    agg = {"count": capi.COUNT, "sum": capi.SUM, "min": capi.MIN, "max": capi.MAX}[case["agg"]]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT), TargetExpr(agg, 1)], [], [0], bigint_count=True)
    return ra, [[np.zeros(len(v), dtype=np.int32), v]]


def cond_case_unit(case):
    """SELECT COUNT_IF(c < 3), SUM_IF(v, c < 3) FROM t GROUP BY k (k = 0 in every row)."""
    from heavydb_amd import capi
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    dbl = case["values_are_double_bits"]
    v = np.array(case["values"], dtype=np.int64)
    v = v.view(np.float64) if dbl else v
    c = np.array(case["cond_values"], dtype=np.int64)
    vnull = np.finfo(np.float64).tiny if dbl else -2**63
    nnv = v[v != vnull] if case["value_nullable"] else v
    nnc = c[c != -2**63] if case["cond_nullable"] else c
    rv = (ExpressionRange(True, 0, 0, case["value_nullable"], float(nnv.min()), float(nnv.max())) if dbl
          else ExpressionRange(True, int(nnv.min()), int(nnv.max()), case["value_nullable"]))
    descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 0)),
             InputColDescriptor(capi.DOUBLE if dbl else capi.INT64, case["value_nullable"], rv),
             InputColDescriptor(capi.INT64, case["cond_nullable"], ExpressionRange(True, int(nnc.min()), int(nnc.max()), case["cond_nullable"]))]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT_IF, cond=Qual(2, capi.LT, 3)),
                                     TargetExpr(capi.SUM_IF, 1, cond=Qual(2, capi.LT, 3))], [], [0], bigint_count=True)
    return ra, [[np.zeros(len(v), dtype=np.int32), v, c]]


def cmp_case_unit(case):
    from heavydb_amd import capi
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    tname, npt = _TYPE[case["type"]]
    t = getattr(capi, tname)
    v = (np.array(case["values"], dtype=np.int64).view(np.float64) if case["values_are_double_bits"]
         else np.array(case["values"], dtype=npt))
    rng_v = (ExpressionRange(True, 0, 0, case["nullable"], -10.0, 10.0) if t == capi.DOUBLE
             else ExpressionRange(True, -10, 10, case["nullable"]))
    ra = RelAlgExecutionUnit([InputColDescriptor(t, case["nullable"], rng_v)], [TargetExpr(capi.COUNT)],
                             [Qual(0, case["op_code"], case["literal"])], [])
    return ra, [[v]]


def test_aggregates_match_reference_functions(oracle):
    """Every aggregate x argument type x nullability: the slot the oracle's row function leaves equals what the
    reference's own agg_* function leaves (bit for bit, doubles included: same operations in the same order)."""
    vec = _agg_vectors()
    assert len(vec["agg"]) == 184 and len(vec["cond"]) == 8
    for case in vec["cond"]:   # COUNT_IF / SUM_IF: agg_count_if[_skip_val], agg_sum_if[_double][_skip_val]
        ra, frags = cond_case_unit(case)
        q, buf, code = oracle.execute(ra.to_plan(), frags, n_threads=1)
        assert code == 0
        rows = np.asarray(buf).view(np.int64).reshape(q.entry_count, -1)
        kq = q.key_bytes // 8
        for j in range(2):
            assert q.target_slot[j] == case["slots"][j] and int(q.init_vals[case["slots"][j]]) == case["init"][j]
            assert int(rows[0, kq + case["slots"][j]]) == case["want"][j], (case["ref_functions"][j], case)
    for case in vec["agg"]:
        ra, frags = agg_case_unit(case)
        q, buf, code = oracle.execute(ra.to_plan(), frags, n_threads=1)
        assert code == 0
        assert q.target_slot[1] == case["slot"] and int(q.init_vals[case["slot"]]) == case["init"], case["ref_function"]
        rows = np.asarray(buf).view(np.int64).reshape(q.entry_count, -1)
        kq = q.key_bytes // 8
        assert int(rows[0, kq + case["slot"]]) == case["want"], (case["ref_function"], case["type"], case["nullable"],
                                                                  case["values"][:6], int(rows[0, kq + case["slot"]]), case["want"])


def test_comparisons_match_reference_functions(oracle):
    vec = _agg_vectors()
    assert len(vec["cmp"]) == 60
    for case in vec["cmp"]:
        ra, frags = cmp_case_unit(case)
        q, buf, code = oracle.execute(ra.to_plan(), frags, n_threads=1)
        assert code == 0
        assert int(np.asarray(buf).view(np.int64).reshape(-1)[0]) == case["want_count"], (case["ref_function"], case["want_count"])


def test_multi_column_perfect_hash_matches_reference_function(oracle):
    """get_matching_group_value_perfect_hash of the reference (vectors in ref_agg_vectors.json) vs the oracle's."""
    vec = _agg_vectors()
    assert len(vec["perfect_multi"]) == 3
    lib = oracle.lib()
    for tr in vec["perfect_multi"]:
        kc, rq, n = tr["key_count"], tr["row_size_quad"], tr["entry_count"]
        buf = np.full(n * rq, 2**63 - 1, dtype=np.int64)
        for e in range(n):
            buf[e * rq + kc:(e + 1) * rq] = 0
        for call in tr["calls"]:
            key = np.array(call["key"], dtype=np.int64)
            off = lib.orc_perfect_hash_slot(buf.ctypes.data, call["hashed_index"], key.ctypes.data, kc, rq)
            assert off == call["returned_quad_offset"], call
        assert buf.tolist() == tr["final_buffer"]


def test_projection_runtime_matches_reference_vectors(oracle):
    """get_scan_output_slot / get_columnar_scan_output_offset (GroupByRuntime.cpp:242-269) restated in oracle.cpp, and the
    projection step built on them (run_fragment's Projection branch), against traces the reference's own functions produced
    (oracle/gen_golden_projection.py -> tests/golden/ref_projection_vectors.json): the slot / offset every call returns — -1
    once the buffer is full — and the final buffer image, row-wise (8-byte slots, agg_id / agg_id_double) and columnar
    (logical-width columns, agg_id_int8 / 16 / 32 / agg_id_float)."""
    import ctypes as C
    import json
    import os
    from heavydb_amd import capi
    from heavydb_amd.executor import InputColDescriptor as D, RelAlgExecutionUnit, TargetExpr
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_projection_vectors.json")))
    lib = oracle.lib()
    lib.orc_get_scan_output_slot.restype = C.c_int64
    lib.orc_get_scan_output_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int64, C.c_uint32]
    lib.orc_get_columnar_scan_output_offset.restype = C.c_int32
    lib.orc_get_columnar_scan_output_offset.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int64]
    EMPTY = 2**63 - 1
    assert len(g["rowwise"]) >= 4 and len(g["columnar"]) >= 3
    for tr in g["rowwise"]:
        ec, rq = tr["entry_count"], tr["row_size_quad"]
        buf = np.zeros(ec * rq, dtype=np.int64)
        buf.reshape(ec, rq)[:, 0] = EMPTY
        for c in tr["calls"]:
            got = lib.orc_get_scan_output_slot(buf.ctypes.data, ec, c["old_total_matched"], c["offset_in_fragment"], rq)
            assert got == c["slot_quad"]
            if got >= 0:
                for s in range(rq - 1):
                    buf[got + s] = np.array([c["dval"]], dtype=np.float64).view(np.int64)[0] if s == 1 else c["ivals"][s]
        assert buf.tolist() == tr["final"]
    for tr in g["columnar"]:
        ec = tr["entry_count"]
        keys = np.full(ec, EMPTY, dtype=np.int64)
        for c in tr["calls"]:
            assert lib.orc_get_columnar_scan_output_offset(keys.ctypes.data, ec, c["old_total_matched"], c["offset_in_fragment"]) == c["offset"]
        assert keys.tolist() == tr["keys"]
        # the same trace as a projection STEP of the oracle: the input columns hold exactly the traced values, every row passes
        n = len(tr["calls"])
        cols = [np.array([c["v8"] for c in tr["calls"]], dtype=np.int8), np.array([c["v16"] for c in tr["calls"]], dtype=np.int16),
                np.array([c["v32"] for c in tr["calls"]], dtype=np.int32), np.array([c["vf"] for c in tr["calls"]], dtype=np.float32)]
        ra = RelAlgExecutionUnit([D(capi.INT8), D(capi.INT16), D(capi.INT32), D(capi.FLOAT)], [TargetExpr(capi.PROJECT, i) for i in range(4)],
                                 scan_limit=ec, output_columnar_hint=capi.OUTPUT_COLUMNAR)
        q, buf, code = oracle.execute(ra.to_plan(), [cols])
        assert code == 0 and q.entry_count == ec
        raw = buf.view(np.int8)
        live = min(n, ec)
        # (the step's key is the row's offset in the fragment = its index here, not the traced random offset)
        assert raw[:8 * ec].view(np.int64).tolist() == list(range(live)) + [EMPTY] * (ec - live)
        for s, (name, dt) in enumerate((("c8", np.int8), ("c16", np.int16), ("c32", np.int32), ("cf_bits", np.int32))):
            o = oracle.col_slot_off(q, s)
            w = np.dtype(dt).itemsize
            assert raw[o:o + w * live].view(dt).tolist() == tr[name][:live], name

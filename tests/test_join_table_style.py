"""Ports of the reference's own join hash table tests (Tests/JoinHashTableTest.cpp Build.*:
PerfectOneToOne1/2 :180-284, PerfectOneToMany1/2 :286-378, KeyedOneToOne :481-537, KeyedOneToMany
:539-597): the same inner key columns, the decoded sets the tests expect (HashTable::toSet) and —
where the test's comment prints it — the physical buffer (slot positions from MurmurHash1 over the
4-byte key components, offsets | counts | payloads), against the oracle's join builds."""
import numpy as np

from heavydb_amd import capi
from tests.helpers import decode_join_table

E32 = 2**31 - 1


def _decoded(oj, min_key=0):
    info, sh = oj.info(), oj.shape()
    return decode_join_table(oj.raw(), info["hash_type"], info["entry_count"], sh["key_components"],
                             sh["component_width"], min_key)


def test_perfect_one_to_one(oracle):
    # | perfect one-to-one | payloads 0 1 2 3 4 5 6 7 8 9 |
    oj = oracle.OracleJoin(np.arange(10, dtype=np.int32), capi.INT32, 0, 9)
    assert oj.info() == dict(hash_type=0, entry_count=10)
    assert _decoded(oj) == {(i,): [i] for i in range(10)}
    assert oj.raw().view(np.int32)[:10].tolist() == list(range(10))
    # | perfect one-to-one | payloads 0 1 2 * 3 4 5 6 * 7 |
    keys = np.array([0, 1, 2, 4, 5, 6, 7, 9], dtype=np.int32)
    oj = oracle.OracleJoin(keys, capi.INT32, 0, 9)
    assert _decoded(oj) == {(0,): [0], (1,): [1], (2,): [2], (4,): [3], (5,): [4], (6,): [5], (7,): [6], (9,): [7]}
    assert oj.raw().view(np.int32)[:10].tolist() == [0, 1, 2, -1, 3, 4, 5, 6, -1, 7]


def test_perfect_one_to_many(oracle):
    # | perfect one-to-many | offsets 0 2 4 6 8 | counts 2 2 2 2 2 | payloads 0 5 1 6 2 7 3 8 4 9 |
    keys = np.array([0, 1, 2, 3, 4, 0, 1, 2, 3, 4], dtype=np.int32)
    oj = oracle.OracleJoin(keys, capi.INT32, 0, 4, one_to_many=1)
    assert oj.info() == dict(hash_type=2, entry_count=5)
    assert _decoded(oj) == {(0,): [0, 5], (1,): [1, 6], (2,): [2, 7], (3,): [3, 8], (4,): [4, 9]}
    assert oj.raw().view(np.int32).tolist() == [0, 2, 4, 6, 8, 2, 2, 2, 2, 2, 0, 5, 1, 6, 2, 7, 3, 8, 4, 9]
    # | perfect one-to-many | offsets 0 * 2 4 6 | counts 2 * 2 2 2 | payloads 0 4 1 5 2 6 3 7 |
    keys = np.array([0, 2, 3, 4, 0, 2, 3, 4], dtype=np.int32)
    oj = oracle.OracleJoin(keys, capi.INT32, 0, 4, one_to_many=1)
    assert _decoded(oj) == {(0,): [0, 4], (2,): [1, 5], (3,): [2, 6], (4,): [3, 7]}
    assert oj.raw().view(np.int32).tolist() == [0, -1, 2, 4, 6, 2, 0, 2, 2, 2, 0, 4, 1, 5, 2, 6, 3, 7]
    # a unique key column asked for OneToOne first stays OneToOne (getHashType() == OneToOne above)
    assert oracle.OracleJoin(np.arange(5, dtype=np.int32), capi.INT32, 0, 4, one_to_many=1).info()["hash_type"] == 0


def test_keyed_one_to_one(oracle):
    # a1 = b and a2 = b: the inner key is (b, b), two 4-byte components, 2 x 3 rows = 6 slots
    # | keyed one-to-one | keys * (1,1,1) (3,3,2) (0,0,0) * * |
    b = np.array([0, 1, 3], dtype=np.int32)
    oj = oracle.OracleJoin([b, b], [capi.INT32, capi.INT32], 0, 0, keyed_entry_count=6)
    assert oj.info() == dict(hash_type=1, entry_count=6)
    assert oj.shape()["key_components"] == 2 and oj.shape()["component_width"] == 4
    assert _decoded(oj) == {(0, 0): [0], (1, 1): [1], (3, 3): [2]}
    tab = oj.raw()[:6 * 3 * 4].view(np.int32).reshape(6, 3)
    assert tab[1].tolist() == [1, 1, 1] and tab[2].tolist() == [3, 3, 2] and tab[3].tolist() == [0, 0, 0]
    assert (tab[[0, 4, 5], 0] == E32).all()


def test_keyed_one_to_many(oracle):
    # | keyed one-to-many | keys * (1,1) (3,3) (0,0) * * | offsets * 0 1 3 * * | counts * 1 2 1 * * |
    # | payloads 1 2 3 0 |
    b = np.array([0, 1, 3, 3], dtype=np.int32)
    oj = oracle.OracleJoin([b, b], [capi.INT32, capi.INT32], 0, 0, one_to_many=1, keyed_entry_count=6)
    assert oj.info() == dict(hash_type=3, entry_count=6)
    assert _decoded(oj) == {(0, 0): [0], (1, 1): [1], (3, 3): [2, 3]}
    raw = oj.raw()
    keys = raw[:6 * 2 * 4].view(np.int32).reshape(6, 2)
    rest = raw[6 * 2 * 4:].view(np.int32)
    assert keys[1].tolist() == [1, 1] and keys[2].tolist() == [3, 3] and keys[3].tolist() == [0, 0]
    assert (keys[[0, 4, 5], 0] == E32).all()
    offsets, counts, payloads = rest[:6], rest[6:12], rest[12:16]
    assert offsets.tolist() == [-1, 0, 1, 3, -1, -1] and counts.tolist() == [0, 1, 2, 1, 0, 0]
    assert payloads[0] == 1 and sorted(payloads[1:3].tolist()) == [2, 3] and payloads[3] == 0
    # probes: the matching sets the row function iterates
    assert oj.matches([3, 3]) == [2, 3] and oj.matches([1, 1]) == [1] and oj.matches([2, 2]) == []

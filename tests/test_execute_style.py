"""Ports of the reference's end-to-end query tests (Tests/ExecuteTest.cpp): ITS `test` table
(rows of the three INSERT statements at :30063-30115, 10 + 5 + 5 rows in fragments of 2 rows, numeric
columns only; `str` as its dictionary ids) and ITS query texts from Select.FilterAndSimpleAggregation
(:1888-1970), Select.GroupByPerfectHash (:11423-11480) and friends — the ones that are a single step
of scan / filter / group by / aggregate.  Exactly like the reference's `c(query, dt)`, the SQL text runs
on SQLite; the same query written as a RelAlgExecutionUnit runs through the oracle and through the
product's row logic (host emulation), and the three row sets must agree."""
import math
import sqlite3

import numpy as np
import pytest

from heavydb_amd import capi
from heavydb_amd.executor import Expr, ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
from tests.cases import Case, col_range
from tests.helpers import F32_ATOL, F32_RTOL

I8, I16, I32, I64, F32, F64 = capi.INT8, capi.INT16, capi.INT32, capi.INT64, capi.FLOAT, capi.DOUBLE
NP = {I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64, F32: np.float32, F64: np.float64}
NULL = {I8: -2**7, I16: -2**15, I32: -2**31, I64: -2**63, F32: np.finfo(np.float32).tiny, F64: np.finfo(np.float64).tiny}
M1, M2 = 1418509395, 1418595795     # '2014-12-13 22:23:15', '2014-12-14 22:23:15'
DAY = 936835200                     # '1999-09-09' in seconds; 10843 days

# name: (storage type, nullable, [row1, row2, row3], encoding, logical type)
COLS = {
    "x": (I32, False, [7, 8, 7]), "w": (I8, True, [-8, -7, -7]), "y": (I32, True, [42, 43, 43]),
    "z": (I16, True, [101, -78, 102]), "t": (I64, True, [1001, 1002, 1002]),
    "f": (F32, True, [1.1, 1.2, 1.3]), "ff": (F32, True, [1.1, 101.2, 1000.3]), "fn": (F32, True, [None, -101.2, -1000.3]),
    "d": (F64, True, [2.2, 2.4, 2.6]), "dn": (F64, True, [None, -2002.4, -220.6]),
    "str": (I32, False, [0, 1, 2], capi.ENC_DICT, 0),                       # 'foo', 'bar', 'baz'
    "m": (I64, True, [M1, M1, M2]),
    "o1": (I16, True, [DAY // 86400, None, DAY // 86400], capi.ENC_DATE_IN_DAYS, 0),  # date encoding fixed(16)
    "fx": (I16, True, [9, None, 11], capi.ENC_FIXED, I32),                  # int encoding fixed(16)
    "u": (I32, True, [None, None, None]), "ofd": (I32, True, [2147483647, None, 1]),
    "ufd": (I32, False, [-2147483648, -2147483647, -1]), "ofq": (I64, True, [None, 2**63 - 1, 1]),
    "ufq": (I64, False, [-1, -2**63, -2**63]), "smallint_nulls": (I16, True, [32767, None, 1]),
    "b": (I8, True, [1, 0, None]), "bn": (I8, False, [1, 0, 1]),      # BOOLEAN: 't', 'f', null / 't', 'f', 't' (NOT NULL)
}
NAMES = list(COLS)
REPEAT = [10, 5, 5]   # g_num_rows = 10


def _table():
    arrays, descs, sql_cols = [], [], []
    for name in NAMES:
        spec = COLS[name]
        t, nullable, vals = spec[:3]
        enc, logical = (spec[3], spec[4]) if len(spec) > 3 else (0, 0)
        stored = np.array([NULL[t] if v is None else v for v, r in zip(vals, REPEAT) for _ in range(r)], dtype=NP[t])
        logical_vals = [None if v is None else (v * 86400 if enc == capi.ENC_DATE_IN_DAYS else
                                               float(np.float32(v)) if t == F32 else v)
                        for v, r in zip(vals, REPEAT) for _ in range(r)]
        arrays.append(stored)
        sql_cols.append(logical_vals)
        if enc == capi.ENC_DATE_IN_DAYS:
            rng = col_range([stored.astype(np.int64) * 86400], I64, False)
            nn = [v for v in logical_vals if v is not None]
            from heavydb_amd.executor import ExpressionRange
            rng = ExpressionRange(True, min(nn), max(nn), any(v is None for v in logical_vals), bucket=86400)
        else:
            rng = col_range([stored], t, nullable)
        descs.append(InputColDescriptor(t, nullable, rng, enc, logical))
    n = sum(REPEAT)
    frags = [[a[i:i + 2] for a in arrays] for i in range(0, n, 2)]  # fragment size 2 (:30049)
    db = sqlite3.connect(":memory:")
    db.execute("CREATE TABLE test (" + ", ".join(NAMES) + ")")
    db.executemany("INSERT INTO test VALUES (" + ",".join("?" * len(NAMES)) + ")", list(zip(*sql_cols)))
    return descs, frags, db


A = {"COUNT": capi.COUNT, "SUM": capi.SUM, "AVG": capi.AVG, "MIN": capi.MIN, "MAX": capi.MAX}
OPS = {"<": capi.LT, ">": capi.GT, "=": capi.EQ, "<>": capi.NE, "<=": capi.LE, ">=": capi.GE,
       "IS NULL": capi.IS_NULL, "IS NOT NULL": capi.IS_NOT_NULL}


# column NAMES here; resolved against the query's own input_col_descs (the reference fetches only the
# columns a query uses, RelAlgExecutionUnit::input_col_descs)
def agg(fn, col=None):
    return ("agg", A[fn], col)


def q(col, op, lit):
    return (col, OPS[op], lit)


def key(i=0):
    return ("key", i, None)


def _unit(descs, frags, targets, quals, group, **kw):
    used = []
    for n in list(group) + [t[2] for t in targets if t[0] == "agg" and t[2]] + [c for c, _, _ in quals]:
        if n not in used:
            used.append(n)
    if not used:
        used = ["x"]
    idx = {n: i for i, n in enumerate(used)}
    src = [NAMES.index(n) for n in used]
    tx = [TargetExpr(capi.PROJECT_KEY, t[1]) if t[0] == "key" else TargetExpr(t[1], -1 if t[2] is None else idx[t[2]])
          for t in targets]
    ra = RelAlgExecutionUnit([descs[i] for i in src], tx, [Qual(idx[c], op, lit) for c, op, lit in quals],
                             [idx[g] for g in group], **kw)
    return ra, [[f[i] for i in src] for f in frags]


# (the reference's SQL text, targets, quals, group-by columns in GROUP BY order)
QUERIES = [
    ("SELECT COUNT(*) FROM test;", [agg("COUNT")], [], []),
    ("SELECT COUNT(f) FROM test;", [agg("COUNT", "f")], [], []),
    ("SELECT COUNT(smallint_nulls), COUNT(*), COUNT(fn) FROM test;",
     [agg("COUNT", "smallint_nulls"), agg("COUNT"), agg("COUNT", "fn")], [], []),
    ("SELECT MIN(x) FROM test;", [agg("MIN", "x")], [], []),
    ("SELECT MAX(x) FROM test;", [agg("MAX", "x")], [], []),
    ("SELECT MIN(z) FROM test;", [agg("MIN", "z")], [], []),
    ("SELECT MAX(z) FROM test;", [agg("MAX", "z")], [], []),
    ("SELECT MIN(t) FROM test;", [agg("MIN", "t")], [], []),
    ("SELECT MAX(t) FROM test;", [agg("MAX", "t")], [], []),
    ("SELECT MIN(ff) FROM test;", [agg("MIN", "ff")], [], []),
    ("SELECT MIN(fn) FROM test;", [agg("MIN", "fn")], [], []),
    ("SELECT SUM(ff) FROM test;", [agg("SUM", "ff")], [], []),
    ("SELECT SUM(fn) FROM test;", [agg("SUM", "fn")], [], []),
    ("SELECT SUM(d) FROM test;", [agg("SUM", "d")], [], []),
    ("SELECT SUM(dn) FROM test;", [agg("SUM", "dn")], [], []),
    ("SELECT COUNT(*) FROM test WHERE x > 6 AND x < 8;", [agg("COUNT")], [q("x", ">", 6), q("x", "<", 8)], []),
    ("SELECT COUNT(*) FROM test WHERE x > 6 AND x < 8 AND z > 100 AND z < 102;", [agg("COUNT")],
     [q("x", ">", 6), q("x", "<", 8), q("z", ">", 100), q("z", "<", 102)], []),
    ("SELECT COUNT(*) FROM test WHERE x <> 7;", [agg("COUNT")], [q("x", "<>", 7)], []),
    ("SELECT COUNT(*) FROM test WHERE z <> 102;", [agg("COUNT")], [q("z", "<>", 102)], []),
    ("SELECT COUNT(*) FROM test WHERE t <> 1002;", [agg("COUNT")], [q("t", "<>", 1002)], []),
    ("SELECT MIN(x) FROM test WHERE x = 7;", [agg("MIN", "x")], [q("x", "=", 7)], []),
    ("SELECT MIN(z) FROM test WHERE z = 101;", [agg("MIN", "z")], [q("z", "=", 101)], []),
    ("SELECT MIN(t) FROM test WHERE t = 1001;", [agg("MIN", "t")], [q("t", "=", 1001)], []),
    ("SELECT AVG(y) FROM test WHERE x > 6 AND x < 8;", [agg("AVG", "y")], [q("x", ">", 6), q("x", "<", 8)], []),
    ("SELECT AVG(y) FROM test WHERE z > 100 AND z < 102;", [agg("AVG", "y")], [q("z", ">", 100), q("z", "<", 102)], []),
    ("SELECT AVG(y) FROM test WHERE t > 1000 AND t < 1002;", [agg("AVG", "y")], [q("t", ">", 1000), q("t", "<", 1002)], []),
    ("SELECT COUNT(*) FROM test WHERE ff < 5.75;", [agg("COUNT")], [q("ff", "<", 5.75)], []),   # 23.0/4.0 folded
    ("SELECT SUM(ofd), MIN(ofd), MAX(ofd), COUNT(ofd), AVG(ofd) FROM test;",
     [agg("SUM", "ofd"), agg("MIN", "ofd"), agg("MAX", "ofd"), agg("COUNT", "ofd"), agg("AVG", "ofd")], [], []),
    ("SELECT COUNT(u), SUM(u), MIN(u), AVG(u) FROM test;", [agg("COUNT", "u"), agg("SUM", "u"), agg("MIN", "u"), agg("AVG", "u")], [], []),
    # (MIN(ufd) / MIN(ufq) are left out on purpose: those NOT NULL columns hold INT32_MIN / INT64_MIN,
    # and a non-grouped MIN is NULL-aware whatever the column says (set_notnull(target, false),
    # OutputBufferInitialization.cpp:281-286), so the reference skips exactly those values)
    ("SELECT MAX(ufd), MAX(ufq), MAX(ofq) FROM test;", [agg("MAX", "ufd"), agg("MAX", "ufq"), agg("MAX", "ofq")], [], []),
    ("SELECT MIN(fx), MAX(fx), SUM(fx), COUNT(fx) FROM test;", [agg("MIN", "fx"), agg("MAX", "fx"), agg("SUM", "fx"), agg("COUNT", "fx")], [], []),
    ("SELECT MIN(o1), MAX(o1), COUNT(o1) FROM test;", [agg("MIN", "o1"), agg("MAX", "o1"), agg("COUNT", "o1")], [], []),
    # Select.GroupByPerfectHash
    ("SELECT COUNT(*) FROM test GROUP BY x ORDER BY x DESC;", [agg("COUNT")], [], ["x"]),
    ("SELECT y, COUNT(*) FROM test GROUP BY y ORDER BY y DESC;", [key(), agg("COUNT")], [], ["y"]),
    ("SELECT str, COUNT(*) FROM test GROUP BY str ORDER BY str DESC;", [key(), agg("COUNT")], [], ["str"]),
    ("SELECT COUNT(*), z FROM test where x = 7 GROUP BY z ORDER BY z DESC;", [agg("COUNT"), key()], [q("x", "=", 7)], ["z"]),
    ("SELECT z as z0, z as z1, COUNT(*) FROM test GROUP BY z0, z1 ORDER BY z0 DESC;", [key(0), key(1), agg("COUNT")], [], ["z", "z"]),
    ("SELECT x, COUNT(y), SUM(y), AVG(y), MIN(y), MAX(y) FROM test GROUP BY x ORDER BY x DESC;",
     [key(), agg("COUNT", "y"), agg("SUM", "y"), agg("AVG", "y"), agg("MIN", "y"), agg("MAX", "y")], [], ["x"]),
    ("SELECT y, SUM(fn), AVG(ff), MAX(f) from test GROUP BY y ORDER BY y DESC;",
     [key(), agg("SUM", "fn"), agg("AVG", "ff"), agg("MAX", "f")], [], ["y"]),
    # multi-column perfect hash
    ("SELECT str, x FROM test GROUP BY x, str ORDER BY str, x;", [key(1), key(0)], [], ["x", "str"]),
    ("SELECT str, x, MAX(smallint_nulls), AVG(y), COUNT(dn) FROM test GROUP BY x, str ORDER BY str, x;",
     [key(1), key(0), agg("MAX", "smallint_nulls"), agg("AVG", "y"), agg("COUNT", "dn")], [], ["x", "str"]),
    ("SELECT str, x, MAX(smallint_nulls), COUNT(dn), COUNT(*) as cnt FROM test GROUP BY x, str ORDER BY cnt, str;",
     [key(1), key(0), agg("MAX", "smallint_nulls"), agg("COUNT", "dn"), agg("COUNT")], [], ["x", "str"]),
    ("SELECT x, str, z, SUM(dn), MAX(dn), AVG(dn) FROM test GROUP BY x, str, z ORDER BY str, z, x;",
     [key(0), key(1), key(2), agg("SUM", "dn"), agg("MAX", "dn"), agg("AVG", "dn")], [], ["x", "str", "z"]),
    ("SELECT x, SUM(dn), str, MAX(dn), z, AVG(dn), COUNT(*) FROM test GROUP BY z, x, str ORDER BY str, z, x;",
     [key(1), agg("SUM", "dn"), key(2), agg("MAX", "dn"), key(0), agg("AVG", "dn"), agg("COUNT")], [], ["z", "x", "str"]),
    ("SELECT x, AVG(ff) AS val FROM test GROUP BY x ORDER BY val;", [key(), agg("AVG", "ff")], [], ["x"]),          # :2022
    ("SELECT COUNT(*) FROM test WHERE d = 2.2", [agg("COUNT")], [q("d", "=", 2.2)], []),                                # :2034
    # :2041-2044 with the DATE literal '1999-09-08' written in seconds (SQLite holds the decoded seconds here)
    ("SELECT COUNT(*) FROM test WHERE o1 > 936748800;", [agg("COUNT")], [q("o1", ">", 936748800)], []),
    ("SELECT COUNT(*) FROM test WHERE o1 <= 936748800;", [agg("COUNT")], [q("o1", "<=", 936748800)], []),
    ("SELECT COUNT(*) FROM test WHERE o1 = 936748800;", [agg("COUNT")], [q("o1", "=", 936748800)], []),
    ("SELECT COUNT(*) FROM test WHERE o1 <> 936748800;", [agg("COUNT")], [q("o1", "<>", 936748800)], []),
    # more of Select.FilterAndSimpleAggregation and friends (found by scanning ExecuteTest.cpp for single-step texts)
    ("SELECT AVG(d) FROM test;", [agg("AVG", "d")], [], []),
    ("SELECT AVG(f) FROM test;", [agg("AVG", "f")], [], []),
    ("SELECT MIN(d) FROM test;", [agg("MIN", "d")], [], []),
    ("SELECT MAX(d) FROM test;", [agg("MAX", "d")], [], []),
    ("SELECT MIN(f) FROM test;", [agg("MIN", "f")], [], []),
    ("SELECT MAX(f) FROM test;", [agg("MAX", "f")], [], []),
    ("SELECT SUM(f) FROM test;", [agg("SUM", "f")], [], []),
    ("SELECT SUM(x), AVG(y) FROM test;", [agg("SUM", "x"), agg("AVG", "y")], [], []),
    ("SELECT COUNT(*) FROM test WHERE x > 8;", [agg("COUNT")], [q("x", ">", 8)], []),
    ("SELECT SUM(x) FROM test WHERE x > 8;", [agg("SUM", "x")], [q("x", ">", 8)], []),          # no row qualifies: NULL
    ("SELECT SUM(d) FROM test WHERE x > 8;", [agg("SUM", "d")], [q("x", ">", 8)], []),
    ("SELECT SUM(f) FROM test WHERE x > 8;", [agg("SUM", "f")], [q("x", ">", 8)], []),
    ("SELECT MIN(x) FROM test WHERE x <> 7 AND x <> 8;", [agg("MIN", "x")], [q("x", "<>", 7), q("x", "<>", 8)], []),
    ("SELECT MIN(x) FROM test WHERE z <> 101 AND z <> 102;", [agg("MIN", "x")], [q("z", "<>", 101), q("z", "<>", 102)], []),
    ("SELECT MIN(x) FROM test WHERE t <> 1001 AND t <> 1002;", [agg("MIN", "x")], [q("t", "<>", 1001), q("t", "<>", 1002)], []),
    ("SELECT COUNT(*) FROM test WHERE f > 1.0 AND f < 1.2;", [agg("COUNT")], [q("f", ">", 1.0), q("f", "<", 1.2)], []),
    ("SELECT COUNT(*) FROM test WHERE f > 1.101 AND f < 1.299;", [agg("COUNT")], [q("f", ">", 1.101), q("f", "<", 1.299)], []),
    ("SELECT COUNT(*) FROM test WHERE f > 1.201 AND f < 1.4;", [agg("COUNT")], [q("f", ">", 1.201), q("f", "<", 1.4)], []),
    ("SELECT COUNT(*) FROM test WHERE f > 1.0 AND f < 1.2 AND d > 2.0 AND d < 2.4;", [agg("COUNT")],
     [q("f", ">", 1.0), q("f", "<", 1.2), q("d", ">", 2.0), q("d", "<", 2.4)], []),
    ("SELECT x, COUNT(x) FROM test GROUP BY x;", [key(), agg("COUNT", "x")], [], ["x"]),
    ("SELECT x, y, COUNT(x) FROM test GROUP BY x,y;", [key(0), key(1), agg("COUNT", "x")], [], ["x", "y"]),
    ("SELECT X, COUNT(*) AS N FROM test GROUP BY teSt.x ORDER BY n DESC;", [key(), agg("COUNT")], [], ["x"]),
    # floating-point group keys (always the baseline layout; the key is the bit pattern of the value as a double)
    ("SELECT COUNT(*) AS n FROM test GROUP BY d ORDER BY n;", [agg("COUNT")], [], ["d"]),
    ("SELECT COUNT(*) AS n FROM test GROUP BY f ORDER BY n;", [agg("COUNT")], [], ["f"]),
    ("SELECT d, COUNT(*), SUM(x) FROM test GROUP BY d;", [key(), agg("COUNT"), agg("SUM", "x")], [], ["d"]),
    ("SELECT f, COUNT(*), MIN(z) FROM test GROUP BY f;", [key(), agg("COUNT"), agg("MIN", "z")], [], ["f"]),
    ("SELECT fn, COUNT(*) FROM test GROUP BY fn;", [key(), agg("COUNT")], [], ["fn"]),                 # NULL FLOAT key
    ("SELECT dn, AVG(y), COUNT(*) FROM test GROUP BY dn;", [key(), agg("AVG", "y"), agg("COUNT")], [], ["dn"]),   # NULL DOUBLE key
    ("SELECT x, dn, f, COUNT(*) FROM test GROUP BY x, dn, f;", [key(0), key(1), key(2), agg("COUNT")], [], ["x", "dn", "f"]),
    # Select.FilterAndMultipleAggregation (:2577, :2581), Select.FilterAndGroupBy (:2822, :2843-2847, :2862),
    # Select.GroupByPushDownFilterIntoExprRange (:5310-5319)
    ("SELECT AVG(x), AVG(y) FROM test;", [agg("AVG", "x"), agg("AVG", "y")], [], []),
    ("SELECT str, AVG(x), COUNT(*) as xx, COUNT(*) as countval FROM test GROUP BY str ORDER BY str;",
     [key(), agg("AVG", "x"), agg("COUNT"), agg("COUNT")], [], ["str"]),
    ("SELECT x, y, COUNT(*) FROM test GROUP BY x, y;", [key(0), key(1), agg("COUNT")], [], ["x", "y"]),
    ("SELECT x, AVG(u), COUNT(*) AS n FROM test GROUP BY x ORDER BY n DESC;", [key(), agg("AVG", "u"), agg("COUNT")], [], ["x"]),
    ("SELECT fx, COUNT(*) n FROM test GROUP BY fx ORDER BY n DESC, fx IS NULL DESC;", [key(), agg("COUNT")], [], ["fx"]),
    ("SELECT x, COUNT(*) AS n FROM test WHERE x > 7 GROUP BY x ORDER BY x", [key(), agg("COUNT")], [q("x", ">", 7)], ["x"]),
    ("SELECT y, COUNT(*) AS n FROM test WHERE y < 43 GROUP BY y ORDER BY n DESC", [key(), agg("COUNT")], [q("y", "<", 43)], ["y"]),
    ("SELECT z, COUNT(*) AS n FROM test WHERE z <= 43 AND y > 10 GROUP BY z ORDER BY n DESC", [key(), agg("COUNT")],
     [q("z", "<=", 43), q("y", ">", 10)], ["z"]),
    ("SELECT t, SUM(y) AS sum_y FROM test WHERE t < 2000 GROUP BY t ORDER BY t DESC", [key(), agg("SUM", "y")], [q("t", "<", 2000)], ["t"]),
    ("SELECT MAX(y) AS n FROM test WHERE x = 7 GROUP BY z ORDER BY n;", [agg("MAX", "y")], [q("x", "=", 7)], ["z"]),   # :3177 without HAVING MAX(x) > 5 (always true)
    # COUNT over NOT NULL columns that hold the inline NULL pattern, GROUPED: every row counts.  (The
    # non-grouped form is left out on purpose: there the reference makes every aggregate with an
    # argument NULL-aware — `target_info.skip_null_val = true`, TargetExprBuilder.cpp:684-690 — so its
    # COUNT(ufd) skips the rows holding INT32_MIN and departs from SQL; restated as is, see
    # test_non_grouped_aggregates_are_null_aware below.)
    ("SELECT x, COUNT(ufd), COUNT(ufq) FROM test GROUP BY x;", [key(), agg("COUNT", "ufd"), agg("COUNT", "ufq")], [], ["x"]),
    # a NOT NULL key column holding INT32_MIN (the inline NULL pattern) next to other keys (:2019, :2028, :5107)
    ("SELECT x, COUNT(*) AS n FROM test GROUP BY x, ufd ORDER BY x, n;", [key(0), agg("COUNT")], [], ["x", "ufd"]),
    ("SELECT COUNT(*) as val FROM test GROUP BY x, y, ufd ORDER BY val;", [agg("COUNT")], [], ["x", "y", "ufd"]),
    ("SELECT ufd, COUNT(*) n FROM test GROUP BY ufd, str ORDER BY ufd, n;", [key(0), agg("COUNT")], [], ["ufd", "str"]),
    # NULL group keys, bigint / date / fixed-encoded keys
    ("SELECT ofd, COUNT(*), SUM(x) FROM test GROUP BY ofd;", [key(), agg("COUNT"), agg("SUM", "x")], [], ["ofd"]),
    ("SELECT u, COUNT(*) FROM test GROUP BY u;", [key(), agg("COUNT")], [], ["u"]),
    ("SELECT smallint_nulls, MIN(w), MAX(t) FROM test GROUP BY smallint_nulls;", [key(), agg("MIN", "w"), agg("MAX", "t")], [], ["smallint_nulls"]),
    ("SELECT m, COUNT(*), AVG(d) FROM test GROUP BY m;", [key(), agg("COUNT"), agg("AVG", "d")], [], ["m"]),
    ("SELECT o1, COUNT(*) FROM test GROUP BY o1;", [key(), agg("COUNT")], [], ["o1"]),
    ("SELECT fx, COUNT(*), SUM(z) FROM test GROUP BY fx;", [key(), agg("COUNT"), agg("SUM", "z")], [], ["fx"]),
    ("SELECT t, w, COUNT(*), MIN(ff) FROM test GROUP BY t, w;", [key(0), key(1), agg("COUNT"), agg("MIN", "ff")], [], ["t", "w"]),
    # IS [NOT] NULL quals and constrained_not_null (a qual `arg IS NOT NULL` makes the grouped aggregate over
    # arg NOT NULL for the step: OutputBufferInitialization.cpp:287,301-324) — ExecuteTest.cpp:1955, :1989,
    # :2023-2026, :2830-2832 (without the always-true HAVING), :2868, :5183-5198, :5770-5771
    ("SELECT SUM(z) FROM test WHERE z IS NOT NULL;", [agg("SUM", "z")], [q("z", "IS NOT NULL", 0)], []),
    ("SELECT COUNT(*) FROM test WHERE u IS NOT NULL;", [agg("COUNT")], [q("u", "IS NOT NULL", 0)], []),
    ("SELECT x, MAX(fn) as val FROM test WHERE fn IS NOT NULL GROUP BY x ORDER BY val;",
     [key(), agg("MAX", "fn")], [q("fn", "IS NOT NULL", 0)], ["x"]),
    ("SELECT MAX(dn) FROM test WHERE dn IS NOT NULL;", [agg("MAX", "dn")], [q("dn", "IS NOT NULL", 0)], []),
    ("SELECT x, MAX(dn) as val FROM test WHERE dn IS NOT NULL GROUP BY x ORDER BY val;",
     [key(), agg("MAX", "dn")], [q("dn", "IS NOT NULL", 0)], ["x"]),
    ("SELECT str, MIN(y) FROM test WHERE y IS NOT NULL GROUP BY str ORDER BY str DESC;",
     [key(), agg("MIN", "y")], [q("y", "IS NOT NULL", 0)], ["str"]),
    ("SELECT x, MAX(z) FROM test WHERE z IS NOT NULL GROUP BY x;", [key(), agg("MAX", "z")], [q("z", "IS NOT NULL", 0)], ["x"]),
    ("SELECT x, SUM(z) FROM test WHERE z IS NOT NULL GROUP BY x ORDER BY x;", [key(), agg("SUM", "z")],
     [q("z", "IS NOT NULL", 0)], ["x"]),
    ("SELECT d, MAX(f) FROM test WHERE f IS NOT NULL GROUP BY d;", [key(), agg("MAX", "f")], [q("f", "IS NOT NULL", 0)], ["d"]),
    ("SELECT d, AVG(f) FROM test WHERE f IS NOT NULL GROUP BY d;", [key(), agg("AVG", "f")], [q("f", "IS NOT NULL", 0)], ["d"]),
    ("SELECT d, SUM(f) FROM test WHERE f IS NOT NULL GROUP BY d;", [key(), agg("SUM", "f")], [q("f", "IS NOT NULL", 0)], ["d"]),
    ("SELECT x, y, SUM(f) FROM test WHERE f IS NOT NULL GROUP BY x, y;", [key(0), key(1), agg("SUM", "f")],
     [q("f", "IS NOT NULL", 0)], ["x", "y"]),
    ("SELECT COUNT(*) FROM test WHERE str IS NULL;", [agg("COUNT")], [q("str", "IS NULL", 0)], []),
    ("SELECT COUNT(*) FROM test WHERE str IS NOT NULL;", [agg("COUNT")], [q("str", "IS NOT NULL", 0)], []),
    ("SELECT str, COUNT(*) FROM test where str IS NOT NULL GROUP BY str ORDER BY str;", [key(), agg("COUNT")],
     [q("str", "IS NOT NULL", 0)], ["str"]),
    ("SELECT x, SUM(fn), COUNT(fn), MIN(dn) FROM test WHERE fn IS NOT NULL GROUP BY x;",
     [key(), agg("SUM", "fn"), agg("COUNT", "fn"), agg("MIN", "dn")], [q("fn", "IS NOT NULL", 0)], ["x"]),
    ("SELECT x, COUNT(*) FROM test WHERE dn IS NULL GROUP BY x;", [key(), agg("COUNT")], [q("dn", "IS NULL", 0)], ["x"]),
]

# The layout the reference's own comments name for a query ("single-column perfect hash", "all these key
# columns are small ranged to force perfect hash", "multi-column perfect hash": ExecuteTest.cpp:11414-11470;
# floating-point keys always take the baseline layout, GroupByAndAggregate.cpp:199-207): reference-held
# intent the layout decisions (plan.cpp / oracle qmd_init) are checked against.
INTENDED_LAYOUT = {sql: capi.GROUP_BY_PERFECT_HASH for sql in [
    "SELECT COUNT(*) FROM test GROUP BY x ORDER BY x DESC;",
    "SELECT y, COUNT(*) FROM test GROUP BY y ORDER BY y DESC;",
    "SELECT str, COUNT(*) FROM test GROUP BY str ORDER BY str DESC;",
    "SELECT COUNT(*), z FROM test where x = 7 GROUP BY z ORDER BY z DESC;",
    "SELECT z as z0, z as z1, COUNT(*) FROM test GROUP BY z0, z1 ORDER BY z0 DESC;",
    "SELECT x, COUNT(y), SUM(y), AVG(y), MIN(y), MAX(y) FROM test GROUP BY x ORDER BY x DESC;",
    "SELECT y, SUM(fn), AVG(ff), MAX(f) from test GROUP BY y ORDER BY y DESC;",
    "SELECT str, x FROM test GROUP BY x, str ORDER BY str, x;",
    "SELECT str, x, MAX(smallint_nulls), AVG(y), COUNT(dn) FROM test GROUP BY x, str ORDER BY str, x;",
    "SELECT str, x, MAX(smallint_nulls), COUNT(dn), COUNT(*) as cnt FROM test GROUP BY x, str ORDER BY cnt, str;",
    "SELECT x, str, z, SUM(dn), MAX(dn), AVG(dn) FROM test GROUP BY x, str, z ORDER BY str, z, x;",
    "SELECT x, SUM(dn), str, MAX(dn), z, AVG(dn), COUNT(*) FROM test GROUP BY z, x, str ORDER BY str, z, x;",
]}
INTENDED_LAYOUT.update({sql: capi.GROUP_BY_BASELINE_HASH for sql in [
    "SELECT COUNT(*) AS n FROM test GROUP BY d ORDER BY n;",
    "SELECT COUNT(*) AS n FROM test GROUP BY f ORDER BY n;",
    "SELECT d, COUNT(*), SUM(x) FROM test GROUP BY d;",
    "SELECT f, COUNT(*), MIN(z) FROM test GROUP BY f;",
    "SELECT d, MAX(f) FROM test WHERE f IS NOT NULL GROUP BY d;",
]})


def _rows(fetch, qmd):
    iv, dv, nu = fetch
    out = []
    for r in range(iv.shape[0]):
        out.append(tuple(None if nu[r, t] else (float(dv[r, t]) if qmd.target_is_fp[t] else int(iv[r, t]))
                         for t in range(qmd.n_targets)))
    return out


def _key(row):
    return tuple((0, 0) if v is None else (1, v) for v in row if not isinstance(v, float)), \
        tuple(v for v in row if isinstance(v, float))


@pytest.mark.parametrize("bigint_count", [False, True], ids=["int_count", "bigint_count"])
@pytest.mark.parametrize("qi", range(len(QUERIES)), ids=[s[0][7:60].replace(" ", "_") for s in QUERIES])
def test_reference_queries(oracle, qi, bigint_count):
    from tests.test_rowlogic_emu import _emu_execute
    sql, targets, quals, group = QUERIES[qi]
    descs, frags, db = _table()
    ra, frags = _unit(descs, frags, targets, quals, group, bigint_count=bigint_count, num_tuples=sum(REPEAT))
    plan = ra.to_plan()
    qm, buf, code = oracle.execute(plan, frags, n_threads=3)
    assert code == 0
    if sql in INTENDED_LAYOUT:
        assert qm.desc_type == INTENDED_LAYOUT[sql], (sql, qm.desc_type)
    fp = [bool(qm.target_is_fp[t]) for t in range(qm.n_targets)]
    want = sorted((tuple(float(v) if f and v is not None else v for v, f in zip(r, fp))
                   for r in db.execute(sql).fetchall()), key=_key)
    got = sorted(_rows(oracle.fetch_rows(qm, buf), qm), key=_key)
    eq, ebuf, ecode = _emu_execute(Case("ref", ra, frags), plan, None)
    assert ecode == 0
    got_emu = sorted(_rows(oracle.fetch_rows(eq, ebuf), eq), key=_key)
    for name, rows in (("oracle", got), ("product row logic", got_emu)):
        assert len(rows) == len(want), (name, sql, want, rows)
        for w, g in zip(want, rows):
            for t, (a, b) in enumerate(zip(w, g)):
                if a is None or b is None:
                    assert a is None and b is None, (name, sql, t, w, g)
                elif fp[t]:
                    rt, at = (F32_RTOL, F32_ATOL) if qm.target_arg_is_f32[t] else (1e-12, 0.0)
                    assert math.isclose(a, b, rel_tol=rt, abs_tol=at), (name, sql, t, w, g)
                else:
                    assert a == b, (name, sql, t, w, g)


# ---- the queries of Select.FilterAndSimpleAggregation that carry EXPRESSIONS (Tests/ExecuteTest.cpp:1901-2035): arithmetic over
# several columns in aggregates and filters, narrowing casts, unary minus, AND inside OR, a division guarded by another
# conjunct, IS NULL of an expression.  The plan is written the way the analyzer types the tree (the narrower operand is cast
# to the common type, a literal takes the column's type) and the binding translates it (integration/Mi355qTranslate.h): a
# filter without a qual shape is a BOOLEAN expression `= 1`, programs beyond 12 nodes are split over expressions.
class EX:
    """an expression over column NAMES; built against the query's own column order (and the index of its first expression)"""
    def __init__(self, names, build):
        self.names, self.build = names, build


def xc(name):
    return EX([name], lambda ix, nc: Expr.col(ix[name]))


def xl(t, v):
    return EX([], lambda ix, nc: Expr.lit(t, v))


def xref(k):   # the value of the plan's earlier expression k
    return EX([], lambda ix, nc: Expr.col(nc + k))


def _x2(method):
    return lambda a, b, t: EX(a.names + b.names, lambda ix, nc: getattr(a.build(ix, nc), method)(b.build(ix, nc), t))


xadd, xsub, xmul, xdiv = _x2("add"), _x2("sub"), _x2("mul"), _x2("div")


def xcast(a, t):
    return EX(a.names, lambda ix, nc: a.build(ix, nc).cast(t))


def xneg(a, t):
    return EX(a.names, lambda ix, nc: a.build(ix, nc).neg(t))


def xcmp(a, op, b):
    code = {"<": capi.EX_LT, ">": capi.EX_GT, "=": capi.EX_EQ, "<>": capi.EX_NE, "<=": capi.EX_LE, ">=": capi.EX_GE}[op]
    return EX(a.names + b.names, lambda ix, nc: a.build(ix, nc).cmp(code, b.build(ix, nc)))


def xlogic(a, op, b, short_circuit=False):
    code = {"AND": capi.EX_AND, "OR": capi.EX_OR}[op]
    return EX(a.names + b.names, lambda ix, nc: a.build(ix, nc).logical(code, b.build(ix, nc), short_circuit))


def xband(name, t, lo, hi):   # name > lo AND name < hi
    return xlogic(xcmp(xc(name), ">", xl(t, lo)), "AND", xcmp(xc(name), "<", xl(t, hi)))


def xbetween(name, t, lo, hi):   # name BETWEEN lo AND hi
    return xlogic(xcmp(xc(name), ">=", xl(t, lo)), "AND", xcmp(xc(name), "<=", xl(t, hi)))


def xcase(cond, then, otherwise, t):
    return EX(cond.names + then.names + otherwise.names,
              lambda ix, nc: Expr.case(cond.build(ix, nc), then.build(ix, nc), otherwise.build(ix, nc), t))


def xnull(t):
    return EX([], lambda ix, nc: Expr.null(t))


def xin(name, t, values):   # name IN (values): the OR of the comparisons
    e = xcmp(xc(name), "=", xl(t, values[0]))
    for v in values[1:]:
        e = xlogic(e, "OR", xcmp(xc(name), "=", xl(t, v)))
    return e


def xnot(a):
    return EX(a.names, lambda ix, nc: a.build(ix, nc).logical_not())


def _unit_x(descs, frags, targets, quals, group, exprs, **kw):
    """like _unit; a target's / qual's / group's column may be ("x", k): expression k of `exprs`"""
    from tests.cases import expr_range
    is_x = lambda c: isinstance(c, tuple) and c[0] == "x"
    used = []
    for n in [g for g in group if not is_x(g)] + [t[2] for t in targets if t[0] == "agg" and t[2] and not is_x(t[2])] + \
            [c for c, _, _ in quals if not is_x(c)] + [n for e in exprs for n in e.names]:
        if n not in used:
            used.append(n)
    idx = {n: i for i, n in enumerate(used)}
    src = [NAMES.index(n) for n in used]
    nc = len(used)
    d2 = [descs[i] for i in src]
    f2 = [[f[i] for i in src] for f in frags]
    col = lambda c: nc + c[1] if is_x(c) else idx[c]
    built = []
    for e in exprs:
        b = e.build(idx, nc)
        built.append(b.with_range(expr_range(b, d2, f2, built[:])))
    tx = [TargetExpr(capi.PROJECT_KEY, t[1]) if t[0] == "key" else TargetExpr(t[1], -1 if t[2] is None else col(t[2])) for t in targets]
    ra = RelAlgExecutionUnit(d2, tx, [Qual(col(c), op, lit) for c, op, lit in quals], [col(g) for g in group], exprs=built, **kw)
    return ra, f2


X0, X1, X2, X3 = ("x", 0), ("x", 1), ("x", 2), ("x", 3)
_xy = xadd(xc("x"), xc("y"), I32)                                  # x + y
_xyz = xadd(_xy, xcast(xc("z"), I32), I32)                         # x + y + z
_xyzt = xadd(xcast(_xyz, I64), xc("t"), I64)                       # x + y + z + t
_xmy = xsub(xc("x"), xc("y"), I32)                                 # x - y
_xy15 = xadd(xmul(xc("x"), xc("y"), I32), xl(I32, 15), I32)        # x * y + 15
T8 = lambda n: xcast(xc(n), I8)
# (SQL text of the reference, targets, quals, group, expressions)
EXPR_QUERIES = [
    ("SELECT SUM(x + y) FROM test;", [agg("SUM", X0)], [], [], [_xy]),
    ("SELECT SUM(x + y + z) FROM test;", [agg("SUM", X0)], [], [], [_xyz]),
    ("SELECT SUM(x + y + z + t) FROM test;", [agg("SUM", X0)], [], [], [_xyzt]),
    ("SELECT COUNT(*) FROM test WHERE x > 6 AND x < 8 OR (z > 100 AND z < 103);", [agg("COUNT")], [q(X1, "=", 1)], [],
     [xband("z", I16, 100, 103), xlogic(xband("x", I32, 6, 8), "OR", xref(0))]),
    ("SELECT COUNT(*) FROM test WHERE x > 6 AND x < 8 OR (z > 100 AND z < 102) OR (t > 1000 AND t < 1002);", [agg("COUNT")],
     [q(X2, "=", 1)], [],
     [xband("z", I16, 100, 102), xband("t", I64, 1000, 1002), xlogic(xlogic(xband("x", I32, 6, 8), "OR", xref(0)), "OR", xref(1))]),
    ("SELECT COUNT(*) FROM test WHERE x + y = 49;", [agg("COUNT")], [q(X0, "=", 49)], [], [_xy]),
    ("SELECT COUNT(*) FROM test WHERE x + y + z = 150;", [agg("COUNT")], [q(X0, "=", 150)], [], [_xyz]),
    ("SELECT COUNT(*) FROM test WHERE x + y + z + t = 1151;", [agg("COUNT")], [q(X0, "=", 1151)], [], [_xyzt]),
    ("SELECT COUNT(*) FROM test WHERE CAST(x as TINYINT) + CAST(y as TINYINT) < CAST(z as TINYINT);", [agg("COUNT")],
     [q(X0, "=", 1)], [], [xcmp(xadd(T8("x"), T8("y"), I8), "<", T8("z"))]),
    ("SELECT COUNT(*) FROM test WHERE CAST(y as TINYINT) / CAST(x as TINYINT) = 6", [agg("COUNT")], [q(X0, "=", 6)], [],
     [xdiv(T8("y"), T8("x"), I8)]),
    ("SELECT SUM(x + y) FROM test WHERE x + y = 49;", [agg("SUM", X0)], [q(X0, "=", 49)], [], [_xy]),
    ("SELECT SUM(x + y + z) FROM test WHERE x + y = 49;", [agg("SUM", X1)], [q(X0, "=", 49)], [], [_xy, _xyz]),
    ("SELECT SUM(x + y + z + t) FROM test WHERE x + y = 49;", [agg("SUM", X1)], [q(X0, "=", 49)], [], [_xy, _xyzt]),
    ("SELECT COUNT(*) FROM test WHERE x - y = -35;", [agg("COUNT")], [q(X0, "=", -35)], [], [_xmy]),
    ("SELECT COUNT(*) FROM test WHERE x - y + z = 66;", [agg("COUNT")], [q(X0, "=", 66)], [], [xadd(_xmy, xcast(xc("z"), I32), I32)]),
    ("SELECT COUNT(*) FROM test WHERE x - y + z + t = 1067;", [agg("COUNT")], [q(X0, "=", 1067)], [],
     [xadd(xcast(xadd(_xmy, xcast(xc("z"), I32), I32), I64), xc("t"), I64)]),
    ("SELECT COUNT(*) FROM test WHERE y - x = 35;", [agg("COUNT")], [q(X0, "=", 35)], [], [xsub(xc("y"), xc("x"), I32)]),
    ("SELECT SUM(2 * x) FROM test WHERE x = 7;", [agg("SUM", X0)], [q("x", "=", 7)], [], [xmul(xl(I32, 2), xc("x"), I32)]),
    ("SELECT SUM(2 * x + z) FROM test WHERE x = 7;", [agg("SUM", X0)], [q("x", "=", 7)], [],
     [xadd(xmul(xl(I32, 2), xc("x"), I32), xcast(xc("z"), I32), I32)]),
    ("SELECT SUM(x + y) FROM test WHERE x - y = -35;", [agg("SUM", X0)], [q(X1, "=", -35)], [], [_xy, _xmy]),
    ("SELECT SUM(x + y - z) FROM test WHERE y - x = 35;", [agg("SUM", X0)], [q(X1, "=", 35)], [],
     [xsub(_xy, xcast(xc("z"), I32), I32), xsub(xc("y"), xc("x"), I32)]),
    ("SELECT SUM(x * y + 15) FROM test WHERE x + y + 1 = 50;", [agg("SUM", X0)], [q(X1, "=", 50)], [], [_xy15, xadd(_xy, xl(I32, 1), I32)]),
    ("SELECT SUM(x * y + 15) FROM test WHERE x + y + z + 1 = 151;", [agg("SUM", X0)], [q(X1, "=", 151)], [],
     [_xy15, xadd(_xyz, xl(I32, 1), I32)]),
    ("SELECT SUM(x * y + 15) FROM test WHERE x + y + z + t + 1 = 1152;", [agg("SUM", X0)], [q(X1, "=", 1152)], [],
     [_xy15, xadd(_xyzt, xl(I64, 1), I64)]),
    ("SELECT MIN(x * y + 15) FROM test WHERE x + y + 1 = 50;", [agg("MIN", X0)], [q(X1, "=", 50)], [], [_xy15, xadd(_xy, xl(I32, 1), I32)]),
    ("SELECT MAX(x * y + 15) FROM test WHERE x + y + z + 1 = 151;", [agg("MAX", X0)], [q(X1, "=", 151)], [],
     [_xy15, xadd(_xyz, xl(I32, 1), I32)]),
    ("SELECT AVG(x + y) FROM test;", [agg("AVG", X0)], [], [], [_xy]),
    ("SELECT AVG(x + y + z) FROM test;", [agg("AVG", X0)], [], [], [_xyz]),
    ("SELECT AVG(x + y + z + t) FROM test;", [agg("AVG", X0)], [], [], [_xyzt]),
    ("SELECT AVG(u * f) FROM test;", [agg("AVG", X0)], [], [], [xmul(xcast(xc("u"), F32), xc("f"), F32)]),
    ("SELECT AVG(u * d) FROM test;", [agg("AVG", X0)], [], [], [xmul(xcast(xc("u"), F64), xc("d"), F64)]),
    ("SELECT SUM(-y) FROM test;", [agg("SUM", X0)], [], [], [xneg(xc("y"), I32)]),
    ("SELECT SUM(-z) FROM test;", [agg("SUM", X0)], [], [], [xneg(xc("z"), I16)]),
    ("SELECT SUM(-t) FROM test;", [agg("SUM", X0)], [], [], [xneg(xc("t"), I64)]),
    ("SELECT SUM(-f) FROM test;", [agg("SUM", X0)], [], [], [xneg(xc("f"), F32)]),
    ("SELECT SUM(-d) FROM test;", [agg("SUM", X0)], [], [], [xneg(xc("d"), F64)]),
    ("SELECT COUNT(*) FROM test WHERE x < y AND 1=1;", [agg("COUNT")], [q(X0, "=", 1)], [], [xcmp(xc("x"), "<", xc("y"))]),   # (folded)
    ("SELECT COUNT(*) FROM test WHERE x < y OR 1<1;", [agg("COUNT")], [q(X0, "=", 1)], [], [xcmp(xc("x"), "<", xc("y"))]),
    ("SELECT COUNT(*) FROM test WHERE (x > 7 AND y / (x - 7) < 44);", [agg("COUNT")], [q(X0, "=", 1)], [],   # the deferred qual
     [xlogic(xcmp(xc("x"), ">", xl(I32, 7)), "AND",
             xcmp(xdiv(xc("y"), xsub(xc("x"), xl(I32, 7), I32), I32), "<", xl(I32, 44)), short_circuit=True)]),
    ("SELECT COUNT(*) FROM test WHERE fx + 1 IS NULL;", [agg("COUNT")], [q(X0, "IS NULL", 0)], [], [xadd(xc("fx"), xl(I32, 1), I32)]),
    # six conjuncts for four quals (:1907): three quals, the rest ANDed in one BOOLEAN expression
    ("SELECT COUNT(*) FROM test WHERE x > 6 AND x < 8 AND z > 100 AND z < 102 AND t > 1000 AND t < 1002;", [agg("COUNT")],
     [q("x", ">", 6), q("x", "<", 8), q("z", ">", 100), q(X0, "=", 1)], [],
     [xlogic(xlogic(xcmp(xc("z"), "<", xl(I16, 102)), "AND", xcmp(xc("t"), ">", xl(I64, 1000))), "AND", xcmp(xc("t"), "<", xl(I64, 1002)))]),
    # Select.In (:7361-7367): an IN list is an OR group of equalities ...
    ("SELECT COUNT(*) FROM test WHERE x IN (7, 8);", [agg("COUNT")], [("x", capi.EQ | (1 << 8), 7), ("x", capi.EQ | (1 << 8), 8)], [], []),
    ("SELECT COUNT(*) FROM test WHERE x IN (9, 10);", [agg("COUNT")], [("x", capi.EQ | (1 << 8), 9), ("x", capi.EQ | (1 << 8), 10)], [], []),
    ("SELECT COUNT(*) FROM test WHERE z IN (101, 102);", [agg("COUNT")], [("z", capi.EQ | (1 << 8), 101), ("z", capi.EQ | (1 << 8), 102)], [], []),
    ("SELECT COUNT(*) FROM test WHERE z IN (201, 202);", [agg("COUNT")], [("z", capi.EQ | (1 << 8), 201), ("z", capi.EQ | (1 << 8), 202)], [], []),
    # ... or, as a value, the OR of the comparisons
    ("SELECT COUNT(*) FROM test WHERE z IN (101, 102) OR (x = 8 AND t IS NULL);", [agg("COUNT")], [q(X1, "=", 1)], [],   # (not the reference's text)
     [EX(["x", "t"], lambda ix, nc: xcmp(xc("x"), "=", xl(I32, 8)).build(ix, nc).logical(capi.EX_AND, Expr.col(ix["t"]).is_null())),
      xlogic(xlogic(xcmp(xc("z"), "=", xl(I16, 101)), "OR", xcmp(xc("z"), "=", xl(I16, 102))), "OR", xref(0))]),
    # Select.Case (:5358-5380).  The SUM's CASE has 19 nodes: its conditions cannot raise, so they are earlier expressions
    ("SELECT SUM(CASE WHEN x BETWEEN 6 AND 7 THEN 1 WHEN x BETWEEN 8 AND 9 THEN 2 ELSE 3 END) FROM test;", [agg("SUM", X2)], [], [],
     [xbetween("x", I32, 6, 7), xbetween("x", I32, 8, 9), xcase(xref(0), xl(I32, 1), xcase(xref(1), xl(I32, 2), xl(I32, 3), I32), I32)]),
    ("SELECT SUM(CASE WHEN x BETWEEN 6 AND 7 THEN 1 END) FROM test;", [agg("SUM", X0)], [], [],
     [xcase(xbetween("x", I32, 6, 7), xl(I32, 1), xnull(I32), I32)]),
    ("SELECT SUM(CASE WHEN x BETWEEN 6 AND 7 THEN 1 WHEN x BETWEEN 8 AND 9 THEN 2 ELSE 3 END) FROM test "
     "WHERE CASE WHEN y BETWEEN 42 AND 43 THEN 5 ELSE 4 END > 4;", [agg("SUM", X3)], [q(X0, ">", 4)], [],
     [xcase(xbetween("y", I32, 42, 43), xl(I32, 5), xl(I32, 4), I32), xbetween("x", I32, 6, 7), xbetween("x", I32, 8, 9),
      xcase(xref(1), xl(I32, 1), xcase(xref(2), xl(I32, 2), xl(I32, 3), I32), I32)]),
    ("SELECT SUM(CASE WHEN x BETWEEN 6 AND 7 THEN 1 WHEN x BETWEEN 8 AND 9 THEN 2 ELSE 3 END) FROM test "     # ASSERT_EQ(NULL, ..) there
     "WHERE CASE WHEN y BETWEEN 44 AND 45 THEN 5 ELSE 4 END > 4;", [agg("SUM", X3)], [q(X0, ">", 4)], [],
     [xcase(xbetween("y", I32, 44, 45), xl(I32, 5), xl(I32, 4), I32), xbetween("x", I32, 6, 7), xbetween("x", I32, 8, 9),
      xcase(xref(1), xl(I32, 1), xcase(xref(2), xl(I32, 2), xl(I32, 3), I32), I32)]),
    ("SELECT CASE WHEN x + y > 50 THEN 77 ELSE 88 END AS foo, COUNT(*) FROM test GROUP BY foo ORDER BY foo;", [key(), agg("COUNT")], [],
     [X0], [xcase(xcmp(_xy, ">", xl(I32, 50)), xl(I32, 77), xl(I32, 88), I32)]),
    ("SELECT SUM(CASE WHEN x BETWEEN 6 AND 7 THEN 1.1 WHEN x BETWEEN 8 AND 9 THEN 2.2 ELSE 3.3 END) FROM test "   # ASSERT_EQ(NULL, ..) there
     "WHERE CASE WHEN y BETWEEN 44 AND 45 THEN 5.1 ELSE 3.9 END > 4;", [agg("SUM", X3)], [q(X0, ">", 4.0)], [],
     [xcase(xbetween("y", I32, 44, 45), xl(F64, 5.1), xl(F64, 3.9), F64), xbetween("x", I32, 6, 7), xbetween("x", I32, 8, 9),
      xcase(xref(1), xl(F64, 1.1), xcase(xref(2), xl(F64, 2.2), xl(F64, 3.3), F64), F64)]),
    ("SELECT SUM(CASE WHEN x BETWEEN 6 AND 7 THEN 1.1 WHEN x BETWEEN 8 AND 9 THEN 2.2 ELSE 3.3 END) FROM test "   # (the same with rows)
     "WHERE CASE WHEN y BETWEEN 42 AND 43 THEN 5.1 ELSE 3.9 END > 4;", [agg("SUM", X3)], [q(X0, ">", 4.0)], [],
     [xcase(xbetween("y", I32, 42, 43), xl(F64, 5.1), xl(F64, 3.9), F64), xbetween("x", I32, 6, 7), xbetween("x", I32, 8, 9),
      xcase(xref(1), xl(F64, 1.1), xcase(xref(2), xl(F64, 2.2), xl(F64, 3.3), F64), F64)]),
    # Select.FilterAndGroupBy / GroupByFloat shapes (:2445, :2818-2820, :5319): expressions as group keys, in the filter and the argument at once
    ("SELECT MIN(x + y) FROM test WHERE x + y > 47 AND x + y < 53 GROUP BY x, y;", [agg("MIN", X0)], [q(X0, ">", 47), q(X0, "<", 53)],
     ["x", "y"], [_xy]),
    ("SELECT MIN(x + y) FROM test WHERE x + y > 47 AND x + y < 53 GROUP BY x + 1, x + y;", [agg("MIN", X0)], [q(X0, ">", 47), q(X0, "<", 53)],
     [X1, X0], [_xy, xadd(xc("x"), xl(I32, 1), I32)]),
    ("SELECT MIN(x + y) AS n FROM test WHERE x + y > 47 AND x + y < 53 GROUP BY f + 1, f + d ORDER BY n;", [agg("MIN", X0)],
     [q(X0, ">", 47), q(X0, "<", 53)], [X1, X2], [_xy, xadd(xc("f"), xl(F32, 1.0), F32), xadd(xcast(xc("f"), F64), xc("d"), F64)]),
    ("SELECT f + d AS s FROM test GROUP BY s ORDER BY s DESC;", [key()], [], [X0], [xadd(xcast(xc("f"), F64), xc("d"), F64)]),
    ("SELECT t + x, AVG(x) AS avg_x FROM test WHERE z <= 50 and t < 2000 GROUP BY t + x ORDER BY avg_x DESC", [key(), agg("AVG", "x")],
     [q("z", "<=", 50), q("t", "<", 2000)], [X0], [xadd(xc("t"), xcast(xc("x"), I64), I64)]),
    # Select.FloatAndDoubleTests (:2388-2420): FLOAT and DOUBLE arithmetic in arguments and filters (x * f is FLOAT arithmetic on
    # CAST(x AS FLOAT); f + d is DOUBLE arithmetic on CAST(f AS DOUBLE))
    ("SELECT SUM(f + d) FROM test;", [agg("SUM", X0)], [], [], [_fd := xadd(xcast(xc("f"), F64), xc("d"), F64)]),
    ("SELECT AVG(x * f) FROM test;", [agg("AVG", X0)], [], [], [xmul(xcast(xc("x"), F32), xc("f"), F32)]),
    ("SELECT AVG(z - 200) FROM test;", [agg("AVG", X0)], [], [], [xsub(xcast(xc("z"), I32), xl(I32, 200), I32)]),
    ("SELECT SUM(CAST(x AS FLOAT)) FROM test;", [agg("SUM", X0)], [], [], [xcast(xc("x"), F32)]),
    ("SELECT SUM(CAST(x AS FLOAT)) FROM test GROUP BY z;", [agg("SUM", X0)], [], ["z"], [xcast(xc("x"), F32)]),
    ("SELECT AVG(CAST(x AS FLOAT)) FROM test;", [agg("AVG", X0)], [], [], [xcast(xc("x"), F32)]),
    ("SELECT AVG(CAST(x AS FLOAT)) FROM test GROUP BY y;", [agg("AVG", X0)], [], ["y"], [xcast(xc("x"), F32)]),
    ("SELECT COUNT(*) FROM test WHERE f > 1.0 AND f < 1.2 OR (d > 2.0 AND d < 3.0);", [agg("COUNT")], [q(X1, "=", 1)], [],
     [xband("d", F64, 2.0, 3.0), xlogic(xband("f", F32, 1.0, 1.2), "OR", xref(0))]),
    ("SELECT SUM(x + y) FROM test WHERE d + f > 3.0 AND d + f < 4.0;", [agg("SUM", X0)], [q(X1, ">", 3.0), q(X1, "<", 4.0)], [],
     [_xy, xadd(xc("d"), xcast(xc("f"), F64), F64)]),
    ("SELECT SUM(f + d) FROM test WHERE x - y = -35;", [agg("SUM", X0)], [q(X1, "=", -35)], [], [_fd, _xmy]),
    ("SELECT SUM(f + d) FROM test WHERE x + y + 1 = 50;", [agg("SUM", X0)], [q(X1, "=", 50)], [], [_fd, xadd(_xy, xl(I32, 1), I32)]),
    ("SELECT SUM(f * d + 15) FROM test WHERE x + y + 1 = 50;", [agg("SUM", X0)], [q(X1, "=", 50)], [],
     [xadd(xmul(xcast(xc("f"), F64), xc("d"), F64), xl(F64, 15.0), F64), xadd(_xy, xl(I32, 1), I32)]),
    ("SELECT MIN(x), AVG(x * y), MAX(y + 7), AVG(x * f + 15), COUNT(*) FROM test WHERE x + y > 47 AND x + y < 51;",
     [agg("MIN", "x"), agg("AVG", X0), agg("MAX", X1), agg("AVG", X2), agg("COUNT")], [q(X3, ">", 47), q(X3, "<", 51)], [],
     [xmul(xc("x"), xc("y"), I32), xadd(xc("y"), xl(I32, 7), I32), xadd(xmul(xcast(xc("x"), F32), xc("f"), F32), xl(F32, 15.0), F32), _xy]),
    # Select.InValues (:2512-2523): lists beyond one program are runs of three comparisons each, ORed by the root
    ("SELECT x FROM test WHERE x IN (8, 9, 10, 11, 12, 13, 14) GROUP BY x ORDER BY x;", [key()], [q(X3, "=", 1)], ["x"],
     [xin("x", I32, [8, 9, 10]), xin("x", I32, [11, 12, 13]), xin("x", I32, [14]), xlogic(xlogic(xref(0), "OR", xref(1)), "OR", xref(2))]),
    ("SELECT y FROM test WHERE y IN (43, 44, 45, 46, 47, 48, 49) GROUP BY y ORDER BY y;", [key()], [q(X3, "=", 1)], ["y"],
     [xin("y", I32, [43, 44, 45]), xin("y", I32, [46, 47, 48]), xin("y", I32, [49]), xlogic(xlogic(xref(0), "OR", xref(1)), "OR", xref(2))]),
    ("SELECT t FROM test WHERE t NOT IN (1001, 1003, 1005, 1007, 1009, -10) GROUP BY t ORDER BY t;", [key()], [q(X2, "=", 1)], ["t"],
     [xin("t", I64, [1001, 1003, 1005]), xin("t", I64, [1007, 1009, -10]), xnot(xlogic(xref(0), "OR", xref(1)))]),
    # Select.FilterAndMultipleAggregation (:2578)
    ("SELECT MIN(x), AVG(x * y), MAX(y + 7), COUNT(*) FROM test WHERE x + y > 47 AND x + y < 51;",
     [agg("MIN", "x"), agg("AVG", X0), agg("MAX", X1), agg("COUNT")], [q(X2, ">", 47), q(X2, "<", 51)], [],
     [xmul(xc("x"), xc("y"), I32), xadd(xc("y"), xl(I32, 7), I32), _xy]),
    ("SELECT x, SUM(-y), COUNT(*) FROM test WHERE NOT (z > 100 AND t = 1002) GROUP BY x;",                    # (not the reference's text)
     [key(), agg("SUM", X0), agg("COUNT")], [q(X1, "=", 1)], ["x"],
     [xneg(xc("y"), I32), EX(["z", "t"], lambda ix, nc: xlogic(xcmp(xc("z"), ">", xl(I16, 100)), "AND",
                                                                xcmp(xc("t"), "=", xl(I64, 1002))).build(ix, nc).logical_not())]),
]


# Select.DivByZero (:7402-7432): EXPECT_THROW for a division by `x - x` in a group key, in MOD, in a filter — and the reference's
# own literal for the short-circuit OR: `WHERE x = x OR y / (x - x) = y` counts every row (2 * g_num_rows), because the operand
# with the unsafe division is evaluated second and only where the first does not decide (codegenLogicalShortCircuit).
_x_minus_x = xsub(xc("x"), xc("x"), I32)
_y_div0 = xdiv(xc("y"), _x_minus_x, I32)
DIVZERO_QUERIES = [
    ("SELECT COUNT(*) FROM test GROUP BY y / (x - x);", [agg("COUNT")], [], [X0], [_y_div0], capi.ERR_DIV_BY_ZERO),
    ("SELECT COUNT(*) FROM test GROUP BY z, y / (x - x);", [agg("COUNT")], [], ["z", X0], [_y_div0], capi.ERR_DIV_BY_ZERO),
    ("SELECT COUNT(*) FROM test GROUP BY MOD(y , (x - x));", [agg("COUNT")], [], [X0],
     [EX(["y", "x"], lambda ix, nc: Expr.col(ix["y"]).mod(_x_minus_x.build(ix, nc), I32))], capi.ERR_DIV_BY_ZERO),
    ("SELECT COUNT(*) FROM test WHERE y / (x - x) = 0;", [agg("COUNT")], [q(X0, "=", 0)], [], [_y_div0], capi.ERR_DIV_BY_ZERO),
    ("SELECT COUNT(*) FROM test WHERE x = x OR  y / (x - x) = y;", [agg("COUNT")], [q(X0, "=", 1)], [],
     [xlogic(xcmp(xc("x"), "=", xc("x")), "OR", xcmp(_y_div0, "=", xc("y")), short_circuit=True)], 2 * 10),
]


# Select.BooleanColumn (:7669-7695): every expectation but the last is the reference's own ASSERT_EQ literal (g_num_rows = 10).
# A bare BOOLEAN column as a filter is the qual `column = 1`; NOT over it is an expression.
BOOLEAN_QUERIES = [
    ("SELECT COUNT(*) FROM test WHERE bn;", [agg("COUNT")], [q("bn", "=", 1)], [], [], [(15,)]),
    ("SELECT COUNT(*) FROM test WHERE b;", [agg("COUNT")], [q("b", "=", 1)], [], [], [(10,)]),
    ("SELECT COUNT(*) FROM test WHERE NOT bn;", [agg("COUNT")], [q(X0, "=", 1)], [], [xnot(xc("bn"))], [(5,)]),
    ("SELECT COUNT(*) FROM test WHERE x < 8 AND bn;", [agg("COUNT")], [q("x", "<", 8), q("bn", "=", 1)], [], [], [(15,)]),
    ("SELECT COUNT(*) FROM test WHERE x < 8 AND NOT bn;", [agg("COUNT")], [q("x", "<", 8), q(X0, "=", 1)], [], [xnot(xc("bn"))], [(0,)]),
    ("SELECT COUNT(*) FROM test WHERE x > 7 OR false;", [agg("COUNT")], [q("x", ">", 7)], [], [], [(5,)]),                  # (folded)
    ("SELECT MAX(x) FROM test WHERE b = CAST('t' AS boolean);", [agg("MAX", "x")], [q("b", "=", 1)], [], [], [(7,)]),
    (" SELECT SUM(2 *(CASE when x = 7 then 1 else 0 END)) FROM test;", [agg("SUM", X0)], [], [],
     [xmul(xl(I32, 2), xcase(xcmp(xc("x"), "=", xl(I32, 7)), xl(I32, 1), xl(I32, 0), I32), I32)], [(30,)]),
    ("SELECT COUNT(*) AS n FROM test GROUP BY x = 7, b ORDER BY n;", [agg("COUNT")], [], [X0, "b"], [xcmp(xc("x"), "=", xl(I32, 7))], None),
    # Select.CastRoundNullable (:7862-7880): EXPECT_EQ(10, ..) and EXPECT_EQ(11, first key); 8 * 1.6 = 12.8 rounds to 13 (DEF_ROUND_NULLABLE)
    ("SELECT COUNT(*) FROM test WHERE CAST(fn AS INT) IS NULL;", [agg("COUNT")], [q(X0, "IS NULL", 0)], [], [xcast(xc("fn"), I32)], [(10,)]),
    ("SELECT CAST(CAST(x AS FLOAT) * 1.6 AS INT) AS key0 FROM test GROUP BY key0 ORDER BY key0;", [key()], [], [X0],
     [xcast(xmul(xcast(xc("x"), F32), xl(F32, 1.6), F32), I32)], [(11,), (13,)]),
    # Select.OstensibleTautologyPredicate (:27648-27668): COUNT(*) WHERE ofd = ofd == COUNT(*) - COUNT_IF(ofd IS NULL)
    ("SELECT COUNT(*) FROM test WHERE ofd = ofd;", [agg("COUNT")], [q(X0, "=", 1)], [], [xcmp(xc("ofd"), "=", xc("ofd"))], [(15,)]),
    # (the same identity over encoded columns — date in days, fixed(16) INT, dictionary ids — decoded by the expression's column node)
    ("SELECT COUNT(*) FROM test WHERE o1 = o1;", [agg("COUNT")], [q(X0, "=", 1)], [], [xcmp(xc("o1"), "=", xc("o1"))], [(15,)]),
    ("SELECT COUNT(*) FROM test WHERE fx = fx AND str = str;", [agg("COUNT")], [q(X0, "=", 1), q(X1, "=", 1)], [],
     [xcmp(xc("fx"), "=", xc("fx")), xcmp(xc("str"), "=", xc("str"))], [(15,)]),
    ("SELECT COUNT(*) FROM test WHERE NOT b;", [agg("COUNT")], [q(X0, "=", 1)], [], [xnot(xc("b"))], [(5,)]),             # (not the reference's: NOT NULL is not TRUE)
    ("SELECT COUNT(*) FROM test WHERE b IS NULL OR NOT bn;", [agg("COUNT")], [q(X0, "=", 1)], [],                          # (not the reference's)
     [EX(["b", "bn"], lambda ix, nc: Expr.col(ix["b"]).is_null().logical(capi.EX_OR, Expr.col(ix["bn"]).logical_not()))], [(10,)]),
]


@pytest.mark.parametrize("qi", range(len(BOOLEAN_QUERIES)), ids=[s[0].strip()[7:60].replace(" ", "_") for s in BOOLEAN_QUERIES])
def test_reference_boolean_column_queries(oracle, qi):
    from tests.test_rowlogic_emu import _emu_execute
    sql, targets, quals, group, exprs, expect = BOOLEAN_QUERIES[qi]
    descs, frags, db = _table()
    ra, frags = _unit_x(descs, frags, targets, quals, group, exprs, num_tuples=sum(REPEAT))
    plan = ra.to_plan()
    qm, buf, code = oracle.execute(plan, frags, n_threads=3)
    eq, ebuf, ecode = _emu_execute(Case("ref", ra, frags), plan, None)
    assert code == 0 and ecode == 0, (sql, code, ecode)
    want = sorted(expect if expect is not None else [tuple(r) for r in db.execute(sql).fetchall()])
    assert sorted(_rows(oracle.fetch_rows(qm, buf), qm)) == want, ("oracle", sql, want)
    assert sorted(_rows(oracle.fetch_rows(eq, ebuf), eq)) == want, ("product row logic", sql, want)


# Select.OverflowAndUnderFlow (:7495-7531): the filters the reference runs with c(..) — no overflow, the count must equal SQLite's —
# and the ones it EXPECT_THROWs: error 7.  Integer literals are INT unless they need BIGINT (RelAlgTranslator::translateLiteral),
# so `z + 32666` is INT arithmetic on CAST(z AS INT) and does not overflow while `x + 2147483640` does for x = 8 (and not for 7).
# `-ufq - ..` is the reference's own case of unary minus over a NOT NULL column that holds the type's minimum.
_zi = xcast(xc("z"), I32)
OVERFLOW_QUERIES = [
    ("SELECT COUNT(*) FROM test WHERE z + 32600 > 0;", [xadd(_zi, xl(I32, 32600), I32)], ">", 0, None),
    ("SELECT COUNT(*) FROM test WHERE z + 32666 > 0;", [xadd(_zi, xl(I32, 32666), I32)], ">", 0, None),
    ("SELECT COUNT(*) FROM test WHERE -32670 - z < 0;", [xsub(xl(I32, -32670), _zi, I32)], "<", 0, None),
    ("SELECT COUNT(*) FROM test WHERE (z + 16333) * 2 > 0;", [xmul(xadd(_zi, xl(I32, 16333), I32), xl(I32, 2), I32)], ">", 0, None),
    ("SELECT COUNT(*) FROM test WHERE x + 2147483640 > 0;", [xadd(xc("x"), xl(I32, 2147483640), I32)], ">", 0, 7),
    ("SELECT COUNT(*) FROM test WHERE -x - 2147483642 < 0;", [xsub(xneg(xc("x"), I32), xl(I32, 2147483642), I32)], "<", 0, 7),
    ("SELECT COUNT(*) FROM test WHERE t + 9223372036854774000 > 0;", [xadd(xc("t"), xl(I64, 9223372036854774000), I64)], ">", 0, None),
    ("SELECT COUNT(*) FROM test WHERE t + 9223372036854775000 > 0;", [xadd(xc("t"), xl(I64, 9223372036854775000), I64)], ">", 0, 7),
    ("SELECT COUNT(*) FROM test WHERE -t - 9223372036854775000 < 0;", [xsub(xneg(xc("t"), I64), xl(I64, 9223372036854775000), I64)], "<", 0, 7),
    ("SELECT COUNT(*) FROM test WHERE ofd + x - 2 > 0;", [xsub(xadd(xc("ofd"), xc("x"), I32), xl(I32, 2), I32)], ">", 0, 7),
    ("SELECT COUNT(*) FROM test WHERE ufd * 3 - ofd * 1024 < -2;",
     [xsub(xmul(xc("ufd"), xl(I32, 3), I32), xmul(xc("ofd"), xl(I32, 1024), I32), I32)], "<", -2, 7),
    ("SELECT COUNT(*) FROM test WHERE ofd * 2 > 0;", [xmul(xc("ofd"), xl(I32, 2), I32)], ">", 0, 7),
    ("SELECT COUNT(*) FROM test WHERE ofq + 1 > 0;", [xadd(xc("ofq"), xl(I64, 1), I64)], ">", 0, 7),
    ("SELECT COUNT(*) FROM test WHERE -ufq - 9223372036854775000 > 0;", [xsub(xneg(xc("ufq"), I64), xl(I64, 9223372036854775000), I64)], ">", 0, 7),
    ("SELECT COUNT(*) FROM test WHERE -92233720368547758 - ofq <= 0;", [xsub(xl(I64, -92233720368547758), xc("ofq"), I64)], "<=", 0, 7),
]


@pytest.mark.parametrize("qi", range(len(OVERFLOW_QUERIES)), ids=[s[0][30:75].replace(" ", "_") for s in OVERFLOW_QUERIES])
def test_reference_overflow_queries(oracle, qi):
    from tests.test_rowlogic_emu import _emu_execute
    sql, exprs, op, lit, expect = OVERFLOW_QUERIES[qi]
    descs, frags, db = _table()
    ra, frags = _unit_x(descs, frags, [agg("COUNT")], [q(X0, op, lit)], [], exprs, num_tuples=sum(REPEAT))
    plan = ra.to_plan()
    qm, buf, code = oracle.execute(plan, frags, n_threads=3)
    eq, ebuf, ecode = _emu_execute(Case("ref", ra, frags), plan, None)
    if expect:
        assert code == ecode == expect, (sql, code, ecode)
    else:
        want = [tuple(r) for r in db.execute(sql).fetchall()]
        assert code == 0 and ecode == 0, (sql, code, ecode)
        assert _rows(oracle.fetch_rows(qm, buf), qm) == want and _rows(oracle.fetch_rows(eq, ebuf), eq) == want, (sql, want)


@pytest.mark.parametrize("qi", range(len(DIVZERO_QUERIES)), ids=[s[0][7:60].replace(" ", "_") for s in DIVZERO_QUERIES])
def test_reference_div_by_zero_queries(oracle, qi):
    from tests.test_rowlogic_emu import _emu_execute
    sql, targets, quals, group, exprs, expect = DIVZERO_QUERIES[qi]
    descs, frags, db = _table()
    ra, frags = _unit_x(descs, frags, targets, quals, group, exprs, num_tuples=sum(REPEAT))
    plan = ra.to_plan()
    qm, buf, code = oracle.execute(plan, frags, n_threads=3)
    eq, ebuf, ecode = _emu_execute(Case("ref", ra, frags), plan, None)
    if expect == capi.ERR_DIV_BY_ZERO:
        assert code == ecode == capi.ERR_DIV_BY_ZERO, (sql, code, ecode)
    else:
        assert code == 0 and ecode == 0, (sql, code, ecode)
        assert _rows(oracle.fetch_rows(qm, buf), qm) == [(expect,)] and _rows(oracle.fetch_rows(eq, ebuf), eq) == [(expect,)], sql


def _check_rows(sql, want, rows, fp, qm, name, rel):
    assert len(rows) == len(want), (name, sql, want, rows)
    for w, g in zip(want, rows):
        for t, (a, b) in enumerate(zip(w, g)):
            if a is None or b is None:
                assert a is None and b is None, (name, sql, t, w, g)
            elif fp[t]:
                rt, at = (F32_RTOL, F32_ATOL) if qm.target_arg_is_f32[t] else (rel, 0.0)
                assert math.isclose(a, b, rel_tol=rt, abs_tol=at), (name, sql, t, w, g)
            else:
                assert a == b, (name, sql, t, w, g)


@pytest.mark.parametrize("qi", range(len(EXPR_QUERIES)), ids=[s[0][7:70].replace(" ", "_") for s in EXPR_QUERIES])
def test_reference_expression_queries(oracle, qi):
    from tests.test_rowlogic_emu import _emu_execute
    sql, targets, quals, group, exprs = EXPR_QUERIES[qi]
    descs, frags, db = _table()
    ra, frags = _unit_x(descs, frags, targets, quals, group, exprs, num_tuples=sum(REPEAT))
    plan = ra.to_plan()
    qm, buf, code = oracle.execute(plan, frags, n_threads=3)
    assert code == 0, sql
    fp = [bool(qm.target_is_fp[t]) for t in range(qm.n_targets)]
    want = sorted((tuple(float(v) if f and v is not None else v for v, f in zip(r, fp)) for r in db.execute(sql).fetchall()), key=_key)
    eq, ebuf, ecode = _emu_execute(Case("ref", ra, frags), plan, None)
    assert ecode == 0, sql
    _check_rows(sql, want, sorted(_rows(oracle.fetch_rows(qm, buf), qm), key=_key), fp, qm, "oracle", 1e-12)
    _check_rows(sql, want, sorted(_rows(oracle.fetch_rows(eq, ebuf), eq), key=_key), fp, qm, "product row logic", 1e-12)


# ---- joins against the reference's `test_inner` (ExecuteTest.cpp:29719-29738: two rows)
INNER = {"x": (I32, False, [7, -9]), "y": (I32, True, [43, 72]), "xx": (I16, True, [7, -9])}
INNER_NAMES = list(INNER)
# join_test (:9785-9799): str / dup_str as the dictionary ids of 'foo', 'bar', 'baz'; dup_str repeats 'foo'
JOIN_TEST = {"x": (I32, False, [7, 8, 9]), "y": (I32, True, [43, None, None]), "str": (I32, False, [0, 1, 2]),
             "dup_str": (I32, False, [0, 0, 1])}
INNER_TABLES = {"test_inner": INNER, "join_test": JOIN_TEST}

# (SQL text, targets [("agg", kind, column, table)], quals on test, group-by (test columns),
#  join: (outer columns, inner columns, LEFT?))
JOIN_QUERIES = [
    ("SELECT COUNT(*) FROM test JOIN test_inner ON test.x = test_inner.x;",                      # :12857
     [("agg", capi.COUNT, None, 0)], [], [], (["x"], ["x"], False)),
    ("SELECT COUNT(*) FROM test LEFT JOIN test_inner ON test.x = test_inner.x WHERE test.y > 42;",  # :13371
     [("agg", capi.COUNT, None, 0)], [q("y", ">", 42)], [], (["x"], ["x"], True)),
    ("SELECT a.x, count(*) FROM test a LEFT JOIN test_inner b ON a.x = b.y WHERE a.x = 7 GROUP BY 1 ORDER BY 2;",  # :13681
     [("key", 0, None, 0), ("agg", capi.COUNT, None, 0)], [q("x", "=", 7)], ["x"], (["x"], ["y"], True)),
    ("SELECT a.x, COUNT(b.y) FROM test a LEFT JOIN test_inner b ON b.x = a.x GROUP BY a.x ORDER BY a.x;",  # :13513 without the LIKE
     [("key", 0, None, 0), ("agg", capi.COUNT, "y", 1)], [], ["x"], (["x"], ["x"], True)),
    ("SELECT a.x, SUM(b.y), MIN(b.xx), AVG(a.d) FROM test a JOIN test_inner b ON b.x = a.x GROUP BY a.x;",
     [("key", 0, None, 0), ("agg", capi.SUM, "y", 1), ("agg", capi.MIN, "xx", 1), ("agg", capi.AVG, "d", 0)], [], ["x"],
     (["x"], ["x"], False)),
    ("SELECT COUNT(*), SUM(a.t) FROM test a JOIN test_inner b ON a.x = b.x AND a.y = b.y;",       # composite key (:13385 shape)
     [("agg", capi.COUNT, None, 0), ("agg", capi.SUM, "t", 0)], [], [], (["x", "y"], ["x", "y"], False)),
    ("SELECT a.y, COUNT(b.x), MAX(b.xx) FROM test a LEFT JOIN test_inner b ON a.x = b.x AND a.y = b.y GROUP BY a.y;",
     [("key", 0, None, 0), ("agg", capi.COUNT, "x", 1), ("agg", capi.MAX, "xx", 1)], [], ["y"], (["x", "y"], ["x", "y"], True)),
    # one-to-many: join_test.dup_str holds 'foo' twice (:12859, :12867, :13482)
    ("SELECT COUNT(*) FROM test a JOIN join_test b ON a.str = b.dup_str;",
     [("agg", capi.COUNT, None, 0)], [], [], (["str"], ["dup_str"], False), "join_test"),
    ("SELECT a.x FROM test a JOIN join_test b ON a.str = b.dup_str GROUP BY a.x ORDER BY a.x;",
     [("key", 0, None, 0)], [], ["x"], (["str"], ["dup_str"], False), "join_test"),
    ("SELECT COUNT(*) FROM test a LEFT JOIN join_test b ON a.str = b.dup_str;",
     [("agg", capi.COUNT, None, 0)], [], [], (["str"], ["dup_str"], True), "join_test"),
    ("SELECT a.str, COUNT(*), SUM(b.x), COUNT(b.y) FROM test a LEFT JOIN join_test b ON a.str = b.dup_str GROUP BY a.str;",
     [("key", 0, None, 0), ("agg", capi.COUNT, None, 0), ("agg", capi.SUM, "x", 1), ("agg", capi.COUNT, "y", 1)], [], ["str"],
     (["str"], ["dup_str"], True), "join_test"),
]


def _join_case(descs, frags, db, spec):
    sql, targets, quals, group, (outer, inner, left) = spec[:5]
    tname = spec[5] if len(spec) > 5 else "test_inner"
    INNER, INNER_NAMES = INNER_TABLES[tname], list(INNER_TABLES[tname])
    used = []
    for n in list(group) + list(outer) + [t[2] for t in targets if t[0] == "agg" and t[2] and t[3] == 0] + [c for c, _, _ in quals]:
        if n not in used:
            used.append(n)
    idx = {n: i for i, n in enumerate(used)}
    src = [NAMES.index(n) for n in used]
    inner_arrays = {n: np.array([NULL[INNER[n][0]] if v is None else v for v in INNER[n][2]], dtype=NP[INNER[n][0]])
                    for n in INNER_NAMES}
    inner_descs = [InputColDescriptor(INNER[n][0], INNER[n][1], col_range([inner_arrays[n]], INNER[n][0], INNER[n][1]))
                   for n in INNER_NAMES]
    tx = []
    for t in targets:
        if t[0] == "key":
            tx.append(TargetExpr(capi.PROJECT_KEY, t[1]))
        elif t[3]:
            tx.append(TargetExpr(t[1], INNER_NAMES.index(t[2]), 1))
        else:
            tx.append(TargetExpr(t[1], -1 if t[2] is None else idx[t[2]]))
    ra = RelAlgExecutionUnit([descs[i] for i in src], tx, [Qual(idx[c], op, lit) for c, op, lit in quals],
                             [idx[g] for g in group], inner_col_descs=inner_descs,
                             join_outer_col=[idx[o] for o in outer] if len(outer) > 1 else idx[outer[0]],
                             join_kind=capi.JOIN_LEFT if left else capi.JOIN_INNER)
    keys = [inner_arrays[n] for n in inner]
    ktypes = [INNER[n][0] for n in inner]
    knull = [INNER[n][1] for n in inner]
    single = len(inner) == 1
    kr = col_range([keys[0]], ktypes[0], knull[0]) if single else None
    from heavydb_amd.executor import ExpressionRange
    case = Case("ref_join", ra, [[f[i] for i in src] for f in frags], [inner_arrays[n] for n in INNER_NAMES],
                keys[0] if single else keys, ktypes[0] if single else ktypes, kr if single else ExpressionRange(),
                False, join_one_to_many=1, join_key_nullable=knull[0] if single else knull)
    if tname not in [r[0] for r in db.execute("SELECT name FROM sqlite_master")]:
        db.execute(f"CREATE TABLE {tname} (" + ", ".join(INNER_NAMES) + ")")
        db.executemany(f"INSERT INTO {tname} VALUES (" + ",".join("?" * len(INNER_NAMES)) + ")",
                       list(zip(*[INNER[n][2] for n in INNER_NAMES])))
    return case, sql


def _compare(sql, db, qm, rows_by_engine):
    fp = [bool(qm.target_is_fp[t]) for t in range(qm.n_targets)]
    want = sorted((tuple(float(v) if f and v is not None else v for v, f in zip(r, fp))
                   for r in db.execute(sql).fetchall()), key=_key)
    for name, rows in rows_by_engine:
        rows = sorted(rows, key=_key)
        assert len(rows) == len(want), (name, sql, want, rows)
        for w, g in zip(want, rows):
            for t, (a, b) in enumerate(zip(w, g)):
                if a is None or b is None:
                    assert a is None and b is None, (name, sql, t, w, g)
                elif fp[t]:
                    rt, at = (F32_RTOL, F32_ATOL) if qm.target_arg_is_f32[t] else (1e-12, 0.0)
                    assert math.isclose(a, b, rel_tol=rt, abs_tol=at), (name, sql, t, w, g)
                else:
                    assert a == b, (name, sql, t, w, g)


@pytest.mark.parametrize("ji", range(len(JOIN_QUERIES)), ids=[s[0][7:70].replace(" ", "_") for s in JOIN_QUERIES])
def test_reference_join_queries(oracle, ji):
    from tests.test_rowlogic_emu import _emu_execute, _oracle_join
    descs, frags, db = _table()
    case, sql = _join_case(descs, frags, db, JOIN_QUERIES[ji])
    plan = case.ra.to_plan()
    oj = _oracle_join(oracle, case)
    qm, buf, code = oracle.execute(plan, case.frags, case.inner, oj, n_threads=3)
    assert code == 0
    eq, ebuf, ecode = _emu_execute(case, plan, oj)
    assert ecode == 0
    _compare(sql, db, qm, [("oracle", _rows(oracle.fetch_rows(qm, buf), qm)),
                           ("product row logic", _rows(oracle.fetch_rows(eq, ebuf), eq))])


def test_non_grouped_aggregates_are_null_aware(oracle):
    """The reference's rule for NonGroupedAggregate steps (TargetExprBuilder.cpp:684-690,
    OutputBufferInitialization.cpp:281-286): every aggregate with an argument skips the inline NULL
    pattern of the argument's type, NOT NULL column or not.  ufd holds INT32_MIN in ten of its twenty
    rows and ufq INT64_MIN in ten: COUNT / MIN see the other ten only — oracle and product agree."""
    from tests.test_rowlogic_emu import _emu_execute
    descs, frags, db = _table()
    ra, fr = _unit(descs, frags, [agg("COUNT", "ufd"), agg("MIN", "ufd"), agg("COUNT", "ufq"), agg("MIN", "ufq"), agg("COUNT")],
                   [], [], num_tuples=sum(REPEAT))
    plan = ra.to_plan()
    qm, buf, code = oracle.execute(plan, fr, n_threads=2)
    eq, ebuf, ecode = _emu_execute(Case("ng", ra, fr), plan, None)
    assert code == 0 and ecode == 0
    for rows in (_rows(oracle.fetch_rows(qm, buf), qm), _rows(oracle.fetch_rows(eq, ebuf), eq)):
        assert rows == [(10, -2147483647, 10, -1, 20)]


# ---- Select.CountIf / Select.SumIf (ExecuteTest.cpp:4020-4198): the reference's own fixture table
# `data_types_basic5` (numeric columns; tests/golden/ref_data_types_basic5_numeric.json, generated from
# Tests/Import/datafiles/data_types_basic5.csv.gz by tests/golden/gen_data_types_basic5.py) and its query
# generators.  The reference checks COUNT_IF(cond) against COUNT(1) WHERE cond and SUM_IF(v, cond) against
# SUM(CASE WHEN cond THEN v END) on its own engine; here the first form runs through the oracle and the product's
# row logic and the second form runs on SQLite.
_B5_TYPES = {"int8_t": I8, "int16_t": I16, "int32_t": I32, "int64_t": I64, "float": F32, "double": F64}
_B5_CONDS = {" IS NULL": (capi.IS_NULL, 0), " IS NOT NULL": (capi.IS_NOT_NULL, 0), " > 0": (capi.GT, 0)}


def basic5_table():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_data_types_basic5_numeric.json")) as f:
        d = json.load(f)
    names = list(d["columns"])
    arrays, descs = {}, {}
    for n in names:
        t = _B5_TYPES[d["types"][n]]
        vals = d["columns"][n]
        a = np.array([NULL[t] if v is None else v for v in vals], dtype=NP[t])
        arrays[n] = a
        descs[n] = InputColDescriptor(t, True, col_range([a], t, True))
    db = sqlite3.connect(":memory:")
    db.execute("CREATE TABLE data_types_basic5 (" + ", ".join(names) + ")")
    py = [[None if v is None else (float(np.float32(v)) if d["types"][n] == "float" else v) for v in d["columns"][n]] for n in names]
    db.executemany("INSERT INTO data_types_basic5 VALUES (" + ",".join("?" * len(names)) + ")", list(zip(*py)))
    return names, arrays, descs, db


def basic5_unit(arrays, descs, cols, targets, quals=(), group=()):
    """cols: the table columns the step fetches; targets / quals / group index into that list."""
    idx = {n: i for i, n in enumerate(cols)}
    tx = []
    for t in targets:
        if t[0] == "key":
            tx.append(TargetExpr(capi.PROJECT_KEY, 0))
        else:
            _, kind, col, cond = t
            c = None if cond is None else Qual(idx[cond[0]], *_B5_CONDS[cond[1]])
            tx.append(TargetExpr(kind, -1 if col is None else idx[col], cond=c))
    ra = RelAlgExecutionUnit([descs[n] for n in cols], tx, [Qual(idx[c], *_B5_CONDS[op]) for c, op in quals],
                             [idx[g] for g in group], num_tuples=len(arrays[cols[0]]))
    n = len(arrays[cols[0]])
    frags = [[arrays[c][i:i + 32] for c in cols] for i in range(0, n, 32)]
    return ra, frags


def _both_engines(oracle, ra, frags):
    from tests.test_rowlogic_emu import _emu_execute
    plan = ra.to_plan()
    qm, buf, code = oracle.execute(plan, frags, n_threads=3)
    assert code == 0
    eq, ebuf, ecode = _emu_execute(Case("ref", ra, frags), plan, None)
    assert ecode == 0
    return qm, [("oracle", _rows(oracle.fetch_rows(qm, buf), qm)), ("product row logic", _rows(oracle.fetch_rows(eq, ebuf), eq))]


def count_if_queries():
    """(SQL as the reference writes it, the COUNT(1) form it is checked against, step description)"""
    names, arrays, descs, db = basic5_table()
    out = []
    for col in names:                                            # 1. non-group by (:4058-4074)
        for cond in [" IS NULL", " > 0"]:
            out.append((f"SELECT COUNT_IF({col}{cond}) FROM data_types_basic5",
                        f"SELECT COUNT(1) FROM data_types_basic5 WHERE {col}{cond}",
                        ([col], [("agg", capi.COUNT_IF, None, (col, cond))], (), ())))
    for col in names:                                            # 2. group-by (:4076-4088)
        out.append((f"SELECT CNT FROM (SELECT {col}, COUNT_IF({col} IS NULL) CNT FROM data_types_basic5 WHERE {col} IS NULL GROUP BY {col}) T",
                    f"SELECT COUNT(1) FROM data_types_basic5 WHERE {col} IS NULL",
                    ([col], [("key",), ("agg", capi.COUNT_IF, None, (col, " IS NULL"))], ((col, " IS NULL"),), (col,))))
    return (names, arrays, descs, db), out


def sum_if_queries():
    names, arrays, descs, db = basic5_table()
    out = []
    for col in names:                                            # 1. non-group by (:4135-4158)
        for op in _B5_CONDS:
            out.append((f"SELECT SUM_IF({col}, {col}{op}) FROM data_types_basic5",
                        f"SELECT SUM(CASE WHEN {col}{op} THEN {col} END) FROM data_types_basic5",
                        ([col], [("agg", capi.SUM_IF, col, (col, op))], (), ()), col))
    for col in names[1:]:                                        # 3. agg col and cond col are different (:4180-4196)
        for op in _B5_CONDS:
            out.append((f"SELECT SUM_IF(Tiny_int, {col}{op}) FROM data_types_basic5",
                        f"SELECT SUM(CASE WHEN {col}{op} THEN Tiny_int END) FROM data_types_basic5",
                        (["Tiny_int", col], [("agg", capi.SUM_IF, "Tiny_int", (col, op))], (), ()), "Tiny_int"))
    return (names, arrays, descs, db), out


def _expected_sum(db, arrays, alt_sql, value_col):
    """SQLite runs the alternative query; BIGINT sums are taken modulo 2^64 like the reference's agg_sum (SQLite
    raises on overflow), selected with the same CASE as a 0 / 1 flag."""
    if value_col == "Big_int":
        flags = [r[0] for r in db.execute(alt_sql.replace(f"SUM(CASE WHEN", "SELECT_FLAG(").replace("SELECT SELECT_FLAG(", "SELECT (CASE WHEN")
                                          .replace(f" THEN {value_col} END)", " THEN 1 ELSE 0 END)")).fetchall()]
        sel = arrays[value_col][np.array(flags, dtype=bool)]
        sel = sel[sel != NULL[I64]]
        return None if sel.size == 0 else int(np.sum(sel.astype(np.uint64), dtype=np.uint64).astype(np.int64))
    return db.execute(alt_sql).fetchone()[0]


def test_select_count_if(oracle):
    (names, arrays, descs, db), queries = count_if_queries()
    assert len(queries) == 18
    for sql, alt, (cols, targets, quals, group) in queries:
        want = db.execute(alt).fetchone()[0]
        ra, frags = basic5_unit(arrays, descs, cols, targets, quals, group)
        qm, engines = _both_engines(oracle, ra, frags)
        for name, rows in engines:
            if group:   # one group (the NULL key) when the column has NULLs, none otherwise; CNT is the last target
                assert [r[-1] for r in rows] == ([want] if want else []), (name, sql, rows, want)
            else:
                assert rows == [(want,)], (name, sql, rows, want)


def test_select_sum_if(oracle):
    (names, arrays, descs, db), queries = sum_if_queries()
    assert len(queries) == 33
    for sql, alt, (cols, targets, quals, group), value_col in queries:
        want = _expected_sum(db, arrays, alt, value_col)
        ra, frags = basic5_unit(arrays, descs, cols, targets, quals, group)
        qm, engines = _both_engines(oracle, ra, frags)
        for name, rows in engines:
            got = rows[0][0]
            if want is None or got is None:
                assert want is None and got is None, (name, sql, got, want)
            elif isinstance(got, float):
                rt, at = (F32_RTOL, F32_ATOL) if value_col == "Float_" else (1e-12, 0.0)
                assert math.isclose(got, want, rel_tol=rt, abs_tol=at), (name, sql, got, want)
            else:
                assert got == want, (name, sql, got, want)


def test_select_sum_if_grouped(oracle):
    """:4160-4178 — SELECT col, SUM_IF(col, col <cond>) FROM test GROUP BY 1 on the `test` table."""
    descs, frags, db = _table()
    for col in ["fn", "dn", "u", "ofd", "smallint_nulls"]:
        for op, (opc, lit) in _B5_CONDS.items():
            alt = f"SELECT {col}, SUM(CASE WHEN {col}{op} THEN {col} END) FROM test GROUP BY 1"
            i = NAMES.index(col)
            ra = RelAlgExecutionUnit([descs[i]], [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.SUM_IF, 0, cond=Qual(0, opc, lit))],
                                     [], [0], num_tuples=sum(REPEAT))
            qm, engines = _both_engines(oracle, ra, [[f[i]] for f in frags])
            _compare(alt, db, qm, engines)


# ---- Select.AggregateOnEmptyTable (ExecuteTest.cpp:2298-2331) and Select.NullGroupBy (:1870-1883)
# empty_test_table (id int, x bigint, y int, z smallint, t tinyint, f float, d double, b boolean), no rows (:10354)
EMPTY_COLS = {"id": I32, "x": I64, "y": I32, "z": I16, "t": I8, "f": F32, "d": F64, "b": I8}
EMPTY_TABLE_QUERIES = []
for _where in ("", " WHERE id > 5"):
    for _fn, _cols in (("AVG", "xyztfd"), ("MIN", "xyztfdb"), ("MAX", "xyztfdb"), ("SUM", "xyztfd"), ("COUNT", "xyztfdb")):
        EMPTY_TABLE_QUERIES.append(("SELECT " + ", ".join(f"{_fn}({c})" for c in _cols) + " FROM empty_test_table" + _where + ";",
                                    _fn, list(_cols), bool(_where)))


def empty_table_unit(fn, cols, with_where, n_frags):
    """n_frags = 0: the table has no fragments; 1: one fragment of zero rows (both occur in the reference: an empty
    table, and "skipped fragment" — ExecuteTest.cpp:2310)."""
    names = ["id"] + cols
    descs = [InputColDescriptor(EMPTY_COLS[n], True, ExpressionRange(False, 0, 0)) for n in names]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(A[fn], 1 + i) for i in range(len(cols))],
                             [Qual(0, capi.GT, 5)] if with_where else [], [], num_tuples=0)
    frags = [[np.zeros(0, dtype=NP[EMPTY_COLS[n]]) for n in names]] * n_frags
    return ra, frags


@pytest.mark.parametrize("n_frags", [0, 1])
@pytest.mark.parametrize("qi", range(10))
def test_select_aggregate_on_empty_table(oracle, qi, n_frags):
    sql, fn, cols, with_where = EMPTY_TABLE_QUERIES[qi]
    db = sqlite3.connect(":memory:")
    db.execute("CREATE TABLE empty_test_table (" + ", ".join(EMPTY_COLS) + ")")
    want = db.execute(sql).fetchall()
    assert want == [tuple([0 if fn == "COUNT" else None] * len(cols))]
    ra, frags = empty_table_unit(fn, cols, with_where, n_frags)
    qm, engines = _both_engines(oracle, ra, frags)
    for name, rows in engines:
        assert rows == want, (name, sql, rows)


def null_group_by_unit(t):
    """CREATE TABLE table_null_group_by (val TEXT | DOUBLE); INSERT NULL; SELECT val FROM ... GROUP BY val"""
    if t == "TEXT":     # a dictionary-encoded string column holding one NULL id
        desc = InputColDescriptor(I32, True, ExpressionRange(True, 0, -1, True), capi.ENC_DICT, 0)
        col = np.array([NULL[I32]], dtype=np.int32)
    else:
        desc = InputColDescriptor(F64, True, ExpressionRange(True, 0, 0, True, 0.0, -1.0))
        col = np.array([NULL[F64]], dtype=np.float64)
    return RelAlgExecutionUnit([desc], [TargetExpr(capi.PROJECT_KEY, 0)], [], [0], num_tuples=1), [[col]]


@pytest.mark.parametrize("t", ["TEXT", "DOUBLE"])
def test_select_null_group_by(oracle, t):
    ra, frags = null_group_by_unit(t)
    try:
        qm, engines = _both_engines(oracle, ra, frags)
    except capi.Mi355qError as e:
        pytest.fail(f"GROUP BY a column that only holds NULL was rejected: {e}")
    for name, rows in engines:
        assert rows == [(None,)], (name, rows)    # one group, its key NULL (the reference only checks that it runs)

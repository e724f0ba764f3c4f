"""Host-side mirror of the reference's operator interface for ONE query step.

Names follow the reference so the parity tests read like Tests/GroupByTest.cpp:73-151:
build a `RelAlgExecutionUnit` by hand, call `Executor.executeWorkUnit`, iterate the
`ResultSet` with `getNextRow` / `rowCount`.

  RelAlgExecutionUnit   QueryEngine/RelAlgExecutionUnit.h:167-218 (the used subset)
  FetchResult           QueryEngine/ColumnFetcher.h:46-49 (raw column pointers per fragment)
  Executor              QueryEngine/Execute.h:417; executeWorkUnit Execute.h:719
  ResultSet             QueryEngine/ResultSet.h:263 (getNextRow), :327 (rowCount)
  HashJoin              QueryEngine/JoinHashTable/HashJoin.h (getInstance -> perfect/baseline)

Everything here is plumbing over the C-ABI in include/mi355q.h; the compute is the HIP
library.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import capi
from .capi import (AVG, COUNT, DOUBLE, INT8, INT16, INT32, INT64, MAX, MIN, PROJECT_KEY, SUM,
                   check)


@dataclass
class ExpressionRange:
    """ExpressionRange (QueryEngine/ExpressionRange.h): what fragment metadata tells the
    planner about a column."""
    valid: bool = False
    min: int = 0
    max: int = 0
    has_nulls: bool = False
    fp_min: float = 0.0
    fp_max: float = 0.0
    bucket: int = 0

    def to_c(self) -> capi.Range:
        return capi.Range(int(self.valid), int(self.has_nulls), int(self.min), int(self.max),
                          float(self.fp_min), float(self.fp_max), int(self.bucket))


@dataclass
class InputColDescriptor:
    """InputColDescriptor + SQLTypeInfo of a fixed-width column: `type` is the chunk's storage
    type; `encoding` / `logical_type` describe kENCODING_FIXED / _DICT / _DATE_IN_DAYS columns
    (decoded on load, DecodersImpl.h); the range is in decoded values."""
    type: int
    nullable: bool = False
    range: ExpressionRange = field(default_factory=ExpressionRange)
    encoding: int = 0
    logical_type: int = 0

    def to_c(self) -> capi.ColDesc:
        return capi.ColDesc(self.type, int(self.nullable), self.encoding, self.logical_type)


@dataclass
class Qual:
    """simple_quals entry: `col <op> literal`."""
    col: int
    op: int
    literal: float | int = 0  # unused by IS_NULL / IS_NOT_NULL
    or_group: int = 0         # 1..3: member of that disjunction (quals sharing a group are OR-ed, everything else AND-ed)


@dataclass
class TargetExpr:
    """target_exprs entry: aggregate kind + argument column (-1 = COUNT(*)); table 1 reads
    an inner-table column through the join's row id.  PROJECT_KEY: `col` is the index of the
    projected key within groupby_exprs (default: the first)."""
    agg: int
    col: int = -1
    table: int = 0
    cond: Optional["Qual"] = None   # COUNT_IF / SUM_IF: the condition `col <op> literal`


@dataclass
class ExprNode:
    """One micro-op of a projected expression (mi355q_expr_node; postfix order)."""
    op: int
    type: int = 0
    arg: int = 0
    ilit: int = 0
    flit: float = 0.0
    null_lit: int = 0     # the node's `reserved`: EX_LIT: 1 = the NULL of `type`; EX_AND / EX_OR: 1 = the short-circuit form


@dataclass
class Expr:
    """A projected expression = a virtual outer column (Analyzer::UOper kCAST / BinOper kPLUS, kMINUS,
    kMULTIPLY over ColumnVar and Constant, as CodeGenerator::codegenCast / codegenArith compile them).
    Build with the helpers: Expr.col(c), Expr.lit(type, v), e.cast(type), e.add(other) / .sub / .mul."""
    nodes: List[ExprNode]
    range: ExpressionRange = field(default_factory=ExpressionRange)

    @staticmethod
    def col(c: int) -> "Expr":
        return Expr([ExprNode(capi.EX_COL, 0, c)])

    @staticmethod
    def lit(type: int, v) -> "Expr":
        fp = type in (DOUBLE, capi.FLOAT)
        return Expr([ExprNode(capi.EX_LIT, type, 0, 0 if fp else int(v), float(v) if fp else 0.0)])

    @staticmethod
    def null(type: int) -> "Expr":
        """the NULL constant of a type (the ELSE of a CASE without one)"""
        return Expr([ExprNode(capi.EX_LIT, type, null_lit=1)])

    def cast(self, type: int) -> "Expr":
        return Expr(self.nodes + [ExprNode(capi.EX_CAST, type)])

    def cmp(self, op: int, other: "Expr") -> "Expr":
        """self <op> other (capi.EX_EQ .. EX_GE; Analyzer::BinOper with a comparison): a BOOLEAN, stored as INT8 1 / 0 / NULL"""
        return Expr(self.nodes + other.nodes + [ExprNode(op, capi.INT8)])

    @staticmethod
    def case(cond: "Expr", then: "Expr", otherwise: "Expr", type: int) -> "Expr":
        """CASE WHEN cond THEN then ELSE otherwise END (Analyzer::CaseExpr): stack order ELSE, THEN, condition"""
        return Expr(otherwise.nodes + then.nodes + cond.nodes + [ExprNode(capi.EX_CASE, type)])

    def logical_not(self) -> "Expr":
        """NOT self (Analyzer::UOper kNOT over a BOOLEAN)"""
        return Expr(self.nodes + [ExprNode(capi.EX_NOT, capi.INT8)])

    def logical(self, op: int, other: "Expr", short_circuit: bool = False) -> "Expr":
        """self AND / OR other (capi.EX_AND / EX_OR over BOOLEANs).  short_circuit: the form the reference emits when an operand
        holds an unsafe division — `self` is evaluated first and `other` only where `self` does not decide"""
        return Expr(self.nodes + other.nodes + [ExprNode(op, capi.INT8, null_lit=1 if short_circuit else 0)])

    def is_null(self) -> "Expr":
        """self IS NULL as a NOT NULL BOOLEAN (Analyzer::UOper kISNULL)"""
        return Expr(self.nodes + [ExprNode(capi.EX_IS_NULL, capi.INT8)])

    def neg(self, type: int) -> "Expr":
        """-self (Analyzer::UOper kUMINUS); `type` = the operand's type"""
        return Expr(self.nodes + [ExprNode(capi.EX_UMINUS, type)])

    def _bin(self, op: int, other: "Expr", type: int) -> "Expr":
        return Expr(self.nodes + other.nodes + [ExprNode(op, type)])

    def add(self, other: "Expr", type: int) -> "Expr":
        return self._bin(capi.EX_ADD, other, type)

    def sub(self, other: "Expr", type: int) -> "Expr":
        return self._bin(capi.EX_SUB, other, type)

    def mul(self, other: "Expr", type: int) -> "Expr":
        return self._bin(capi.EX_MUL, other, type)

    def div(self, other: "Expr", type: int) -> "Expr":
        return self._bin(capi.EX_DIV, other, type)

    def mod(self, other: "Expr", type: int) -> "Expr":
        return self._bin(capi.EX_MOD, other, type)

    def with_range(self, r: ExpressionRange) -> "Expr":
        return Expr(self.nodes, r)

    def result(self, descs: Sequence[InputColDescriptor], prior: Sequence["Expr"] = ()) -> Tuple[int, bool]:
        """(type, nullable) of the value: the typing rules of plan.cpp lower_exprs.  `prior`: the plan's earlier expressions
        (a column index >= len(descs) reads the value of one of them)."""
        st: List[Tuple[int, bool]] = []
        for n in self.nodes:
            if n.op == capi.EX_COL and n.arg >= len(descs):
                j = n.arg - len(descs)
                st.append(prior[j].result(descs, prior[:j]))
            elif n.op == capi.EX_COL:
                d = descs[n.arg]
                lt = d.logical_type or (INT32 if d.encoding == capi.ENC_DICT else
                                        INT64 if d.encoding == capi.ENC_DATE_IN_DAYS else d.type)
                st.append((lt, bool(d.nullable)))
            elif n.op == capi.EX_LIT:
                st.append((n.type, bool(n.null_lit)))
            elif n.op == capi.EX_CAST:
                st[-1] = (n.type, st[-1][1])
            elif n.op in (capi.EX_NOT, capi.EX_UMINUS):
                st[-1] = (n.type, st[-1][1])
            elif n.op == capi.EX_IS_NULL:
                st[-1] = (capi.INT8, False)
            elif n.op == capi.EX_CASE:
                st.pop()
                t = st.pop()
                e = st.pop()
                st.append((n.type, t[1] or e[1]))
            else:
                b = st.pop()
                a = st.pop()
                st.append((n.type, a[1] or b[1]))
        assert len(st) == 1
        return st[0]

    def to_c(self) -> capi.Expr:
        e = capi.Expr()
        if len(self.nodes) > capi.MAX_EXPR_NODES:
            raise ValueError("expression too long")
        e.n_nodes = len(self.nodes)
        for i, n in enumerate(self.nodes):
            e.nodes[i] = capi.ExprNode(n.op, n.type, n.arg, int(n.null_lit), int(n.ilit), float(n.flit))
        e.range = self.range.to_c()
        return e


@dataclass
class RelAlgExecutionUnit:
    input_col_descs: List[InputColDescriptor]
    target_exprs: List[TargetExpr]
    simple_quals: List[Qual] = field(default_factory=list)
    groupby_exprs: List[int] = field(default_factory=list)
    inner_col_descs: List[InputColDescriptor] = field(default_factory=list)
    join_outer_col: int | Sequence[int] = -1   # one column, or a list for a composite key
    join_table: Optional["HashJoin"] = None
    join_kind: int = 0                          # capi.JOIN_INNER / JOIN_LEFT
    # ExecutionOptions / globals shaping the layout
    max_groups_buffer_entry_guess: int = 16384  # Execute.cpp:111
    bigint_count: bool = False
    output_columnar_hint: int = 0  # capi.OUTPUT_COLUMNAR: columnar result buffer (g_enable_columnar_output)
    num_tuples: int = 0   # rows of the input tables (0 = unknown / small): COUNT(*)-only group-bys
                          # get 4-byte slots while this is <= UINT32_MAX (pick_target_compact_width)
    # projected expressions: expression k is the virtual outer column len(input_col_descs) + k
    exprs: List[Expr] = field(default_factory=list)
    # Projection steps (every target capi.PROJECT, no groupby_exprs): LIMIT + OFFSET of a projection without ORDER BY
    # (RelAlgExecutionUnit::scan_limit); 0 = none, the buffer then has max_groups_buffer_entry_guess entries
    scan_limit: int = 0

    def col_type(self, c: int) -> int:
        """storage type of outer column c; for a virtual column the expression's result type"""
        n = len(self.input_col_descs)
        return self.input_col_descs[c].type if c < n else self.exprs[c - n].result(self.input_col_descs, self.exprs[:c - n])[0]

    def to_plan(self) -> capi.Plan:
        p = capi.Plan()
        p.abi_version = capi.ABI_VERSION
        if len(self.input_col_descs) > capi.MAX_COLS or len(self.inner_col_descs) > capi.MAX_COLS:
            raise ValueError("too many columns")
        p.n_cols = len(self.input_col_descs)
        for i, c in enumerate(self.input_col_descs):
            p.cols[i] = c.to_c()
            p.col_ranges[i] = c.range.to_c()
        p.n_inner_cols = len(self.inner_col_descs)
        for i, c in enumerate(self.inner_col_descs):
            p.inner_cols[i] = c.to_c()
            p.inner_col_ranges[i] = c.range.to_c()
        if len(self.simple_quals) > capi.MAX_QUALS:
            raise ValueError("too many quals")
        p.n_quals = len(self.simple_quals)
        for i, q in enumerate(self.simple_quals):
            is_fp = self.col_type(q.col) in (DOUBLE, capi.FLOAT)
            p.quals[i] = capi.Qual(q.col, q.op | (q.or_group << 8), 0 if is_fp else int(q.literal),
                                   float(q.literal) if is_fp else 0.0)
        if len(self.groupby_exprs) > capi.MAX_GROUP_COLS:
            raise ValueError("too many group-by columns")
        p.n_group_cols = len(self.groupby_exprs)
        for i, g in enumerate(self.groupby_exprs):
            p.group_cols[i] = g
        if len(self.target_exprs) > capi.MAX_TARGETS:
            raise ValueError("too many targets")
        p.n_targets = len(self.target_exprs)
        for i, t in enumerate(self.target_exprs):
            ct = capi.Target(t.agg, t.col, t.table, 0)
            if t.cond is not None:
                is_fp = self.col_type(t.cond.col) in (DOUBLE, capi.FLOAT)
                ct.cond = capi.Qual(t.cond.col, t.cond.op, 0 if is_fp else int(t.cond.literal),
                                    float(t.cond.literal) if is_fp else 0.0)
            p.targets[i] = ct
        jcols = list(self.join_outer_col) if isinstance(self.join_outer_col, (list, tuple)) else [self.join_outer_col]
        if len(jcols) > capi.MAX_GROUP_COLS:
            raise ValueError("too many join key columns")
        p.join_outer_col = jcols[0]
        p.n_join_cols = len(jcols) if len(jcols) > 1 else 0
        for i, c in enumerate(jcols):
            p.join_outer_cols[i] = c
        p.join_kind = self.join_kind
        p.join_table = self.join_table.handle if self.join_table is not None else None
        p.max_groups_buffer_entry_guess = self.max_groups_buffer_entry_guess
        p.bigint_count = int(self.bigint_count)
        p.output_columnar_hint = int(self.output_columnar_hint)
        p.num_tuples = int(self.num_tuples)
        if len(self.exprs) > capi.MAX_EXPRS or len(self.exprs) + len(self.input_col_descs) > capi.MAX_COLS:
            raise ValueError("too many expressions")
        p.n_exprs = len(self.exprs)
        for i, e in enumerate(self.exprs):
            p.exprs[i] = e.to_c()
        p.scan_limit = int(self.scan_limit)
        return p


@dataclass
class FetchResult:
    """Raw device column pointers per (fragment, column) + rows per fragment."""
    col_buffers: List[List[int]]      # [frag][col] -> device address
    num_rows: List[int]               # [frag]
    inner_col_buffers: List[int] = field(default_factory=list)
    inner_num_rows: int = 0
    device_id: int = 0
    keepalive: list = field(default_factory=list)  # owners of the memory (e.g. tensors)
    inner_version: int = 0   # generation of the inner columns' content (mi355q_inputs.inner_version): bump it when
                             # an inner column is rewritten in place, or its memory is reused for another column

    @staticmethod
    def from_tensors(frag_cols: Sequence[Sequence["object"]], inner_cols: Sequence["object"] = (),
                     device_id: int = 0) -> "FetchResult":
        """Convenience: torch tensors (already on the device) -> FetchResult."""
        bufs, rows, keep = [], [], []
        for cols in frag_cols:
            bufs.append([int(t.data_ptr()) for t in cols])
            rows.append(int(cols[0].numel()))
            keep.extend(cols)
        inner = [int(t.data_ptr()) for t in inner_cols]
        keep.extend(inner_cols)
        return FetchResult(bufs, rows, inner, int(inner_cols[0].numel()) if inner_cols else 0,
                           device_id, keep)

    def to_c(self, n_cols: int) -> Tuple[capi.Inputs, list]:
        n_frags = len(self.col_buffers)
        flat = (C.c_void_p * max(1, n_frags * n_cols))()
        for f, cols in enumerate(self.col_buffers):
            if len(cols) != n_cols:
                raise ValueError("fragment column count mismatch")
            for c, ptr in enumerate(cols):
                flat[f * n_cols + c] = ptr
        rows = (C.c_int64 * max(1, n_frags))(*self.num_rows)
        inner = (C.c_void_p * max(1, len(self.inner_col_buffers)))(*self.inner_col_buffers)
        inp = capi.Inputs()
        inp.device_id = self.device_id
        inp.n_frags = n_frags
        inp.col_buffers = C.cast(flat, C.POINTER(C.c_void_p))
        inp.num_rows = C.cast(rows, C.POINTER(C.c_int64))
        inp.inner_col_buffers = C.cast(inner, C.POINTER(C.c_void_p))
        inp.inner_num_rows = self.inner_num_rows
        inp.inner_version = int(self.inner_version)
        return inp, [flat, rows, inner]


class HashJoin:
    """Join hash table handle (HashJoin::getInstance, HashJoin.cpp:286): perfect when there is
    one key column and its range allows, keyed (baseline) otherwise; OneToOne first, rebuilt
    as OneToMany when a key repeats (one_to_many=1) — or OneToOne only (0, a duplicate is an
    error) / OneToMany straight away (2)."""

    def __init__(self, handle: int):
        self.handle = handle
        self._lib = capi.load_library()

    def invalidate_payload(self) -> None:
        """Drop the per-inner-column payloads the table caches (see FetchResult.inner_version)."""
        check(self._lib.mi355q_join_invalidate_payload(self.handle), "join_invalidate_payload")

    def payload_info(self) -> dict:
        b, ms, ver = C.c_int64(), C.c_float(), C.c_int64()
        check(self._lib.mi355q_join_payload_info(self.handle, C.byref(b), C.byref(ms), C.byref(ver)), "join_payload_info")
        return dict(bytes=b.value, build_ms=ms.value, inner_version=ver.value)

    @staticmethod
    def getInstance(key_buffer, num_rows: int, key_type, key_range: ExpressionRange,
                    key_nullable=False, device_id: int = 0, prefer_baseline: bool = False,
                    max_perfect_entries: int = 0, stream: int | None = None,
                    one_to_many: int = 0, keyed_entry_count: int = 0) -> "HashJoin":
        """key_buffer / key_type / key_nullable: scalars for one key column, equal-length lists
        for a composite key."""
        lib = capi.load_library()
        bufs = list(key_buffer) if isinstance(key_buffer, (list, tuple)) else [key_buffer]
        types = list(key_type) if isinstance(key_type, (list, tuple)) else [key_type]
        nulls = list(key_nullable) if isinstance(key_nullable, (list, tuple)) else [key_nullable] * len(bufs)
        assert len(bufs) == len(types) == len(nulls) <= capi.MAX_GROUP_COLS
        spec = capi.JoinSpec(device_id, types[0], int(nulls[0]), int(prefer_baseline),
                             bufs[0], num_rows, key_range.to_c(), max_perfect_entries)
        spec.n_keys = len(bufs)
        spec.one_to_many = one_to_many
        spec.keyed_entry_count = keyed_entry_count
        for i in range(1, len(bufs)):
            spec.more_key_types[i - 1] = types[i]
            spec.more_key_nullables[i - 1] = int(nulls[i])
            spec.more_key_buffers[i - 1] = bufs[i]
        out = C.c_void_p()
        check(lib.mi355q_join_build(C.byref(spec), stream, C.byref(out)), "join_build")
        return HashJoin(out.value)

    def info(self) -> dict:
        ht, ec, mn, mx = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
        ptr, nbytes, ms = C.c_void_p(), C.c_int64(), C.c_float()
        check(self._lib.mi355q_join_info(self.handle, C.byref(ht), C.byref(ec), C.byref(mn),
                                         C.byref(mx), C.byref(ptr), C.byref(nbytes),
                                         C.byref(ms)))
        kc, kw = C.c_int32(), C.c_int32()
        check(self._lib.mi355q_join_key_shape(self.handle, C.byref(kc), C.byref(kw)))
        return dict(hash_type=ht.value, entry_count=ec.value, min_key=mn.value, max_key=mx.value,
                    device_ptr=ptr.value, bytes=nbytes.value, build_ms=ms.value,
                    key_components=kc.value, component_width=kw.value)

    def free(self):
        if self.handle:
            self._lib.mi355q_join_free(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


def rows_to_arrow(q: capi.QMD, ival: np.ndarray, dval: np.ndarray, is_null: np.ndarray,
                  names: Optional[Sequence[str]] = None):
    """fetch_rows output -> pyarrow.Table (host-side; columns int64 / float64 with null masks)."""
    import pyarrow as pa
    cols, fields = [], []
    for t in range(q.n_targets):
        fp = bool(q.target_is_fp[t])
        vals = dval[:, t] if fp else ival[:, t]
        mask = is_null[:, t].astype(bool)
        cols.append(pa.array(vals, type=pa.float64() if fp else pa.int64(), mask=mask if mask.any() else None))
        fields.append(names[t] if names else f"target_{t}")
    return pa.table(cols, names=fields)


class ResultSet:
    """Owner of one ResultSetStorage buffer in HeavyDB row-wise layout, on the device."""

    def __init__(self, handle: int, report: Optional[capi.ExecReport] = None):
        self.handle = handle
        self.report = report
        self._lib = capi.load_library()
        self._rows = None
        self._cursor = 0

    # -- descriptor / storage
    def getQueryMemDesc(self) -> capi.QMD:
        q = capi.QMD()
        check(self._lib.mi355q_result_qmd(self.handle, C.byref(q)))
        return q

    def entryCount(self) -> int:
        return self.getQueryMemDesc().entry_count

    def device_ptr(self) -> int:
        return self._lib.mi355q_result_device_ptr(self.handle)

    def nbytes(self) -> int:
        return self._lib.mi355q_result_bytes(self.handle)

    def getStorage(self) -> np.ndarray:
        """The raw buffer (ResultSetStorage::buff_) copied to the host, as int64 quads
        shaped [entry_count, row_size/8] — or flat for a columnar descriptor (`columns()` splits it)."""
        q = self.getQueryMemDesc()
        buf = np.empty(self.nbytes() // 8, dtype=np.int64)
        check(self._lib.mi355q_result_copy_to_host(self.handle, buf.ctypes.data, buf.nbytes))
        return buf if q.output_columnar else buf.reshape(q.entry_count, q.row_size // 8)

    def columns(self) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """A columnar buffer (output_columnar_) split into its (group columns, slot columns), each of
        entry_count elements: int64 keys, int64 or int32 slots."""
        q = self.getQueryMemDesc()
        if not q.output_columnar:
            raise ValueError("row-wise result: use getStorage()")
        raw = self.getStorage().view(np.int8)
        n = q.entry_count
        keys, slots = [], []
        for g in range(0 if q.keyless else q.group_col_count):
            o = self._lib.mi355q_qmd_group_col_offset(C.byref(q), g)
            keys.append(raw[o:o + 8 * n].view(np.int64))
        for s in range(q.slot_count):
            o = self._lib.mi355q_qmd_slot_col_offset(C.byref(q), s)
            w = q.slot_bytes[s] or q.slot_width
            slots.append(raw[o:o + w * n].view({1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[w]))
        return keys, slots

    # -- iteration
    def sort(self, target_idx: int, top_n: int, out_rows_dev: int, desc: bool = True,
             nulls_first: bool = False) -> int:
        """ORDER BY target [DESC] LIMIT top_n on the device (ResultSet::sort with one order
        entry, ResultSet.h:278): the best rows land, in order, in the caller's device buffer
        (top_n rows of this layout); returns how many were written."""
        n = C.c_int64()
        check(self._lib.mi355q_result_topk(self.handle, target_idx, int(desc), int(nulls_first), top_n,
                                           out_rows_dev, C.byref(n), None), "result_topk")
        return n.value

    def sort_by(self, order_entries, out_rows_dev: int, limit: int = 0, offset: int = 0) -> int:
        """ORDER BY several targets [LIMIT limit OFFSET offset] on the device (ResultSet::sort with a
        list of Analyzer::OrderEntry).  order_entries: [(target_idx, desc, nulls_first), ...], most
        significant first; limit 0 = every live row.  The rows land, in order, in the caller's device
        buffer (whole rows of this layout); returns how many were written."""
        oe = (capi.OrderEntry * len(order_entries))()
        for i, (t, desc, nf) in enumerate(order_entries):
            oe[i].target_idx, oe[i].descending, oe[i].nulls_first = int(t), int(bool(desc)), int(bool(nf))
        n = C.c_int64()
        check(self._lib.mi355q_result_sort(self.handle, oe, len(order_entries), int(limit), int(offset),
                                           out_rows_dev, C.byref(n), None), "result_sort")
        return n.value

    def rowCount(self) -> int:
        return self._lib.mi355q_result_row_count(self.handle)

    def totalMatched(self) -> int:
        """Projection results: rows that passed the quals (the kernel's total_matched); > entryCount() when a
        scan_limit cut the output.  -1 for any other result."""
        return self._lib.mi355q_result_total_matched(self.handle)

    def append(self, that: "ResultSet", stream: int | None = None) -> None:
        """ResultSet::append (ResultSet.cpp:307-335): a Projection result followed by another one's rows — how the
        reference puts the devices' / kernels' projections together (Executor::resultsUnion)."""
        check(self._lib.mi355q_result_append(self.handle, that.handle, C.c_void_p(stream or 0)), "result_append")

    def fetch(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """All rows in entry order: (ival[n,t], dval[n,t], is_null[n,t])."""
        q = self.getQueryMemDesc()
        n = self.rowCount()
        nt = q.n_targets
        ival = np.zeros((max(n, 1), nt), dtype=np.int64)
        dval = np.zeros((max(n, 1), nt), dtype=np.float64)
        nul = np.zeros((max(n, 1), nt), dtype=np.int8)
        got = C.c_int64()
        check(self._lib.mi355q_result_fetch_rows(self.handle, n, ival.ctypes.data,
                                                 dval.ctypes.data, nul.ctypes.data,
                                                 C.byref(got)))
        k = got.value
        return ival[:k], dval[:k], nul[:k]

    def to_columns(self, torch):
        """ColumnarResults on the device (ColumnarResults.cpp materializeAllColumnsGroupBy): one
        dense int64 tensor per target holding the non-empty entries in entry order; floating-point
        targets carry double bits (`.view(torch.float64)`), SQL NULL is the inline sentinel."""
        q = self.getQueryMemDesc()
        n = self.rowCount()
        cols = [torch.empty(max(n, 1), dtype=torch.int64, device="cuda") for _ in range(q.n_targets)]
        ptrs = (C.c_void_p * q.n_targets)(*[int(c.data_ptr()) for c in cols])
        got = C.c_int64()
        check(self._lib.mi355q_result_to_columns(self.handle, ptrs, q.n_targets, C.byref(got), None),
              "result_to_columns")
        return [c[:got.value] for c in cols], got.value

    def to_arrow(self, names: Optional[Sequence[str]] = None):
        """The rows as a pyarrow.Table — what ArrowResultSetConverter::convertToArrow
        (QueryEngine/ArrowResultSetConverter.cpp) produces from a ResultSet: one column per
        target, BIGINT -> int64, DOUBLE -> float64, SQL NULLs as validity bits."""
        return rows_to_arrow(self.getQueryMemDesc(), *self.fetch(), names=names)

    def to_arrow_native(self, names: Optional[Sequence[str]] = None):
        """The same table through the library's own Arrow C Data Interface export
        (mi355q_result_export_arrow: device-side ColumnarResults -> host buffers -> ArrowArray /
        ArrowSchema structs), imported zero-copy by pyarrow."""
        import pyarrow as pa
        c_schema = C.create_string_buffer(72)   # struct ArrowSchema: 9 words (Arrow C Data Interface)
        c_array = C.create_string_buffer(80)    # struct ArrowArray: 10 words
        ps, pa_ = C.addressof(c_schema), C.addressof(c_array)
        arr = None
        if names is not None:
            keep = [n.encode() for n in names]
            arr = (C.c_char_p * len(keep))(*keep)
        check(self._lib.mi355q_result_export_arrow(self.handle, arr, ps, pa_, None), "result_export_arrow")
        batch = pa.RecordBatch._import_from_c(pa_, ps)
        return pa.Table.from_batches([batch])

    def getNextRow(self) -> list:
        """One row of target values (None = SQL NULL); [] when exhausted."""
        if self._rows is None:
            q = self.getQueryMemDesc()
            ival, dval, nul = self.fetch()
            fp = [bool(q.target_is_fp[t]) for t in range(q.n_targets)]
            self._rows = [
                [None if nul[r, t] else (float(dval[r, t]) if fp[t] else int(ival[r, t]))
                 for t in range(q.n_targets)] for r in range(ival.shape[0])]
        if self._cursor >= len(self._rows):
            return []
        row = self._rows[self._cursor]
        self._cursor += 1
        return row

    def reduce(self, that: "ResultSet", stream: int | None = None) -> None:
        check(self._lib.mi355q_result_reduce(self.handle, that.handle, stream), "reduce")
        self._rows = None

    def free(self):
        if self.handle:
            self._lib.mi355q_result_free(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


class Executor:
    """One per device.  executeWorkUnit = compile-free plan -> kernel family selection ->
    launch -> ResultSet on the device."""

    def __init__(self, device_id: int = 0):
        self.device_id = device_id
        self._lib = capi.load_library()

    def initQueryMemoryDescriptor(self, ra_exe_unit: RelAlgExecutionUnit) -> capi.QMD:
        q = capi.QMD()
        plan = ra_exe_unit.to_plan()
        check(self._lib.mi355q_qmd_init(C.byref(plan), C.byref(q)), "qmd_init")
        return q

    def executeWorkUnit(self, ra_exe_unit: RelAlgExecutionUnit, fetch_result: FetchResult,
                        stream: int | None = None, out_buffer: int | None = None,
                        force_generic: bool = False, kernel_variant: int = 0,
                        scratch_bytes: int = 0, allow_retry: bool = True, pass_rows: int = 0,
                        probe_keyed_passes: int = 0, flags: int = 0, tune_blocks_per_cu: int = 0,
                        tune_cus: int = 0, tune_overlap_cus: int = 0) -> ResultSet:
        """Runs the step.  A negative code (ran out of group slots) doubles the baseline
        table and retries, as RelAlgExecutor::executeWorkUnit does after its cardinality
        estimation (RelAlgExecutor.cpp:4143-4145, :4194-4231)."""
        guess = ra_exe_unit.max_groups_buffer_entry_guess
        try:
            for _ in range(24):
                plan = ra_exe_unit.to_plan()
                inp, keep = fetch_result.to_c(plan.n_cols)
                opts = capi.ExecOptions()
                opts.stream = stream
                opts.out_buffer = out_buffer
                opts.force_generic = int(force_generic)
                opts.kernel_variant = kernel_variant
                opts.scratch_bytes = scratch_bytes
                opts.pass_rows = pass_rows
                opts.probe_keyed_passes = probe_keyed_passes
                opts.flags = flags
                opts.tune_blocks_per_cu = tune_blocks_per_cu
                opts.tune_cus = tune_cus
                opts.tune_overlap_cus = tune_overlap_cus
                out = C.c_void_p()
                rep = capi.ExecReport()
                code = self._lib.mi355q_execute(C.byref(plan), C.byref(inp), C.byref(opts),
                                                C.byref(out), C.byref(rep))
                del keep
                if code == 0:
                    return ResultSet(out.value, rep)
                # only a baseline-hash table grows with the guess; on a perfect-hash layout code 3
                # means a key outside its declared range and a retry would repeat the same step
                grows = code < 0 or (code == capi.ERR_OUT_OF_SLOTS and
                                     self.initQueryMemoryDescriptor(ra_exe_unit).desc_type
                                     == capi.GROUP_BY_BASELINE_HASH)
                # (a table never needs more entries than twice the rows it is fed: past that, growing cannot help and
                # the code is reported instead of being retried with ever larger tables)
                fed = 2 * max(sum(fetch_result.num_rows), 8192)
                if (grows and allow_retry and out_buffer is None and ra_exe_unit.groupby_exprs
                        and ra_exe_unit.max_groups_buffer_entry_guess < fed):
                    ra_exe_unit.max_groups_buffer_entry_guess *= 2
                    continue
                raise capi.Mi355qError(code, "execute")
            raise capi.Mi355qError(capi.ERR_OUT_OF_SLOTS, "execute (retries exhausted)")
        finally:
            if not allow_retry:
                ra_exe_unit.max_groups_buffer_entry_guess = guess


    def _opts(self, stream, out_buffer, force_generic, kernel_variant, scratch_bytes, pass_rows=0,
              probe_keyed_passes=0, flags=0, tune_blocks_per_cu=0) -> capi.ExecOptions:
        opts = capi.ExecOptions()
        opts.stream = stream
        opts.out_buffer = out_buffer
        opts.force_generic = int(force_generic)
        opts.kernel_variant = kernel_variant
        opts.scratch_bytes = scratch_bytes
        opts.pass_rows = pass_rows
        opts.probe_keyed_passes = probe_keyed_passes
        opts.flags = flags
        opts.tune_blocks_per_cu = tune_blocks_per_cu
        return opts

    def executeWorkUnitAsync(self, ra_exe_unit: RelAlgExecutionUnit, fetch_result: FetchResult,
                             stream: int | None = None, out_buffer: int | None = None, force_generic: bool = False,
                             kernel_variant: int = 0, scratch_bytes: int = 0, flags: int = 0):
        """mi355q_execute_async: the step is enqueued and the call returns; the ResultSet may be handed to
        stream-ordered consumers on the same stream at once, `wait()` on the returned PendingStep gives the
        error code (raised) and the report.  No retry ladder: an out-of-slots table is the caller's to grow."""
        plan = ra_exe_unit.to_plan()
        inp, keep = fetch_result.to_c(plan.n_cols)
        opts = self._opts(stream, out_buffer, force_generic, kernel_variant, scratch_bytes, flags=flags)
        out, pend = C.c_void_p(), C.c_void_p()
        code = self._lib.mi355q_execute_async(C.byref(plan), C.byref(inp), C.byref(opts), C.byref(out), C.byref(pend))
        del keep
        if code and not pend.value:
            raise capi.Mi355qError(code, "execute_async")
        rs = ResultSet(out.value, capi.ExecReport()) if out.value else None
        return rs, PendingStep(self._lib, pend.value, rs)

    def reserveWorkspace(self, ra_exe_unit: RelAlgExecutionUnit, fetch_result: FetchResult, scratch_bytes: int = 0,
                         kernel_variant: int = 0) -> int:
        """mi355q_reserve_workspace: allocate what the step will need from the per-device workspace now (the
        cold first call of a 10 B-row GROUP BY otherwise spends 1 - 2 s in hipMalloc).  Returns the bytes held."""
        plan = ra_exe_unit.to_plan()
        inp, keep = fetch_result.to_c(plan.n_cols)
        opts = self._opts(None, None, False, kernel_variant, scratch_bytes)
        got = C.c_int64()
        check(self._lib.mi355q_reserve_workspace(C.byref(plan), C.byref(inp), C.byref(opts), C.byref(got)),
              "reserve_workspace")
        del keep
        return got.value

    def explain(self, ra_exe_unit: RelAlgExecutionUnit, frag_rows: Sequence[int], kernel_variant: int = 0,
                inner_rows: int = 0, flags: int = 0) -> str:
        """mi355q_explain: the route a step of this plan would take over fragments of these sizes (ExecutionOptions::
        just_explain of the reference shows the generated kernel; this shows which members of the fixed family run).
        No column data is needed: only the shape of the input decides."""
        plan = ra_exe_unit.to_plan()
        fr = FetchResult([[0] * plan.n_cols for _ in frag_rows], list(frag_rows), [0] * 8 if inner_rows else [], inner_rows)
        inp, keep = fr.to_c(plan.n_cols)
        opts = self._opts(None, None, False, kernel_variant, 0, flags=flags)
        buf = C.create_string_buffer(512)
        got = C.c_int64()
        check(self._lib.mi355q_explain(C.byref(plan), C.byref(inp), C.byref(opts), buf, 512, C.byref(got)), "explain")
        del keep
        return buf.value.decode()


class PendingStep:
    """A step enqueued by Executor.executeWorkUnitAsync (mi355q_pending)."""

    def __init__(self, lib, handle, result_set):
        self._lib, self.handle, self.result_set = lib, handle, result_set

    recomputed = False  # mi355q_wait answered MI355Q_STEP_RECOMPUTED: stream-ordered consumers must be redone

    def wait(self):
        """Blocks until the step has finished; raises on an error code; returns the ResultSet (with its report).
        `self.recomputed` tells whether the step was re-run inside the wait (MI355Q_STEP_RECOMPUTED): whatever was
        enqueued behind the first launches read the table of the abandoned attempt."""
        if self.handle is None:
            return self.result_set
        rep = capi.ExecReport()
        code = self._lib.mi355q_wait(self.handle, C.byref(rep))
        self.handle = None
        if code == capi.STEP_RECOMPUTED:
            self.recomputed = True
            code = 0
        if code:
            raise capi.Mi355qError(code, "wait")
        if self.result_set is not None:
            self.result_set.report = rep
        return self.result_set


def generate_column(dst_ptr: int, n_rows: int, kind: int, seed: int, a: int = 0, b: int = 0,
                    c: int = 0, a_f: float = 0.0, null_every: int = 0, row_offset: int = 0,
                    device_id: int = 0, stream: int | None = None) -> None:
    """Device-side synthetic column (same splitmix64 stream as the oracle's generator)."""
    lib = capi.load_library()
    check(lib.mi355q_generate_column(device_id, dst_ptr, n_rows, row_offset, kind, seed, a, b, c,
                                     a_f, null_every, stream), "generate_column")

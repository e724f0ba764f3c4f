"""ctypes wrapper of oracle/liboracle.so (and oracle/_ref/libref_runtime.so when built).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (heavydb_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

from heavydb_amd import capi  # struct definitions only (include/mi355q.h mirror)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_LIB = os.path.join(HERE, "_ref", "libref_runtime.so")


def build(force: bool = False) -> None:
    """Compile the oracle (and, when /root/reference is present, oracle/_ref)."""
    targets = ["all"]
    if os.path.isdir("/root/reference/QueryEngine"):
        targets.append("ref")
    if force:
        subprocess.run(["make", "-C", HERE, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", HERE] + targets, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "oracle.cpp")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            build()
        l = C.CDLL(LIB)
        P = C.POINTER
        l.orc_murmur3.restype = C.c_uint32
        l.orc_murmur3.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        l.orc_murmur1.restype = C.c_uint32
        l.orc_murmur1.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        l.orc_qmd_init.restype = C.c_int32
        l.orc_qmd_init.argtypes = [P(capi.Plan), P(capi.QMD)]
        l.orc_init_buffer.restype = None
        l.orc_init_buffer.argtypes = [P(capi.QMD), C.c_void_p]
        l.orc_get_group_value_slot.restype = C.c_int64
        l.orc_get_group_value_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_uint32,
                                               C.c_uint32]
        l.orc_get_group_value_fast_slot.restype = C.c_int64
        l.orc_get_group_value_fast_slot.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint32]
        l.orc_get_group_value_fast_bucket_slot.restype = C.c_int64
        l.orc_get_group_value_fast_bucket_slot.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                                           C.c_uint32]
        l.orc_get_group_value_n_slot.restype = C.c_int64
        l.orc_get_group_value_n_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                 C.c_uint32, C.c_uint32]
        l.orc_perfect_hash_slot.restype = C.c_int64
        l.orc_perfect_hash_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        l.orc_decode_col.restype = C.c_int64
        l.orc_decode_col.argtypes = [P(capi.ColDesc), C.c_void_p, C.c_int64]
        l.orc_join_build.restype = C.c_void_p
        l.orc_join_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                     C.c_int64, C.c_int, C.c_int64, P(C.c_int32)]
        l.orc_join_build_n.restype = C.c_void_p
        l.orc_join_build_n.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64,
                                       C.c_int64, C.c_int, C.c_int64, C.c_int32, C.c_int64, P(C.c_int32)]
        l.orc_join_matches.restype = C.c_int32
        l.orc_join_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        l.orc_join_shape.restype = C.c_int32
        l.orc_join_shape.argtypes = [C.c_void_p, P(C.c_int32), P(C.c_int32), P(C.c_int64)]
        l.orc_join_free.restype = None
        l.orc_join_free.argtypes = [C.c_void_p]
        l.orc_join_probe.restype = C.c_int64
        l.orc_join_probe.argtypes = [C.c_void_p, C.c_int64]
        l.orc_join_info.restype = C.c_int32
        l.orc_join_info.argtypes = [C.c_void_p, P(C.c_int32), P(C.c_int64)]
        l.orc_join_buffer.restype = C.c_void_p
        l.orc_join_buffer.argtypes = [C.c_void_p]
        l.orc_execute.restype = C.c_int32
        l.orc_execute.argtypes = [P(capi.Plan), P(capi.Inputs), C.c_void_p, C.c_int32,
                                  C.c_void_p, P(capi.QMD)]
        for fn in ("orc_buffer_bytes", "orc_col_group_off", "orc_col_slot_off"):
            getattr(l, fn).restype = C.c_int64
        l.orc_buffer_bytes.argtypes = [P(capi.QMD)]
        l.orc_get_columnar_group_bin_offset.restype = C.c_uint32
        l.orc_get_columnar_group_bin_offset.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
        l.orc_get_group_value_columnar_slot.restype = C.c_int32
        l.orc_get_group_value_columnar_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        l.orc_col_group_off.argtypes = [P(capi.QMD), C.c_int32]
        l.orc_col_slot_off.argtypes = [P(capi.QMD), C.c_int32]
        l.orc_reduce.restype = C.c_int32
        l.orc_reduce.argtypes = [P(capi.QMD), C.c_void_p, C.c_void_p]
        l.orc_row_count.restype = C.c_int64
        l.orc_row_count.argtypes = [P(capi.QMD), C.c_void_p]
        l.orc_fetch_rows.restype = C.c_int32
        l.orc_fetch_rows.argtypes = [P(capi.QMD), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, P(C.c_int64)]
        l.orc_last_total_matched.restype = C.c_int64
        l.orc_last_total_matched.argtypes = []
        l.orc_splitmix64.restype = C.c_uint64
        l.orc_splitmix64.argtypes = [C.c_uint64]
        l.orc_generate_column.restype = C.c_int32
        l.orc_generate_column.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                          C.c_uint64, C.c_int64, C.c_int64, C.c_int64,
                                          C.c_double, C.c_int32]
        _lib = l
    return _lib


def ref_lib() -> Optional[C.CDLL]:
    """The reference's own runtime functions, when oracle/_ref has been built."""
    if not os.path.exists(REF_LIB):
        return None
    r = C.CDLL(REF_LIB)
    r.MurmurHash3.restype = C.c_uint32
    r.MurmurHash3.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    r.MurmurHash1.restype = C.c_uint32
    r.MurmurHash1.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    r.get_group_value.restype = C.c_void_p
    r.get_group_value.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                  C.c_uint32]
    r.get_group_value_fast.restype = C.c_void_p
    r.get_group_value_fast.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint32]
    r.hash_join_idx.restype = C.c_int64
    r.hash_join_idx.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64]
    r.baseline_hash_join_idx_64.restype = C.c_int64
    r.baseline_hash_join_idx_64.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    r.fixed_width_int_decode.restype = C.c_int64
    r.fixed_width_int_decode.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    r.fixed_width_double_decode.restype = C.c_double
    r.fixed_width_double_decode.argtypes = [C.c_void_p, C.c_int64]
    r.get_composite_key_index_64.restype = C.c_int64
    r.get_composite_key_index_64.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    r.get_composite_key_index_32.restype = C.c_int64
    r.get_composite_key_index_32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    r.baseline_hash_join_idx_32.restype = C.c_int64
    r.baseline_hash_join_idx_32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    r.fixed_width_unsigned_decode.restype = C.c_int64
    r.fixed_width_unsigned_decode.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    r.fixed_width_small_date_decode.restype = C.c_int64
    r.fixed_width_small_date_decode.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64]
    return r


NP_DTYPE = {capi.INT8: np.int8, capi.INT16: np.int16, capi.INT32: np.int32,
            capi.INT64: np.int64, capi.DOUBLE: np.float64, capi.FLOAT: np.float32}


def murmur3(data: bytes, seed: int = 0) -> int:
    return lib().orc_murmur3(data, len(data), seed)


def murmur1(data: bytes, seed: int = 0) -> int:
    return lib().orc_murmur1(data, len(data), seed)


def qmd_init(plan: capi.Plan) -> capi.QMD:
    q = capi.QMD()
    code = lib().orc_qmd_init(C.byref(plan), C.byref(q))
    if code:
        raise capi.Mi355qError(code, "oracle qmd_init")
    return q


class OracleJoin:
    """keys / key_type / nullable: one inner key column, or equal-length lists for a composite
    key.  one_to_many: 0 OneToOne only, 1 rebuild as OneToMany on a duplicate, 2 OneToMany."""

    def __init__(self, keys, key_type, min_key: int, max_key: int,
                 nullable=False, prefer_baseline: bool = False,
                 max_perfect_entries: int = 0, one_to_many: int = 0, keyed_entry_count: int = 0):
        cols = list(keys) if isinstance(keys, (list, tuple)) else [keys]
        types = list(key_type) if isinstance(key_type, (list, tuple)) else [key_type]
        nulls = list(nullable) if isinstance(nullable, (list, tuple)) else [nullable] * len(cols)
        self.keys = [np.ascontiguousarray(k, dtype=NP_DTYPE[t]) for k, t in zip(cols, types)]
        n = len(self.keys)
        ptrs = (C.c_void_p * n)(*[k.ctypes.data for k in self.keys])
        err = C.c_int32()
        self.handle = lib().orc_join_build_n(ptrs, (C.c_int32 * n)(*types), (C.c_int32 * n)(*[int(x) for x in nulls]),
                                             n, len(self.keys[0]), min_key, max_key, int(prefer_baseline),
                                             max_perfect_entries, one_to_many, keyed_entry_count, C.byref(err))
        self.err = err.value
        if not self.handle:
            raise capi.Mi355qError(self.err, "oracle join build")

    def probe(self, key: int) -> int:
        return lib().orc_join_probe(self.handle, key)

    def matches(self, key) -> list:
        """Row ids of the matching set of a (composite) key."""
        k = np.atleast_1d(np.asarray(key, dtype=np.int64))
        ids = np.zeros(1 << 16, dtype=np.int32)
        n = lib().orc_join_matches(self.handle, k.ctypes.data, ids.ctypes.data, len(ids))
        return [int(x) for x in ids[:n]]

    def info(self):
        ht, ec = C.c_int32(), C.c_int64()
        lib().orc_join_info(self.handle, C.byref(ht), C.byref(ec))
        nk, w, nb = C.c_int32(), C.c_int32(), C.c_int64()
        lib().orc_join_shape(self.handle, C.byref(nk), C.byref(w), C.byref(nb))
        self._shape = (nk.value, w.value, nb.value)
        return dict(hash_type=ht.value, entry_count=ec.value)

    def shape(self):
        self.info()
        return dict(key_components=self._shape[0], component_width=self._shape[1], bytes=self._shape[2])

    def raw(self) -> np.ndarray:
        """The whole hash join buffer as bytes."""
        nb = self.shape()["bytes"]
        ptr = lib().orc_join_buffer(self.handle)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nb,)).copy()

    def buffer(self) -> np.ndarray:
        """OneToOne tables as arrays: int32[entries] (perfect) or [entries, components + 1]."""
        i = self.info()
        sh = self.shape()
        raw = self.raw()
        if i["hash_type"] == 0:
            return raw.view(np.int32).copy()
        dt = np.int32 if sh["component_width"] == 4 else np.int64
        stride = sh["key_components"] + (1 if i["hash_type"] == 1 else 0)
        return raw[:i["entry_count"] * stride * sh["component_width"]].view(dt).reshape(i["entry_count"], stride).copy()

    def __del__(self):
        try:
            if self.handle:
                lib().orc_join_free(self.handle)
        except Exception:
            pass


def keyless_key_is_null_aware(q: capi.QMD) -> bool:
    """get_keyless_info only guards MAX against NULL-aware key targets (GroupByAndAggregate.cpp:489-648)."""
    if not q.keyless:
        return False
    for t in range(q.n_targets):
        if q.target_slot[t] in (q.idx_target_as_key, q.idx_target_as_key - 1) and q.target_skip_null[t]:
            if q.target_slot[t] == q.idx_target_as_key or q.target_agg[t] == capi.AVG:
                return True
    return False


def execute(plan: capi.Plan, frag_cols: Sequence[Sequence[np.ndarray]],
            inner_cols: Sequence[np.ndarray] = (), join: Optional[OracleJoin] = None,
            n_threads: int = 1) -> Tuple[capi.QMD, np.ndarray, int]:
    """Run the step on host numpy columns.  Returns (qmd, buffer[entry_count,row_quads],
    code)."""
    q = qmd_init(plan)
    if n_threads > 1 and keyless_key_is_null_aware(q):
        # a keyless layout whose "key" slot is a NULL-aware aggregate: an all-NULL group of a partial
        # buffer looks empty and ResultSetStorage::reduce drops its other slots (reference behaviour,
        # DESIGN section 2) — the merged result depends on how rows were dealt to kernels.  One kernel is
        # the reading that does not.
        n_threads = 1
    n_frags = len(frag_cols)
    n_cols = plan.n_cols
    flat = (C.c_void_p * max(1, n_frags * n_cols))()
    keep = []
    rows = (C.c_int64 * max(1, n_frags))()
    for f, cols in enumerate(frag_cols):
        assert len(cols) == n_cols
        for c, a in enumerate(cols):
            a = np.ascontiguousarray(a, dtype=NP_DTYPE[plan.cols[c].type])
            keep.append(a)
            flat[f * n_cols + c] = a.ctypes.data
        rows[f] = len(cols[0]) if cols else 0
    inner = (C.c_void_p * max(1, len(inner_cols)))()
    for c, a in enumerate(inner_cols):
        a = np.ascontiguousarray(a, dtype=NP_DTYPE[plan.inner_cols[c].type])
        keep.append(a)
        inner[c] = a.ctypes.data
    inp = capi.Inputs()
    inp.device_id = -1
    inp.n_frags = n_frags
    inp.col_buffers = C.cast(flat, C.POINTER(C.c_void_p))
    inp.num_rows = C.cast(rows, C.POINTER(C.c_int64))
    inp.inner_col_buffers = C.cast(inner, C.POINTER(C.c_void_p))
    inp.inner_num_rows = len(inner_cols[0]) if len(inner_cols) else 0
    buf = _alloc(q)
    out_q = capi.QMD()
    code = lib().orc_execute(C.byref(plan), C.byref(inp), join.handle if join else None,
                             n_threads, buf.ctypes.data, C.byref(out_q))
    return out_q, buf, code


def last_total_matched() -> int:
    """total_matched of this thread's last Projection run through execute()"""
    return lib().orc_last_total_matched()


def buffer_bytes(q: capi.QMD) -> int:
    return lib().orc_buffer_bytes(C.byref(q))


def col_group_off(q: capi.QMD, g: int) -> int:
    return lib().orc_col_group_off(C.byref(q), g)


def col_slot_off(q: capi.QMD, s: int) -> int:
    return lib().orc_col_slot_off(C.byref(q), s)


def _alloc(q: capi.QMD) -> np.ndarray:
    """[entry_count, row quads] for a row-wise descriptor, flat int64 for a columnar one."""
    if q.output_columnar:
        return np.zeros(buffer_bytes(q) // 8, dtype=np.int64)
    return np.empty((q.entry_count, q.row_size // 8), dtype=np.int64)


def init_buffer(q: capi.QMD) -> np.ndarray:
    buf = _alloc(q)
    lib().orc_init_buffer(C.byref(q), buf.ctypes.data)
    return buf


def reduce(q: capi.QMD, this_buf: np.ndarray, that_buf: np.ndarray) -> int:
    assert this_buf.flags.c_contiguous and that_buf.flags.c_contiguous
    return lib().orc_reduce(C.byref(q), this_buf.ctypes.data, that_buf.ctypes.data)


def row_count(q: capi.QMD, buf: np.ndarray) -> int:
    return lib().orc_row_count(C.byref(q), buf.ctypes.data)


def fetch_rows(q: capi.QMD, buf: np.ndarray):
    n = row_count(q, buf)
    nt = q.n_targets
    ival = np.zeros((max(n, 1), nt), dtype=np.int64)
    dval = np.zeros((max(n, 1), nt), dtype=np.float64)
    nul = np.zeros((max(n, 1), nt), dtype=np.int8)
    got = C.c_int64()
    lib().orc_fetch_rows(C.byref(q), buf.ctypes.data, n, ival.ctypes.data, dval.ctypes.data,
                         nul.ctypes.data, C.byref(got))
    return ival[:got.value], dval[:got.value], nul[:got.value]


def generate_column(n_rows: int, kind: int, seed: int, a: int = 0, b: int = 0, c: int = 0,
                    a_f: float = 0.0, null_every: int = 0, row_offset: int = 0) -> np.ndarray:
    dt = {capi.GEN_I32_UNIFORM31: np.int32, capi.GEN_I32_MOD: np.int32,
          capi.GEN_I64_MOD: np.int64, capi.GEN_I64_MOD_MUL: np.int64,
          capi.GEN_F64_UNIT: np.float64}[kind]
    out = np.empty(n_rows, dtype=dt)
    code = lib().orc_generate_column(out.ctypes.data, n_rows, row_offset, kind, seed, a, b, c,
                                     a_f, null_every)
    assert code == 0
    return out


class GenSpec(C.Structure):
    """One synthetic column of a streamed run (orc_gen_spec)."""
    _fields_ = [("kind", C.c_int32), ("null_every", C.c_int32), ("seed", C.c_uint64),
                ("a", C.c_int64), ("b", C.c_int64), ("c", C.c_int64), ("a_f", C.c_double)]


def host_threads_for_tables(table_bytes: int, want: Optional[int] = None, reserve: float = 0.5) -> int:
    """How many kernels (one private output buffer each) this host can run at once: the core count,
    bounded by `reserve` of MemAvailable divided by one buffer."""
    n = want or os.cpu_count() or 1
    try:
        with open("/proc/meminfo") as f:
            avail_kb = next(int(l.split()[1]) for l in f if l.startswith("MemAvailable"))
        n = min(n, max(1, int(avail_kb * 1024 * reserve) // max(table_bytes, 1)))
    except Exception:
        pass
    return max(1, n)


def execute_streamed(plan: capi.Plan, gens: Sequence[tuple], total_rows: int, frag_rows: int = 32_000_000,
                     block_rows: int = 1 << 20, inner_cols: Sequence[np.ndarray] = (),
                     join: Optional[OracleJoin] = None, n_threads: int = 1, reduce_threads: int = 1,
                     row_offset: int = 0):
    """The step over a GENERATED table (BASELINE-size runs): every fragment is produced inside the
    kernel thread that scans it with the generator of generate_column.  gens: per column
    (kind, seed, a, b, c, a_f[, null_every]).  Returns (qmd, buffer, code, timing) with timing =
    dict(init_s, kernels_s, generate_s, reduce_s)."""
    l = lib()
    l.orc_execute_streamed.restype = C.c_int32
    l.orc_execute_streamed.argtypes = [C.POINTER(capi.Plan), C.POINTER(GenSpec), C.c_int64, C.c_int64, C.c_int64,
                                       C.c_int64, C.POINTER(C.c_void_p), C.c_int64, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p, C.POINTER(capi.QMD), C.POINTER(C.c_double)]
    assert len(gens) == plan.n_cols
    g = (GenSpec * len(gens))()
    for i, spec in enumerate(gens):
        kind, seed, a, b, c, a_f = spec[:6]
        g[i].kind, g[i].seed, g[i].a, g[i].b, g[i].c, g[i].a_f = kind, seed, a, b, c, a_f
        g[i].null_every = spec[6] if len(spec) > 6 else 0
    q = qmd_init(plan)
    keep = []
    inner = (C.c_void_p * max(1, len(inner_cols)))()
    for c_, a_ in enumerate(inner_cols):
        a_ = np.ascontiguousarray(a_, dtype=NP_DTYPE[plan.inner_cols[c_].type])
        keep.append(a_)
        inner[c_] = a_.ctypes.data
    buf = _alloc(q)
    out_q = capi.QMD()
    timing = (C.c_double * 4)()
    code = l.orc_execute_streamed(C.byref(plan), g, total_rows, row_offset, frag_rows, block_rows,
                                  C.cast(inner, C.POINTER(C.c_void_p)), len(inner_cols[0]) if len(inner_cols) else 0,
                                  join.handle if join else None, n_threads, reduce_threads, buf.ctypes.data,
                                  C.byref(out_q), timing)
    return out_q, buf, code, dict(init_s=timing[0], kernels_s=timing[1], generate_s=timing[2], reduce_s=timing[3])


def sort(q: capi.QMD, buf: np.ndarray, order_entries, limit: int = 0, offset: int = 0) -> np.ndarray:
    """ResultSet::sort: entry indices of the live rows ordered by [(target_idx, desc, nulls_first), ...]."""
    l = lib()
    l.orc_sort.restype = C.c_int64
    l.orc_sort.argtypes = [C.POINTER(capi.QMD), C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]
    oe = (capi.OrderEntry * len(order_entries))()
    for i, (t, desc, nf) in enumerate(order_entries):
        oe[i].target_idx, oe[i].descending, oe[i].nulls_first = int(t), int(bool(desc)), int(bool(nf))
    buf = np.ascontiguousarray(buf)
    out = np.zeros(max(int(q.entry_count), 1), dtype=np.int64)
    n = l.orc_sort(C.byref(q), buf.ctypes.data, oe, len(order_entries), limit, offset, out.ctypes.data)
    return out[:n]

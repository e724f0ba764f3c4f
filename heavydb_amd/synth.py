"""Synthetic columnar tables of BASELINE.md section 3 / SURVEY.md section 8(d), generated ON the
device (10 B-row tables never touch the host) with the counter-based generator
`u = splitmix64(seed ^ row * 0x9E3779B97F4A7C15)`, per-column seed 0xC0FFEE00 + col_idx, in
32 M-row fragments (Fragmenter/FragmentDefaultValues.h:19).  torch is used only to own the
HBM allocations.

Each builder returns (RelAlgExecutionUnit, FetchResult, info) for one rank's shard of the
table: fragment f belongs to rank f % world (the reference's fragment -> device rule,
InsertOrderFragmenter.cpp:435-443 / QueryFragmentDescriptor.cpp:159).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

from . import capi
from .capi import (AVG, COUNT, DOUBLE, GEN_F64_UNIT, GEN_I32_MOD, GEN_I32_UNIFORM31, GEN_I64_MOD,
                   GEN_I64_MOD_MUL, INT32, INT64, LT, PROJECT_KEY, SUM)
from .executor import (ExpressionRange, FetchResult, HashJoin, InputColDescriptor, Qual,
                       RelAlgExecutionUnit, TargetExpr, generate_column)

FRAGMENT_ROWS = 32_000_000
SEED0 = 0xC0FFEE00


@dataclass
class ColSpec:
    type: int
    kind: int
    a: int = 0
    b: int = 0
    c: int = 0
    a_f: float = 0.0
    range: ExpressionRange = None


def _torch_dtype(torch, t):
    return {INT32: torch.int32, INT64: torch.int64, DOUBLE: torch.float64}[t]


def my_fragments(total_rows: int, rank: int, world: int, frag_rows: int = FRAGMENT_ROWS) -> List[Tuple[int, int]]:
    """[(row_offset, n_rows)] of the fragments this rank owns."""
    out = []
    f, off = 0, 0
    while off < total_rows:
        n = min(frag_rows, total_rows - off)
        if f % world == rank:
            out.append((off, n))
        off += n
        f += 1
    return out


def generate_table(torch, specs: List[ColSpec], frags: List[Tuple[int, int]], device_id: int):
    """One contiguous allocation per column; fragment chunks are views into it."""
    n_local = sum(n for _, n in frags)
    cols = [torch.empty(max(n_local, 1), dtype=_torch_dtype(torch, s.type), device=f"cuda:{device_id}")
            for s in specs]
    col_buffers, num_rows = [], []
    lo = 0
    for off, n in frags:
        ptrs = []
        for ci, (s, t) in enumerate(zip(specs, cols)):
            ptr = int(t.data_ptr()) + lo * t.element_size()
            generate_column(ptr, n, s.kind, SEED0 + ci, s.a, s.b, s.c, s.a_f, 0, off, device_id)
            ptrs.append(ptr)
        col_buffers.append(ptrs)
        num_rows.append(n)
        lo += n
    torch.cuda.synchronize(device_id)
    return cols, col_buffers, num_rows


def _descs(specs):
    return [InputColDescriptor(s.type, False, s.range) for s in specs]


def gen_tuples(specs: List[ColSpec]):
    """(kind, seed, a, b, c, a_f) per column: what a host-side generator needs to reproduce the table."""
    return [(s.kind, SEED0 + ci, s.a, s.b, s.c, s.a_f) for ci, s in enumerate(specs)]


# ---- cfg1: SELECT COUNT(*) FROM t WHERE i32 < k
def cfg1(torch, total_rows=100_000_000, rank=0, world=1, device_id=0, k=2**30):
    specs = [ColSpec(INT32, GEN_I32_UNIFORM31, range=ExpressionRange(True, 0, 2**31 - 1))]
    frags = my_fragments(total_rows, rank, world)
    cols, bufs, rows = generate_table(torch, specs, frags, device_id)
    ra = RelAlgExecutionUnit(_descs(specs), [TargetExpr(COUNT)], [Qual(0, LT, k)])
    return ra, FetchResult(bufs, rows, device_id=device_id, keepalive=cols), dict(bytes_per_row=4, gens=gen_tuples(specs))


# ---- cfg2: SELECT key, SUM(val) FROM t GROUP BY key   (1 K int32 keys -> perfect hash)
def cfg2(torch, total_rows=1_000_000_000, rank=0, world=1, device_id=0, keyless=False):
    specs = [ColSpec(INT32, GEN_I32_MOD, a=1000, b=0, range=ExpressionRange(True, 0, 999)),
             ColSpec(INT64, GEN_I64_MOD, a=1_000_001 if not keyless else 1_000_000,
                     b=-500_000 if not keyless else 1,
                     range=ExpressionRange(True, -500_000, 500_000) if not keyless
                     else ExpressionRange(True, 1, 1_000_000))]
    frags = my_fragments(total_rows, rank, world)
    cols, bufs, rows = generate_table(torch, specs, frags, device_id)
    ra = RelAlgExecutionUnit(_descs(specs), [TargetExpr(PROJECT_KEY), TargetExpr(SUM, 1)], groupby_exprs=[0])
    return ra, FetchResult(bufs, rows, device_id=device_id, keepalive=cols), dict(bytes_per_row=12, gens=gen_tuples(specs))


# ---- cfg3: SELECT key, COUNT(*), AVG(f64) FROM t [WHERE i32 < k] GROUP BY key
def cfg3(torch, total_rows=10_000_000_000, rank=0, world=1, device_id=0, filtered=True,
         n_keys=10_000_000, k=2**30, prepartitioned=False):
    stride = 1_000_003
    # prepartitioned: the table arrives hash-partitioned by key — rank r only holds the keys
    # ((i * world + r) * stride + 7), so the per-rank results are disjoint and need no merge
    key_spec = (ColSpec(INT64, GEN_I64_MOD_MUL, a=max(n_keys // world, 1), b=stride * world, c=7 + stride * rank,
                        range=ExpressionRange(True, 7, (n_keys - 1) * stride + 7))
                if prepartitioned and world > 1 else
                ColSpec(INT64, GEN_I64_MOD_MUL, a=n_keys, b=stride, c=7,
                        range=ExpressionRange(True, 7, (n_keys - 1) * stride + 7)))
    specs = [key_spec,
             ColSpec(DOUBLE, GEN_F64_UNIT, a_f=1000.0, range=ExpressionRange(True, 0, 0, False, 0.0, 1000.0))]
    quals = []
    if filtered:
        specs.append(ColSpec(INT32, GEN_I32_UNIFORM31, range=ExpressionRange(True, 0, 2**31 - 1)))
        quals = [Qual(2, LT, k)]
    frags = my_fragments(total_rows, rank, world)
    cols, bufs, rows = generate_table(torch, specs, frags, device_id)
    ra = RelAlgExecutionUnit(_descs(specs), [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(AVG, 1)],
                             quals, [0], max_groups_buffer_entry_guess=2 * n_keys)  # 50 % fill
    return ra, FetchResult(bufs, rows, device_id=device_id, keepalive=cols), \
        dict(bytes_per_row=20 if filtered else 16, n_keys=n_keys, gens=gen_tuples(specs))


# ---- cfg4: fact JOIN dim ON fact.k = dim.k ; SUM(fact.v) [, SUM(dim.w)]
def dim_keys_with_holes(xp, dim_rows, mul=1, holes=0):
    """the dimension's keys: 0 .. dim_rows - 1 (x mul); holes = h > 0: without the keys whose index is 5 (mod h) — a perfect
    table with EMPTY slots (the general hash_join_idx probe: round 5's dense dimension is planned away as a range filter)"""
    k = xp.arange(dim_rows, dtype=xp.int64)
    if holes:
        k = k[k % holes != 5]
    return k * mul


def cfg4(torch, total_rows=10_000_000_000, rank=0, world=1, device_id=0, dim_rows=100_000_000,
         sparse=False, sum_dim=False, holes=0):
    mul = 1_000_003 if sparse else 1
    dev = f"cuda:{device_id}"
    key_span = dim_rows                      # the fact keys and the key range cover [0, key_span) x mul whatever the holes
    dim_k = torch.arange(dim_rows, dtype=torch.int64, device=dev)
    if holes:
        dim_k = dim_k[dim_k % holes != 5].contiguous()
    dim_k = dim_k * mul
    dim_rows = int(dim_k.numel())
    dim_w = torch.empty(dim_rows, dtype=torch.int64, device=dev)
    generate_column(int(dim_w.data_ptr()), dim_rows, GEN_I64_MOD, SEED0 + 100, 2001, -1000, 0, 0.0, 0, 0, device_id)
    krange = ExpressionRange(True, 0, (key_span - 1) * mul)
    hj = HashJoin.getInstance(int(dim_k.data_ptr()), dim_rows, INT64, krange, device_id=device_id)
    specs = [ColSpec(INT64, GEN_I64_MOD_MUL if sparse else GEN_I64_MOD, a=key_span, b=mul if sparse else 0, c=0,
                     range=krange),
             ColSpec(INT64, GEN_I64_MOD, a=2_000_001, b=-1_000_000, range=ExpressionRange(True, -10**6, 10**6))]
    frags = my_fragments(total_rows, rank, world)
    cols, bufs, rows = generate_table(torch, specs, frags, device_id)
    targets = [TargetExpr(SUM, 1)] + ([TargetExpr(SUM, 1, 1)] if sum_dim else [])
    ra = RelAlgExecutionUnit(_descs(specs), targets,
                             inner_col_descs=[InputColDescriptor(INT64, False, krange),
                                              InputColDescriptor(INT64, False, ExpressionRange(True, -1000, 1000))],
                             join_outer_col=0, join_table=hj)
    fr = FetchResult(bufs, rows, [int(dim_k.data_ptr()), int(dim_w.data_ptr())], dim_rows, device_id,
                     keepalive=cols + [dim_k, dim_w, hj])
    return ra, fr, dict(bytes_per_row=16, join=hj.info(), gens=gen_tuples(specs), dim_mul=mul, dim_holes=holes, key_span=key_span,
                        dim_w_gen=(GEN_I64_MOD, SEED0 + 100, 2001, -1000, 0, 0.0))


CONFIGS = {"cfg1": cfg1, "cfg2": cfg2, "cfg3": lambda *a, **k: cfg3(*a, filtered=False, **k),
           "cfg3f": lambda *a, **k: cfg3(*a, filtered=True, **k), "cfg4": cfg4}
DEFAULT_ROWS = {"cfg1": 100_000_000, "cfg2": 1_000_000_000, "cfg3": 10_000_000_000,
                "cfg3f": 10_000_000_000, "cfg4": 10_000_000_000}


# ---- projection: SELECT v0, ..., v{n_out-1} FROM t WHERE i32 < k   (row-emitting filter / project)
def projection(torch, total_rows=1_000_000_000, n_out=3, selectivity=0.5, rank=0, world=1, device_id=0, columnar=False,
               scan_limit=0, entry_guess=None, cols_cache=None):
    """One INT32 filter column uniform in [0, 2^31) and n_out value columns (INT64 / DOUBLE alternating).  The output
    buffer is sized like the reference sizes it after its COUNT(*) pre-flight (RelAlgExecutor::getFilteredCountAll):
    the expected number of matches plus a small margin, unless entry_guess is given.  cols_cache: a dict that keeps the
    generated table between calls with the same (rows, n columns) — the generated values do not depend on n_out."""
    from .capi import PROJECT
    specs = [ColSpec(INT32, GEN_I32_UNIFORM31, range=ExpressionRange(True, 0, 2**31 - 1))]
    for i in range(n_out):
        specs.append(ColSpec(INT64, GEN_I64_MOD, a=1_000_000_007, b=-500_000_000, range=ExpressionRange(True, -500_000_000, 500_000_006))
                     if i % 2 == 0 else
                     ColSpec(DOUBLE, GEN_F64_UNIT, a_f=1000.0, range=ExpressionRange(True, 0, 0, False, 0.0, 1000.0)))
    frags = my_fragments(total_rows, rank, world)
    key = (total_rows, rank, world, device_id)
    if cols_cache is not None and key in cols_cache and len(cols_cache[key][0]) >= len(specs):
        cols, bufs, rows = cols_cache[key]
        cols, bufs = cols[:len(specs)], [b[:len(specs)] for b in bufs]
    else:
        cols, bufs, rows = generate_table(torch, specs, frags, device_id)
        if cols_cache is not None:
            cols_cache.clear()
            cols_cache[key] = (cols, bufs, rows)
    k = int(selectivity * 2**31)
    n_local = sum(rows)
    guess = entry_guess if entry_guess is not None else min(int(n_local * selectivity * 1.01) + 4096, 2**31 - 1)
    ra = RelAlgExecutionUnit(_descs(specs), [TargetExpr(PROJECT, 1 + i) for i in range(n_out)], [Qual(0, LT, k)],
                             max_groups_buffer_entry_guess=max(guess, 1), scan_limit=scan_limit,
                             output_columnar_hint=capi.OUTPUT_COLUMNAR if columnar else 0)
    return ra, FetchResult(bufs, rows, device_id=device_id, keepalive=cols), dict(
        bytes_per_row=4 + 8 * n_out, out_bytes_per_row=8 + 8 * n_out, gens=gen_tuples(specs), k=k, cols=cols)

"""world_size-2 (and 3) CPU runs of the multi-GPU merge choreography (heavydb_amd/multi_gpu.py)
over the `gloo` backend.  The collectives, the split/exchange/fold order and the ownership
rules are exactly the ones the GPU path runs over RCCL; only the ShardOps backend differs: here
it is numpy + the oracle (test infrastructure), on the GPU it is HipShard (C-ABI kernels).

Each rank executes the oracle on its fragments (fragment f -> rank f % world, the reference's
rule), merges, and rank 0 compares with the oracle run over ALL fragments."""
from __future__ import annotations

import os
import socket
import sys
import traceback

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMPTY64 = 2**63 - 1


class NumpyShard:
    """ShardOps over numpy buffers + the oracle's reduce/init (CPU stand-in for HipShard)."""

    def __init__(self, torch, orc, q, buf=None):
        self._torch, self._orc, self._q = torch, orc, q
        self._np = buf if buf is not None else orc.init_buffer(q)
        self._np = np.ascontiguousarray(self._np)
        if not q.output_columnar:  # a columnar buffer stays flat (columns end to end)
            self._np = self._np.reshape(q.entry_count, q.row_size // 8)

    def _rows(self):
        from tests.helpers import columnar_to_rows
        return columnar_to_rows(self._q, self._np) if self._q.output_columnar else self._np

    def qmd(self):
        return self._q

    def buffer(self):
        return self._torch.from_numpy(self._np)  # shares memory: all_reduce results land in place

    def partition_rows(self, n_parts):
        rows = self._rows()
        live = rows[rows[:, 0] != EMPTY64]
        part = _owner(self._q, live, n_parts)
        order = np.argsort(part, kind="stable")
        counts = np.bincount(part, minlength=n_parts).tolist()
        return self._torch.from_numpy(np.ascontiguousarray(live[order])), [int(c) for c in counts]

    def fresh_like(self):
        return NumpyShard(self._torch, self._orc, self._q)

    def merge_rows(self, rows):
        rows = rows.numpy()
        n = rows.shape[0]
        if not n:
            return
        assert n <= self._q.entry_count
        that = self._orc.init_buffer(self._q)
        if self._q.output_columnar:  # lay the received rows into the first n entries, column by column
            flat, e = that.view(np.int64), self._q.entry_count
            for c in range(rows.shape[1]):
                flat[c * e:c * e + n] = rows[:, c]
        else:
            that = that.reshape(self._q.entry_count, -1)
            that[:n] = rows
        assert self._orc.reduce(self._q, self._np, that) == 0

    # ---- slice exchange (multi_gpu._merge_keyed_by_slices): the numpy twins of mi355q_shard_pads /
    # mi355q_shard_merge_range
    def boundary_pads(self, world, pad_rows):
        from heavydb_amd.multi_gpu import slice_bounds
        rows, e = self._rows(), self._q.entry_count
        b = slice_bounds(e, world)
        pads = np.zeros((world, pad_rows, rows.shape[1]), dtype=np.int64)
        ok = np.zeros(world, dtype=np.int32)
        for r in range(world):
            idx = (b[r + 1] + np.arange(pad_rows)) % e
            pads[r] = rows[idx]
            ok[r] = int((rows[idx, 0] == EMPTY64).any())
        return self._torch.from_numpy(pads), self._torch.from_numpy(ok)

    def merge_range(self, rows, home_lo, home_hi):
        from tests.helpers import murmur3_u64
        rows = rows.numpy().reshape(-1, self._np.shape[1])
        live = rows[rows[:, 0] != EMPTY64]
        home = (murmur3_u64(live[:, 0]) % np.uint64(self._q.entry_count)).astype(np.int64)
        mine = live[(home >= home_lo) & (home < home_hi)]
        for o in range(0, len(mine), self._q.entry_count):  # merge_rows lays at most entry_count rows into a buffer
            self.merge_rows(self._torch.from_numpy(np.ascontiguousarray(mine[o:o + self._q.entry_count])))

    def merge_slices(self, recv_main, recv_pads, n_src, home_lo, home_hi):
        # twin of mi355q_shard_merge_slices: the same fold in one call (16 sources at most, like the library)
        if n_src > 16:
            return False
        assert recv_main.shape[0] == n_src * (home_hi - home_lo) and recv_pads.shape[0] == n_src
        self.merge_range(recv_main, home_lo, home_hi)
        self.merge_range(recv_pads.reshape(-1, recv_pads.shape[-1]), home_lo, home_hi)
        return True

    def reduce_from(self, other_buffer):
        other = np.ascontiguousarray(other_buffer.numpy()).reshape(self._np.shape)
        assert self._orc.reduce(self._q, self._np, other) == 0


def _owner(q, live, n_parts):
    """The shard function of mi355q_shard_partition: upper bits of the key hash — of the single
    int64 key, or of all the key bytes for a multi-column key."""
    from tests.helpers import murmur3_u64, murmur3_words
    assert q.key_width == 8
    if q.group_col_count > 1:
        kq = q.key_bytes // 8
        words = np.ascontiguousarray(live[:, :kq]).view(np.uint32).reshape(live.shape[0], 2 * kq)
        h = murmur3_words(words[:, :2 * q.group_col_count])
    else:
        h = murmur3_u64(live[:, 0])
    return ((h * np.uint64(n_parts)) >> np.uint64(32)).astype(np.int64)


def _table(shape, seed=7):
    from heavydb_amd import capi
    if shape.endswith("_columnar"):  # the same step with a columnar result buffer
        ra, frags = _table(shape[:-len("_columnar")], seed)
        ra.output_columnar_hint = capi.OUTPUT_COLUMNAR
        return ra, frags
    from heavydb_amd.executor import (ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit,
                                      TargetExpr)
    rng = np.random.default_rng(seed)
    n = 60_000
    if shape == "keyed":
        n_keys = 9_000
        key = (rng.integers(0, n_keys, n) * 1000003 + 7).astype(np.int64)
        val = (rng.random(n) * 1000.0).astype(np.float64)
        fil = rng.integers(0, 2**31 - 1, n).astype(np.int32)
        descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, (n_keys - 1) * 1000003 + 7)),
                 InputColDescriptor(capi.DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 1000.0)),
                 InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 2**31 - 1))]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1)],
                                 [Qual(2, capi.LT, 2**30)], [0], max_groups_buffer_entry_guess=2 * n_keys)
        cols = [key, val, fil]
    elif shape in ("keyed_sliced", "keyed_sliced_dense"):
        # a table long enough for the slice exchange (entry_count / world >= 4 pads); the dense variant
        # fills it to ~85 % so that probe clusters cross the slice boundaries (strays + pads matter)
        n_keys = 40_000 if shape == "keyed_sliced" else 68_000
        n = 200_000
        key = (rng.integers(0, n_keys, n) * 1000003 + 7).astype(np.int64)
        val = (rng.random(n) * 1000.0).astype(np.float64)
        ival = rng.integers(-10**6, 10**6, n).astype(np.int64)
        ival[rng.random(n) < 0.1] = -(2**63)
        descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, (n_keys - 1) * 1000003 + 7)),
                 InputColDescriptor(capi.DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 1000.0)),
                 InputColDescriptor(capi.INT64, True, ExpressionRange(True, -10**6, 10**6, True))]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1),
                                         TargetExpr(capi.MIN, 2), TargetExpr(capi.SUM, 2)],
                                 [], [0], max_groups_buffer_entry_guess=80_000)
        cols = [key, val, ival]
    elif shape == "keyed_two_columns":  # multi-column baseline key: sharded by the whole key's hash
        k0 = (rng.integers(0, 900, n) * 1000003 + 7).astype(np.int64)
        k1 = rng.integers(-5, 5, n).astype(np.int64)
        k1[rng.random(n) < 0.1] = -(2**63)
        val = (rng.random(n) * 1000.0).astype(np.float64)
        descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, 899 * 1000003 + 7)),
                 InputColDescriptor(capi.INT64, True, ExpressionRange(True, -5, 4, True)),
                 InputColDescriptor(capi.DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 1000.0))]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.PROJECT_KEY, 1),
                                         TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 2)],
                                 groupby_exprs=[0, 1], max_groups_buffer_entry_guess=30_000)
        cols = [k0, k1, val]
    elif shape == "keyed_compact":  # COUNT(*)-only: 4-byte slots, 16-byte rows travel whole
        key = (rng.integers(0, 9000, n) * 1000003 + 7).astype(np.int64)
        descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, 8999 * 1000003 + 7))]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT)], groupby_exprs=[0], max_groups_buffer_entry_guess=18_000)
        cols = [key]
    elif shape == "projection":  # SELECT a, v FROM t WHERE f < 2^30: rows are emitted, not aggregated — nothing is reduced
        a = rng.integers(-1000, 1000, n).astype(np.int64)
        v = (rng.random(n) * 10.0).astype(np.float64)
        fil = rng.integers(0, 2**31 - 1, n).astype(np.int32)
        descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, -1000, 999)),
                 InputColDescriptor(capi.DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 10.0)),
                 InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 2**31 - 1))]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT, 0), TargetExpr(capi.PROJECT, 1)], [Qual(2, capi.LT, 2**30)],
                                 max_groups_buffer_entry_guess=n)
        cols = [a, v, fil]
    elif shape == "perfect_float":  # float slots: merged by the reduce rule, not by all_reduce
        key = rng.integers(0, 300, n).astype(np.int32)
        val = (rng.random(n) * 10.0).astype(np.float32)
        descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 299)),
                 InputColDescriptor(capi.FLOAT, False, ExpressionRange(True, 0, 0, False, 0.0, 10.0))]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 1), TargetExpr(capi.MAX, 1),
                                         TargetExpr(capi.COUNT)], groupby_exprs=[0])
        cols = [key, val]
    elif shape == "perfect":
        key = rng.integers(0, 1000, n).astype(np.int32)
        val = rng.integers(-500_000, 500_001, n).astype(np.int64)
        descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 999)),
                 InputColDescriptor(capi.INT64, False, ExpressionRange(True, -500_000, 500_000))]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 1), TargetExpr(capi.MIN, 1),
                                         TargetExpr(capi.MAX, 1)], groupby_exprs=[0])
        cols = [key, val]
    elif shape == "perfect_two_columns_unprojected":  # every key quad has to be merged, projected or not
        # sorted first key: every fragment (hence every rank) sees only a slice of the groups
        k0 = (np.arange(n) * 40 // n).astype(np.int32)
        k1 = rng.integers(-3, 4, n).astype(np.int64)
        val = rng.integers(1, 1000, n).astype(np.int64)
        descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 39)),
                 InputColDescriptor(capi.INT64, False, ExpressionRange(True, -3, 3)),
                 InputColDescriptor(capi.INT64, False, ExpressionRange(False))]  # NOT NULL: the all_reduce branch
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.MAX, 2), TargetExpr(capi.MIN, 2)], groupby_exprs=[0, 1])
        cols = [k0, k1, val]
    elif shape == "perfect_nullable":  # NULL-aware slots: exercises the all_gather + reduce branch
        key = rng.integers(0, 50, n).astype(np.int32)
        val = rng.integers(-1000, 1000, n).astype(np.int64)
        val[rng.random(n) < 0.3] = -(2**63)
        descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 49)),
                 InputColDescriptor(capi.INT64, True, ExpressionRange(True, -1000, 999, True))]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 1), TargetExpr(capi.COUNT, 1),
                                         TargetExpr(capi.AVG, 1)], groupby_exprs=[0])
        cols = [key, val]
    else:  # non-grouped COUNT(*) WHERE
        fil = rng.integers(0, 2**31 - 1, n).astype(np.int32)
        descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 2**31 - 1))]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT)], [Qual(0, capi.LT, 2**30)])
        cols = [fil]
    n_frags = 7
    cuts = np.linspace(0, n, n_frags + 1).astype(int)
    frags = [[c[cuts[i]:cuts[i + 1]] for c in cols] for i in range(n_frags)]
    return ra, frags


def _bind_host_simulation(torch):
    """This process's `device` is the host simulation of the library (tests/hostsim, every kernel file compiled for the
    CPU): HipShard — the ShardOps the GPU path uses — then runs the product's own mi355q_execute / mi355q_shard_* /
    slice-merge kernels on tensors that are plain CPU tensors, and gloo moves them."""
    from heavydb_amd import capi
    from tests.helpers import hostsim_lib
    capi._lib = capi.load_library(hostsim_lib(real_fast=True))
    real = {n: getattr(torch, n) for n in ("zeros", "empty", "full", "arange")}
    strip = lambda f: (lambda *a, **k: f(*a, **{x: y for x, y in k.items() if x != "device" or not str(y).startswith("cuda")}))  # noqa: E731
    for n, f in real.items():
        setattr(torch, n, strip(f))

    class _Stream:
        def synchronize(self):
            pass
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.synchronize = lambda *a, **k: None


def _aligned_copy(a):
    a = np.ascontiguousarray(a)
    raw = np.empty(a.nbytes + 64, np.uint8)
    off = (-raw.ctypes.data) % 64
    out = raw[off:off + a.nbytes].view(a.dtype)
    out[...] = a
    return out


def _worker(rank, world, port, shape, errq, real_library=False):
    try:
        import torch
        import torch.distributed as dist
        from heavydb_amd import capi
        from heavydb_amd.multi_gpu import merge
        from oracle import oracle as orc
        from tests.helpers import compare_buffers, compare_rows
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if real_library:
            _bind_host_simulation(torch)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        prepart = shape == "keyed_prepartitioned"
        mismatched = shape == "keyed_sliced_mismatched"
        ra, frags = _table("keyed" if prepart else "keyed_sliced" if mismatched else shape)
        if mismatched and rank % 2 == 1:
            # what Executor.executeWorkUnit's retry ladder leaves behind on a rank whose shard ran out of slots:
            # a table twice the size of its peers' (the slice exchange must not run then)
            ra.max_groups_buffer_entry_guess *= 2
        plan = ra.to_plan()
        if prepart:
            # rows dealt to the ranks BY KEY (every key on exactly one rank): no exchange, gather only
            cols = [np.concatenate([f[c] for f in frags]) for c in range(len(frags[0]))]
            sel = ((cols[0] - 7) // 1000003) % world == rank
            mine = [[c[sel] for c in cols]]
        else:
            mine = [f for i, f in enumerate(frags) if i % world == rank]
        if real_library:
            # the rank's step through the product's own executor and kernels, result storage owned by a tensor (HipShard)
            from heavydb_amd.executor import Executor, FetchResult
            from heavydb_amd.multi_gpu import HipShard
            keep = [[_aligned_copy(c) for c in f] for f in mine]
            fr = FetchResult([[c.ctypes.data for c in f] for f in keep], [len(f[0]) for f in keep], keepalive=keep)
            shard = HipShard.execute(torch, Executor(0), ra, fr)
            q = shard.qmd()
        else:
            q, buf, code = orc.execute(plan, mine, n_threads=1)
            assert code == 0
            shard = NumpyShard(torch, orc, q, buf)
        before = shard.buffer().numpy().copy()
        out = merge(shard, dist, torch, gather_to_rank0=True, prepartitioned=prepart)
        if q.desc_type == capi.PROJECTION:
            # every rank keeps the rows of its own fragments (Executor::resultsUnion appends the devices' results): the part
            # is the oracle's over this rank's fragments, and the parts together are as many rows as the whole step emits
            assert out is shard and np.array_equal(before, out.buffer().numpy())
            q_mine, want_mine, code = orc.execute(plan, mine, n_threads=1)
            assert code == 0
            mine_rows = orc.fetch_rows(q_mine, want_mine)
            compare_rows(q_mine, mine_rows, orc.fetch_rows(q, out.buffer().numpy().reshape(-1)), 0.0)
            total = torch.tensor([len(mine_rows[0])], dtype=torch.int64)
            dist.all_reduce(total)
            q_all, want, code = orc.execute(plan, frags, n_threads=2)
            assert code == 0 and int(total.item()) == orc.row_count(q_all, want) > 0
            dist.barrier()
            dist.destroy_process_group()
            return
        if prepart and rank != 0:
            assert out is shard and np.array_equal(before, out.buffer().numpy())  # nothing moved
        if mismatched:
            from heavydb_amd import multi_gpu
            assert multi_gpu.LAST_KEYED_PATH == "partition", multi_gpu.LAST_KEYED_PATH
        elif shape.startswith("keyed_sliced") or shape == "keyed":
            from heavydb_amd import multi_gpu
            assert multi_gpu.LAST_KEYED_PATH == "slices", multi_gpu.LAST_KEYED_PATH
        q_all, want, code = orc.execute(plan, frags, n_threads=2)
        assert code == 0
        got = out.buffer().numpy()
        if q.output_columnar:  # compared as row images (tests/helpers.columnar_to_rows)
            from tests.helpers import columnar_to_rows, rowwise_qmd
            assert got.ndim == 1 and got.nbytes == orc.buffer_bytes(q)
            got, want = columnar_to_rows(q, got), columnar_to_rows(q_all, want)
            q_all = rowwise_qmd(q_all)
        if q.desc_type == capi.GROUP_BY_BASELINE_HASH:
            # every rank owns exactly the keys of its shard after the all-to-all ...
            live = got[got[:, 0] != EMPTY64]
            from heavydb_amd.multi_gpu import slice_bounds, slice_exchange_ok
            if slice_exchange_ok(q, world) and not prepart and not mismatched:  # ownership by home-slot range
                from tests.helpers import murmur3_u64
                home = (murmur3_u64(live[:, 0]) % np.uint64(q.entry_count)).astype(np.int64)
                owner = np.searchsorted(np.array(slice_bounds(q.entry_count, world)[1:]), home, side="right")
            else:
                owner = _owner(q, live, world)
            if rank != 0:
                assert prepart or (owner == rank).all()
            else:  # ... and rank 0 additionally gathered everything
                compare_buffers(q_all, want, got.reshape(-1), 1e-9)
                compare_rows(q_all, orc.fetch_rows(q_all, want), orc.fetch_rows(q_all, got.reshape(-1)), 1e-9)
        else:
            compare_buffers(q_all, want, got.reshape(-1), 1e-9)  # dense: every rank holds the full result
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # surface the traceback in the parent
        errq.put((rank, traceback.format_exc()))
        raise


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,shape", [(2, "keyed_sliced"), (3, "keyed_sliced_dense"), (5, "keyed_sliced_dense"),
                                         (8, "keyed_sliced"), (3, "keyed_sliced_mismatched")])
def test_slice_exchange_over_gloo(world, shape):
    """The keyed merge by home-slot slices (no partition pass, no count exchange) at world 2 / 3 / 5 / 8:
    uneven slice lengths, probe clusters crossing the boundaries, the wrap-around pad of the last slice."""
    test_merge_over_gloo(shape, world)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("shape", ["keyed", "keyed_two_columns", "keyed_compact", "perfect", "perfect_nullable",
                                   "perfect_float", "non_grouped", "perfect_two_columns_unprojected",
                                   "perfect_columnar", "perfect_nullable_columnar", "keyed_columnar",
                                   "keyed_two_columns_columnar", "perfect_two_columns_unprojected_columnar", "keyed_prepartitioned",
                                   "projection", "projection_columnar"])
def test_merge_over_gloo(shape, world, real_library=False):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    orc.lib()  # build once in the parent
    if real_library:
        from tests.helpers import hostsim_lib
        hostsim_lib(real_fast=True)
    ctx = mp.get_context("spawn")
    errq = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, errq, real_library)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    errs = []
    while not errq.empty():
        errs.append(errq.get())
    for p in procs:
        if p.is_alive():
            p.terminate()
            errs.append((-1, "timeout"))
    assert not errs, "\n".join(f"rank {r}:\n{t}" for r, t in errs)
    assert all(p.exitcode == 0 for p in procs)


@pytest.mark.parametrize("world,shape", [(2, "keyed_sliced"), (3, "keyed_sliced_dense"), (2, "keyed"), (2, "keyed_two_columns"),
                                         (3, "keyed_compact"), (2, "perfect"), (3, "perfect_nullable"), (2, "perfect_float"),
                                         (2, "non_grouped"), (2, "keyed_columnar"), (2, "perfect_columnar"),
                                         (2, "keyed_prepartitioned"), (3, "keyed_sliced_mismatched"), (2, "projection"),
                                         (3, "projection_columnar")])
def test_merge_over_gloo_with_the_librarys_own_code(world, shape):
    """The same choreography with the PRODUCT on every rank instead of the numpy twin: each rank's step runs through
    mi355q_execute, its shard is a HipShard (mi355q_shard_pads / _merge_slices with the LDS slice fold / _partition /
    _merge_rows / _reduce, the calls the GPU path makes), on the host simulation of the library (tests/hostsim: every
    kernel file compiled for the CPU); gloo carries the tensors RCCL would.  Rank 0 compares with the oracle over all
    fragments.  What this cannot show is RCCL itself and timing — the ABI sequence and its device code, it can."""
    test_merge_over_gloo(shape, world, real_library=True)


def test_prepartitioned_key_streams_are_disjoint(oracle):
    """bench.py --prepartitioned: the per-rank key generators of heavydb_amd/synth.cfg3 (same
    splitmix64 stream as the device generator) produce disjoint slices of the original key set."""
    from heavydb_amd import capi
    n_keys, stride, world, n = 1000, 1_000_003, 4, 20_000
    full = set(oracle.generate_column(n, capi.GEN_I64_MOD_MUL, 0xC0FFEE00, n_keys, stride, 7).tolist())
    seen = set()
    for rank in range(world):
        k = set(oracle.generate_column(n, capi.GEN_I64_MOD_MUL, 0xC0FFEE00, n_keys // world, stride * world,
                                       7 + stride * rank).tolist())
        assert len(k) == n_keys // world and not (k & seen) and k <= full
        assert all(((x - 7) // stride) % world == rank for x in k)
        seen |= k
    assert seen == full

"""Multi-GPU execution of one query step: one process per GPU, fragments dealt round-robin
(fragment f -> rank f % world, the reference's rule, InsertOrderFragmenter.cpp:435-443), each
rank runs the step on its shard, and the partial ResultSets are merged ON the devices over
RCCL/xGMI (`torch.distributed`, backend "nccl" = RCCL on ROCm).

This replaces the reference's multi-device merge — every device's whole output buffer copied
to the host and reduced pairwise on CPU threads (Executor::reduceMultiDeviceResultSets,
Execute.cpp:1772-1792; ResultSetStorage::reduce, ResultSetReduction.cpp:203) — with:

  dense layouts (non-grouped, perfect hash): identical slot <-> key mapping on every rank.
      * every slot additive / min / max and NOT NULL  -> one all_reduce per (dtype, op) group
      * anything else (NULL-aware slots, key projections) -> all_gather of the (small) buffers
        followed by the device reduce kernel, i.e. exactly ResultSetStorage::reduce semantics
  keyed layout (baseline hash): slot positions differ per rank, so each rank compacts its
      live entries into `world` runs by key hash, the runs are exchanged with ONE all_to_all
      (every pair uses its own xGMI link; no ring), and each rank folds what it receives into
      a fresh table.  The result stays hash-partitioned across the ranks: rank r owns the keys
      with shard(key) == r.  `gather_to_rank0=True` additionally collects the rows on rank 0.

The collective choreography is backend-agnostic (`ShardOps`); the product backend is
`HipShard` (C-ABI kernels).  The CPU/gloo unit tests plug a numpy+oracle backend into the
same functions.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Protocol, Tuple

from . import capi
from .capi import check


class ShardOps(Protocol):
    def qmd(self) -> capi.QMD: ...
    def buffer(self): ...                       # torch int64 tensor [entry_count, row_quad]; flat when columnar
    def partition_rows(self, n_parts: int): ...  # -> (ROW-WISE rows tensor [live, row_quad], counts list)
    def fresh_like(self) -> "ShardOps": ...
    def merge_rows(self, rows) -> None: ...
    def reduce_from(self, other_buffer) -> None: ...
    # optional (slice exchange): boundary_pads(world, pad_rows) -> (pads [world, pad_rows, rq], ok int32[world]);
    #                            merge_range(rows, home_lo, home_hi)


def _dense_slot_ops(q: capi.QMD) -> Optional[List[Tuple[int, str, bool]]]:
    """[(quad column, 'sum'|'min'|'max', is_fp)] when every quad of the row can be merged by
    a plain all_reduce; None when NULL-aware or projected slots need the reduce kernel."""
    if q.slot_width != 8:  # 4-byte slots: two per quad -> the reduce kernel
        return None
    kq = q.key_bytes // 8
    ops: List[Tuple[int, str, bool]] = []
    for k in range(kq):  # every key quad (a multi-column perfect hash stores one per group column)
        ops.append((k, "min", False))  # EMPTY_KEY_64 = INT64_MAX, so MIN keeps the key
    for t in range(q.n_targets):
        s = q.target_slot[t]
        agg = q.target_agg[t]
        if s < 0:
            continue
        if q.target_skip_null[t] or agg == capi.PROJECT_KEY or q.target_arg_is_f32[t]:
            return None
        fp = bool(q.target_arg_is_fp[t])
        if agg in (capi.COUNT, capi.COUNT_IF):
            ops.append((kq + s, "sum", False))
        elif agg in (capi.SUM, capi.SUM_IF):
            ops.append((kq + s, "sum", fp))
        elif agg == capi.AVG:
            ops.append((kq + s, "sum", fp))
            ops.append((kq + s + 1, "sum", False))
        elif agg == capi.MIN:
            ops.append((kq + s, "min", fp))
        elif agg == capi.MAX:
            ops.append((kq + s, "max", fp))
        else:
            return None
    return ops


def merge_dense(shard: ShardOps, dist, torch, group=None) -> ShardOps:
    q = shard.qmd()
    world = dist.get_world_size(group)
    if world == 1:
        return shard
    buf = shard.buffer()
    ops = _dense_slot_ops(q)
    red = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}
    if ops is not None and q.output_columnar:
        # columnar buffer: quad column c of the row image IS a contiguous column of entry_count
        # 8-byte values at c * entry_count quads (group columns first, then the slots; 8-byte
        # slots, or _dense_slot_ops would have said no) -> reduced in place, no staging copy
        flat = buf.view(-1)
        n = q.entry_count
        for col, op, fp in ops:
            view = flat[col * n:(col + 1) * n]
            dist.all_reduce(view.view(torch.float64) if fp else view, op=red[op], group=group)
        return shard
    if ops is not None:
        groups = {}
        for col, op, fp in ops:
            groups.setdefault((op, fp), []).append(col)
        for (op, fp), cols in groups.items():
            view = buf.view(torch.float64) if fp else buf
            t = view[:, cols].contiguous()
            dist.all_reduce(t, op=red[op], group=group)
            view[:, cols] = t
        return shard
    # exact ResultSetStorage::reduce semantics: gather all partial buffers, fold locally
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf.contiguous(), group=group)
    rank = dist.get_rank(group)
    out = shard.fresh_like()
    for r in range(world):  # same order on every rank -> identical result everywhere
        out.reduce_from(gathered[r] if r != rank else buf)
    return out


def merge_keyed(shard: ShardOps, dist, torch, group=None, gather_to_rank0: bool = False,
                prepartitioned: bool = False) -> ShardOps:
    """prepartitioned: the input ROWS were already dealt to the ranks by key (every key lives on one
    rank), so the per-rank tables are disjoint and there is nothing to exchange ("no final
    collective, just gather"); only the optional gather to rank 0 runs."""
    q = shard.qmd()
    world = dist.get_world_size(group)
    if world == 1:
        return shard
    rq = q.row_size // 8
    if prepartitioned:
        return _gather_to_rank0(shard, shard, dist, torch, group, rq) if gather_to_rank0 else shard
    global LAST_KEYED_PATH
    LAST_KEYED_PATH = "partition"
    # The slice exchange sends table rows IN PLACE with split sizes every rank computes from ITS OWN entry count:
    # every rank must hold a table of the same geometry.  Executor.executeWorkUnit doubles the entry guess per rank
    # when a shard runs out of slots, so skewed shards can end up with different table sizes — ranks would then take
    # different collective sequences or pass mismatched split sizes (a hang, or a silently wrong merge).  One tiny
    # all-reduce settles it: the slices path only runs when (entry_count, row_size, slice-path-capable) agree on
    # every rank; the partition path below re-hashes rows and tolerates any mix of table sizes.
    mine = slice_exchange_ok(q, world) and hasattr(shard, "boundary_pads")
    dev = shard.buffer().device   # the collective's tensors live where the table lives (RCCL: HBM; gloo: host)
    geom = torch.tensor([q.entry_count, -q.entry_count, q.row_size, -q.row_size, 1 if mine else 0], dtype=torch.int64,
                        device=dev)
    dist.all_reduce(geom, op=dist.ReduceOp.MIN, group=group)
    g = [int(x) for x in geom.cpu().tolist()]
    agree = g[0] == -g[1] and g[2] == -g[3] and g[4] == 1
    if agree:
        out = _merge_keyed_by_slices(shard, dist, torch, group)
        if out is not None:
            LAST_KEYED_PATH = "slices"
            return _gather_to_rank0(out, shard, dist, torch, group, rq) if gather_to_rank0 else out
    rows, counts = shard.partition_rows(world)
    send_counts = torch.tensor(counts, dtype=torch.int64, device=rows.device)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    recv = [int(x) for x in recv_counts.cpu().tolist()]
    recv_rows = torch.empty((sum(recv), rq), dtype=torch.int64, device=rows.device)
    dist.all_to_all_single(recv_rows.view(-1), rows.contiguous().view(-1),
                           output_split_sizes=[c * rq for c in recv],
                           input_split_sizes=[c * rq for c in counts], group=group)
    global LAST_MERGE_BYTES_SENT
    LAST_MERGE_BYTES_SENT = int((sum(counts) - counts[dist.get_rank(group)]) * q.row_size)
    out = shard.fresh_like()
    out.merge_rows(recv_rows)
    return _gather_to_rank0(out, shard, dist, torch, group, rq) if gather_to_rank0 else out


LAST_KEYED_PATH = ""   # which keyed merge ran last in this process ("slices" / "partition"): for tests and bench.py
LAST_MERGE_BYTES_SENT = 0  # bytes this rank handed to the collectives of the last merge (bench.py reports it)
LAST_SLICE_FOLD = ""   # how the received slices were folded last ("lds": mi355q_shard_merge_slices / "rows")
SLICE_PAD_ROWS = 1024  # rows after a slice's end that travel with it (the tail of a boundary-crossing cluster)


def slice_bounds(entry_count: int, world: int) -> List[int]:
    """bound(r) = r * entry_count // world: rank r owns the home slots [bound(r), bound(r + 1))."""
    return [r * entry_count // world for r in range(world + 1)]


def slice_exchange_ok(q: capi.QMD, world: int) -> bool:
    """The shapes the slice exchange takes: row-wise baseline table, one 8-byte key, 8-byte slots, and
    slices comfortably longer than the pad."""
    return (q.desc_type == capi.GROUP_BY_BASELINE_HASH and not q.output_columnar and q.group_col_count == 1
            and q.key_width == 8 and q.slot_width == 8 and q.entry_count // world >= 4 * SLICE_PAD_ROWS)


def _merge_keyed_by_slices(shard: ShardOps, dist, torch, group) -> Optional[ShardOps]:
    """Keyed merge without a partition pass and without exchanging counts: rank r
    owns the keys whose home slot is in slice r of the table; linear probing keeps a key at or just after
    its home slot, so every rank sends slice r of ITS table to rank r in place — the split sizes are the
    slice lengths, known to everybody — followed by a second, tiny all_to_all with the SLICE_PAD_ROWS
    rows after each slice (clusters that cross a boundary).  The receiver folds the `world` slices and
    pads of its range into a fresh table, keeping only keys whose home slot is in its range.  One
    device->host read at the very end (the all-reduced "every pad ended in an empty slot" flag); when
    it says no, None is returned and the caller runs the general partition + exchange path."""
    q = shard.qmd()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rq = q.row_size // 8
    b = slice_bounds(q.entry_count, world)
    table = shard.buffer()
    pads, ok = shard.boundary_pads(world, SLICE_PAD_ROWS)        # [world, pad, rq] rows, int32[world]
    my_len = b[rank + 1] - b[rank]
    recv_main = torch.empty((world * my_len, rq), dtype=torch.int64, device=table.device)
    dist.all_to_all_single(recv_main.view(-1), table.contiguous().view(-1),
                           output_split_sizes=[my_len * rq] * world,
                           input_split_sizes=[(b[r + 1] - b[r]) * rq for r in range(world)], group=group)
    recv_pads = torch.empty_like(pads)
    dist.all_to_all_single(recv_pads.view(-1), pads.contiguous().view(-1), group=group)
    global LAST_MERGE_BYTES_SENT
    LAST_MERGE_BYTES_SENT = int((q.entry_count - my_len) * q.row_size + (world - 1) * SLICE_PAD_ROWS * q.row_size)
    flag = ok.min().to(torch.int64).reshape(1)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    out = shard.fresh_like()
    # one LDS fold of all slices and pads when the backend has it and takes the layout ...
    folded = out.merge_slices(recv_main, recv_pads, world, b[rank], b[rank + 1]) if hasattr(out, "merge_slices") else False
    if folded is None:      # ... overflowed half way: start from a fresh table
        out = shard.fresh_like()
    if not folded:          # ... else row by row with the reduce kernel
        out.merge_range(recv_main, b[rank], b[rank + 1])
        out.merge_range(recv_pads.view(-1, rq), b[rank], b[rank + 1])
    global LAST_SLICE_FOLD
    LAST_SLICE_FOLD = "lds" if folded else "rows"
    if int(flag.item()) == 0:   # a cluster longer than the pad somewhere: the general path
        return None
    return out


def _gather_to_rank0(out: ShardOps, shard: ShardOps, dist, torch, group, rq: int) -> ShardOps:
    """Rank 0 additionally folds every other rank's (disjoint) rows into its table."""
    world = dist.get_world_size(group)
    mine, _ = out.partition_rows(1)
    n_mine = torch.tensor([mine.shape[0]], dtype=torch.int64, device=mine.device)
    sizes = [torch.empty_like(n_mine) for _ in range(world)]
    dist.all_gather(sizes, n_mine, group=group)
    sizes = [int(s.item()) for s in sizes]
    rank = dist.get_rank(group)
    if rank == 0:
        bufs = [torch.empty((s, rq), dtype=torch.int64, device=mine.device) for s in sizes]
        reqs = [dist.irecv(bufs[r], src=r, group=group) for r in range(1, world) if sizes[r]]
        for r_ in reqs:
            r_.wait()
        for r in range(1, world):
            if sizes[r]:
                out.merge_rows(bufs[r])
    elif mine.shape[0]:
        dist.send(mine.contiguous(), dst=0, group=group)
    return out


def merge(shard: ShardOps, dist, torch, group=None, gather_to_rank0: bool = False,
          prepartitioned: bool = False) -> ShardOps:
    """The final-aggregate merge of one step across the ranks of `group`."""
    if shard.qmd().desc_type == capi.PROJECTION:
        # a projection is not reduced: the devices' results are laid one behind the other (Executor::resultsUnion ->
        # ResultSet::append, Execute.cpp:1642-1694, ResultSet.cpp:307-335).  Here every rank keeps the rows of its own
        # fragments on its device — the result is the ranks' parts in rank order, total_matched their sum (all_reduce
        # by the caller where it needs the number); parts on ONE device are put together by mi355q_result_append.
        return shard
    if shard.qmd().desc_type == capi.GROUP_BY_BASELINE_HASH:
        return merge_keyed(shard, dist, torch, group, gather_to_rank0, prepartitioned)
    return merge_dense(shard, dist, torch, group)


# ------------------------------------------------------------------------------ HIP backend
class HipShard:
    """ShardOps over the C-ABI: the result storage is a torch tensor (so RCCL can see it)
    handed to the library as a caller-owned output buffer."""

    def __init__(self, torch, qmd: capi.QMD, device_id: int, buf=None, handle=None):
        self._torch = torch
        self._lib = capi.load_library()
        self._qmd = qmd
        self.device_id = device_id
        self._buf = buf if buf is not None else HipShard._alloc(torch, self._lib, qmd, device_id)
        if handle is None:
            h = C.c_void_p()
            check(self._lib.mi355q_result_create(C.byref(qmd), device_id, int(self._buf.data_ptr()),
                                                 C.byref(h)), "result_create")
            handle = h.value
        self.handle = handle

    @staticmethod
    def _alloc(torch, lib, q: capi.QMD, device_id: int):
        """[entry_count, row quads] for a row-wise descriptor, flat quads for a columnar one."""
        if q.output_columnar:
            return torch.empty(lib.mi355q_qmd_buffer_bytes(C.byref(q)) // 8, dtype=torch.int64,
                               device=f"cuda:{device_id}")
        return torch.empty((q.entry_count, q.row_size // 8), dtype=torch.int64, device=f"cuda:{device_id}")

    @staticmethod
    def execute(torch, executor, ra_exe_unit, fetch_result, **kw) -> "HipShard":
        """Run the step with the result storage owned by a torch tensor."""
        q = executor.initQueryMemoryDescriptor(ra_exe_unit)
        buf = HipShard._alloc(torch, capi.load_library(), q, executor.device_id)
        rs = executor.executeWorkUnit(ra_exe_unit, fetch_result, out_buffer=int(buf.data_ptr()),
                                      allow_retry=False, **kw)
        sh = HipShard(torch, q, executor.device_id, buf, rs.handle)
        sh.report = rs.report
        rs.handle = None  # ownership moved
        return sh

    @staticmethod
    def prepare(executor, ra_exe_unit, fetch_result, scratch_bytes: int = 0, kernel_variant: int = 0, force_generic: bool = False,
                flags: int = 0, tune_cus: int = 0, tune_overlap_cus: int = 0, probe_keyed_passes: int = 0,
                tune_blocks_per_cu: int = 0):
        """Everything of a step that does not change from one execution to the next, built once: the plan and input structs of
        the C-ABI, the options, the layout descriptor (the reference compiles a step once and runs it many times too —
        the code cache, NativeCodegen.cpp).  `execute_prepared` then only allocates the result storage and calls the library."""
        lib = capi.load_library()
        plan = ra_exe_unit.to_plan()
        inp, keep = fetch_result.to_c(plan.n_cols)
        opts = executor._opts(None, None, force_generic, kernel_variant, scratch_bytes, probe_keyed_passes=probe_keyed_passes,
                              flags=flags, tune_blocks_per_cu=tune_blocks_per_cu)
        opts.tune_cus = tune_cus
        opts.tune_overlap_cus = tune_overlap_cus
        return dict(lib=lib, plan=plan, inp=inp, keep=(keep, fetch_result), opts=opts, qmd=executor.initQueryMemoryDescriptor(ra_exe_unit),
                    device_id=executor.device_id)

    @staticmethod
    def execute_prepared(torch, prep) -> "HipShard":
        """One execution of a prepared step (mi355q_execute; no retry ladder), result storage owned by a torch tensor."""
        lib, q = prep["lib"], prep["qmd"]
        buf = HipShard._alloc(torch, lib, q, prep["device_id"])
        prep["opts"].out_buffer = int(buf.data_ptr())
        out = C.c_void_p()
        rep = capi.ExecReport()
        code = lib.mi355q_execute(C.byref(prep["plan"]), C.byref(prep["inp"]), C.byref(prep["opts"]), C.byref(out), C.byref(rep))
        if code:
            raise capi.Mi355qError(code, "execute")
        sh = HipShard(torch, q, prep["device_id"], buf, out.value)
        sh.report = rep
        return sh

    def result_set(self):
        from .executor import ResultSet
        rs = ResultSet(self.handle)
        rs._keep = self  # the tensor owns the storage
        rs.free = lambda: None
        return rs

    def qmd(self) -> capi.QMD:
        return self._qmd

    def buffer(self):
        return self._buf

    def partition_rows(self, n_parts: int):
        torch = self._torch
        rq = self._qmd.row_size // 8
        # worst case every entry is live; the run lengths (read back anyway for the exchange)
        # say how many rows were written — no separate row-count pass
        rows = torch.empty((max(self._qmd.entry_count, 1), rq), dtype=torch.int64, device=self._buf.device)
        counts = torch.zeros(n_parts, dtype=torch.int64, device=self._buf.device)
        check(self._lib.mi355q_shard_partition(self.handle, n_parts, int(rows.data_ptr()),
                                               int(counts.data_ptr()), None), "shard_partition")
        c = [int(x) for x in counts.cpu().tolist()]
        return rows[:sum(c)], c

    def fresh_like(self) -> "HipShard":
        return HipShard(self._torch, self._qmd, self.device_id)

    def boundary_pads(self, world: int, pad_rows: int):
        torch = self._torch
        rq = self._qmd.row_size // 8
        pads = torch.empty((world, pad_rows, rq), dtype=torch.int64, device=self._buf.device)
        ok = torch.zeros(world, dtype=torch.int32, device=self._buf.device)
        torch.cuda.current_stream().synchronize()
        check(self._lib.mi355q_shard_pads(self.handle, world, pad_rows, int(pads.data_ptr()), int(ok.data_ptr()), None),
              "shard_pads")
        torch.cuda.synchronize(self._buf.device)
        return pads, ok

    def merge_range(self, rows, home_lo: int, home_hi: int) -> None:
        n = int(rows.shape[0])
        if n:
            self._torch.cuda.current_stream().synchronize()
            check(self._lib.mi355q_shard_merge_range(self.handle, int(rows.data_ptr()), n, home_lo, home_hi, None),
                  "shard_merge_range")

    def merge_slices(self, recv_main, recv_pads, n_src: int, home_lo: int, home_hi: int):
        """All received slices ([n_src * (home_hi - home_lo), rq]) and pads ([n_src, pad, rq]) in one LDS fold.
        True = done; False = layout not taken (self untouched); None = overflow (self incomplete)."""
        if n_src > 16:
            return False
        rq = self._qmd.row_size // 8
        step = (home_hi - home_lo) * rq * 8
        pad_rows = int(recv_pads.shape[1])
        slices = (C.c_void_p * n_src)(*[int(recv_main.data_ptr()) + i * step for i in range(n_src)])
        pads = (C.c_void_p * n_src)(*[int(recv_pads.data_ptr()) + i * pad_rows * rq * 8 for i in range(n_src)])
        self._torch.cuda.current_stream().synchronize()
        code = self._lib.mi355q_shard_merge_slices(self.handle, slices, pads, n_src, pad_rows, home_lo, home_hi, None)
        if code == capi.ERR_UNSUPPORTED:
            return False
        if code == capi.ERR_OUT_OF_SLOTS:
            return None
        check(code, "shard_merge_slices")
        return True

    def merge_rows(self, rows) -> None:
        n = int(rows.shape[0])
        if n:
            self._torch.cuda.current_stream().synchronize()
            check(self._lib.mi355q_shard_merge_rows(self.handle, int(rows.data_ptr()), n, None),
                  "shard_merge_rows")

    def reduce_from(self, other_buffer) -> None:
        """this += other (a populated buffer of the same layout on this device)."""
        self._torch.cuda.current_stream().synchronize()
        h = C.c_void_p()
        check(self._lib.mi355q_result_wrap(C.byref(self._qmd), self.device_id,
                                           int(other_buffer.data_ptr()), C.byref(h)), "result_wrap")
        try:
            check(self._lib.mi355q_result_reduce(self.handle, h.value, None), "result_reduce")
        finally:
            self._lib.mi355q_result_free(h.value)

    def free(self):
        if self.handle:
            self._lib.mi355q_result_free(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass

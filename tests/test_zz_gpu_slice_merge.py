"""The device pieces of the multi-GPU slice exchange (mi355q_shard_pads / mi355q_shard_merge_range) with
the ranks SIMULATED on one device: `world` partial tables are computed from the fragments each rank would
own (fragment f -> rank f % world), slices and pads are moved with plain tensor copies exactly as
multi_gpu._merge_keyed_by_slices's two all_to_all calls would, every rank's merge runs, and the union of
the per-rank results is compared with the oracle over all fragments; every rank must own exactly the
keys whose home slot lies in its range.  (The collective choreography itself runs over gloo in
tests/test_multi_gpu_gloo.py.)"""
from __future__ import annotations

import numpy as np
import pytest

from heavydb_amd import capi
from tests.helpers import compare_buffers, murmur3_u64

pytestmark = pytest.mark.gpu
EMPTY64 = 2**63 - 1


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.load_library()
    return torch


# slot programs: "mixed" aggregates two value columns (only the row-by-row fold takes it); "f64" and
# "nullable_i64" are programs of the partitioned family, which the LDS fold (mi355q_shard_merge_slices) takes
@pytest.mark.parametrize("fold", ["rows", "lds"])
@pytest.mark.parametrize("world,fill,prog", [(2, 0.5, "mixed"), (8, 0.5, "f64"), (5, 0.85, "nullable_i64"),
                                             (16, 0.5, "nullable_i64"), (3, 0.7, "f64")])
def test_slice_merge_with_simulated_ranks(torch_cuda, oracle, world, fill, prog, fold):
    from heavydb_amd.executor import (Executor, ExpressionRange, FetchResult, InputColDescriptor, RelAlgExecutionUnit,
                                      TargetExpr)
    from heavydb_amd.multi_gpu import SLICE_PAD_ROWS, HipShard, slice_bounds, slice_exchange_ok
    torch = torch_cuda
    rng = np.random.default_rng(17 + world)
    entries = 400_000
    n_keys = int(entries * fill)
    n, n_frags = 3_000_000, 16
    key = (rng.integers(0, n_keys, n) * 1000003 + 7).astype(np.int64)
    val = (rng.random(n) * 1000.0).astype(np.float64)
    ival = rng.integers(-10**6, 10**6, n).astype(np.int64)
    ival[rng.random(n) < 0.1] = -2**63
    descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, (n_keys - 1) * 1000003 + 7)),
             InputColDescriptor(capi.DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 1000.0)),
             InputColDescriptor(capi.INT64, True, ExpressionRange(True, -10**6, 10**6, True))]
    targets = {"mixed": [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1),
                         TargetExpr(capi.MIN, 2), TargetExpr(capi.SUM, 2)],
               "f64": [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1),
                       TargetExpr(capi.MIN, 1), TargetExpr(capi.MAX, 1)],
               "nullable_i64": [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 2), TargetExpr(capi.PROJECT_KEY),
                                TargetExpr(capi.MIN, 2), TargetExpr(capi.COUNT, 2)]}[prog]
    ra = RelAlgExecutionUnit(descs, targets, [], [0], max_groups_buffer_entry_guess=entries)
    cuts = np.linspace(0, n, n_frags + 1).astype(int) // 4 * 4
    cuts[-1] = n
    cols = [key, val, ival]
    dev = [torch.from_numpy(c).cuda() for c in cols]
    ex = Executor(0)
    shards = []
    for r in range(world):
        mine = [f for f in range(n_frags) if f % world == r]
        fr = FetchResult([[int(t.data_ptr()) + int(cuts[f]) * 8 for t in dev] for f in mine],
                         [int(cuts[f + 1] - cuts[f]) for f in mine], keepalive=dev)
        shards.append(HipShard.execute(torch, ex, ra, fr))
    q = shards[0].qmd()
    assert slice_exchange_ok(q, world) and q.entry_count == entries
    rq = q.row_size // 8
    b = slice_bounds(entries, world)
    pads, oks = zip(*[s.boundary_pads(world, SLICE_PAD_ROWS) for s in shards])
    assert all(int(o.min().item()) == 1 for o in oks)
    merged = []
    for r in range(world):
        # what the two all_to_all calls deliver to rank r: slice r of every rank's table, pad r of every rank
        recv_main = torch.cat([s.buffer()[b[r]:b[r + 1]] for s in shards]).contiguous()
        recv_pads = torch.cat([p[r] for p in pads]).contiguous()
        out = shards[r].fresh_like()
        if fold == "lds":
            took = out.merge_slices(recv_main, recv_pads.view(world, SLICE_PAD_ROWS, rq), world, b[r], b[r + 1])
            assert took is (prog != "mixed")
            if not took:
                continue
        else:
            out.merge_range(recv_main, b[r], b[r + 1])
            out.merge_range(recv_pads.view(-1, rq), b[r], b[r + 1])
        torch.cuda.synchronize()
        rows = out.buffer().cpu().numpy()
        live = rows[rows[:, 0] != EMPTY64]
        home = (murmur3_u64(live[:, 0]) % np.uint64(entries)).astype(np.int64)
        assert ((home >= b[r]) & (home < b[r + 1])).all()
        merged.append(live)
    if not merged:
        return   # the LDS fold declined the layout on every rank (asserted above)
    allrows = np.concatenate(merged)
    assert len(np.unique(allrows[:, 0])) == len(allrows)      # every key on exactly one rank
    got = np.full((entries, rq), 0, dtype=np.int64)
    got[:, 0] = EMPTY64
    got[:len(allrows)] = allrows                               # compare_buffers matches baseline tables as key -> slots maps
    frags = [[c[cuts[f]:cuts[f + 1]] for c in cols] for f in range(n_frags)]
    qo, want, code = oracle.execute(ra.to_plan(), frags, n_threads=4)
    assert code == 0
    compare_buffers(qo, want, got.reshape(-1), 1e-9)

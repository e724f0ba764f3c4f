#!/usr/bin/env python
"""Times the device pieces of the 8-GPU keyed merge on ONE device (what one rank does per step at cfg5):
8 partial tables of the headline query are computed from 1/8 of the rows each (fragment f -> rank f % 8),
then rank 0's share of the slice exchange is replayed with tensor copies standing in for the two
all_to_all calls: mi355q_shard_pads on its own table, the fold of the 8 received slices + pads
(mi355q_shard_merge_range), and — for comparison — the general path's mi355q_shard_partition +
mi355q_shard_merge_rows.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from heavydb_amd import capi, synth
    from heavydb_amd.executor import Executor
    from heavydb_amd.multi_gpu import SLICE_PAD_ROWS, HipShard, slice_bounds
    capi.load_library()
    world = 8
    total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000_000
    ex = Executor(0)
    shards = []
    step_ms = []
    for r in range(world):
        ra, fr, info = synth.cfg3(torch, total, r, world, 0, filtered=True)
        HipShard.execute(torch, ex, ra, fr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sh = HipShard.execute(torch, ex, ra, fr)
        torch.cuda.synchronize()
        step_ms.append((time.perf_counter() - t0) * 1e3)
        shards.append(sh)
        del fr
        torch.cuda.empty_cache()
    q = shards[0].qmd()
    rq = q.row_size // 8
    b = slice_bounds(q.entry_count, world)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    pads_ms, (pads0, ok0) = timed(lambda: shards[0].boundary_pads(world, SLICE_PAD_ROWS))
    pads = [s.boundary_pads(world, SLICE_PAD_ROWS)[0] for s in shards]
    recv_main = torch.cat([s.buffer()[b[0]:b[1]] for s in shards]).contiguous()
    recv_pads = torch.cat([p[0] for p in pads]).contiguous()

    def fold():
        out = shards[0].fresh_like()
        out.merge_range(recv_main, b[0], b[1])
        out.merge_range(recv_pads.view(-1, rq), b[0], b[1])
        return out
    fold_ms, out = timed(fold)
    live = int((out.buffer()[:, 0] != 2**63 - 1).sum().item())

    took = []

    def fold_lds():
        o = shards[0].fresh_like()
        took.append(o.merge_slices(recv_main, recv_pads.view(world, SLICE_PAD_ROWS, rq), world, b[0], b[1]))
        return o
    lds_ms, out_lds = timed(fold_lds)
    live_lds = int((out_lds.buffer()[:, 0] != 2**63 - 1).sum().item())
    fresh_ms, _ = timed(lambda: shards[0].fresh_like())

    def general():
        rows, counts = shards[0].partition_rows(world)
        o = shards[0].fresh_like()
        o.merge_rows(rows[:counts[0]].repeat(world, 1))   # as many rows as 8 peers would send
        return o
    gen_ms, _ = timed(general)
    print(json.dumps({"world": world, "rows_per_rank": total // world, "step_ms_per_rank": round(sum(step_ms) / world, 2),
                      "slice_path_ms": {"pads": round(pads_ms, 3), "fold_slices_and_pads_incl_fresh_table": round(fold_ms, 3),
                                        "lds_fold_incl_fresh_table": round(lds_ms, 3), "lds_fold_took": bool(took and took[-1]),
                                        "groups_after_lds_fold": live_lds, "fresh_table_alone": round(fresh_ms, 3),
                                        "exchange_bytes_per_pair": int((b[1] - b[0]) * q.row_size)},
                      "general_path_ms": {"partition_plus_merge_rows_incl_fresh_table": round(gen_ms, 3)},
                      "groups_owned_by_rank0": live, "pads_ok": int(ok0.min().item())}))


if __name__ == "__main__":
    main()

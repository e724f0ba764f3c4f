#!/usr/bin/env python
"""Times the per-rank pieces of the keyed multi-GPU merge on ONE GPU (the collective itself
cannot run here): compaction of a 10 M-group table into `world` runs, and folding `world`
received runs (same total row count) into a fresh table."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heavydb_amd import synth
from heavydb_amd.executor import Executor
from heavydb_amd.multi_gpu import HipShard

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_250_000_000
ra, fr, info = synth.cfg3(torch, rows, filtered=True)
ex = Executor(0)
sh = HipShard.execute(torch, ex, ra, fr)
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter(); part_rows, counts = sh.partition_rows(world); torch.cuda.synchronize(); t1 = time.perf_counter()
    out = sh.fresh_like(); torch.cuda.synchronize(); t2 = time.perf_counter()
    out.merge_rows(part_rows); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"world {world}: live rows {part_rows.shape[0]}  partition {1e3*(t1-t0):.2f} ms  fresh table {1e3*(t2-t1):.2f} ms  "
          f"merge {1e3*(t3-t2):.2f} ms  (all-to-all payload per rank {part_rows.shape[0]*part_rows.shape[1]*8*(world-1)/world/1e6:.0f} MB)")
    out.free()

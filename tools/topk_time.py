#!/usr/bin/env python
"""Times ORDER BY COUNT(*) DESC LIMIT k on the 10 M-group result of the headline query."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heavydb_amd import synth
from heavydb_amd.executor import Executor

ra, fr, info = synth.cfg3(torch, 1_000_000_000, filtered=True)
rs = Executor(0).executeWorkUnit(ra, fr)
q = rs.getQueryMemDesc()
rq = q.row_size // 8
for k in (10, 100, 4096):
    out = torch.empty((k, rq), dtype=torch.int64, device="cuda")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = rs.sort(1, k, int(out.data_ptr()), desc=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"top-{k} by COUNT(*) DESC over {q.entry_count} entries ({rs.rowCount()} groups): {dt*1e3:.3f} ms; best counts {out[:3,1].tolist()}")

# the FULL sort (mi355q_result_sort: one device radix sort of (key, entry) pairs per order entry — rocPRIM, where the
# reference calls thrust::sort_by_key — then a row gather), every live row, one and two order entries
n_live = rs.rowCount()
out_all = torch.empty((n_live, rq), dtype=torch.int64, device="cuda")
for order, tag in (([(1, True, False)], "COUNT(*) DESC"), ([(1, True, False), (0, False, False)], "COUNT(*) DESC, key ASC")):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = rs.sort_by(order, int(out_all.data_ptr()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"full sort ORDER BY {tag} over {q.entry_count} entries ({n} rows out, {n * q.row_size / 1e6:.0f} MB gathered): {dt*1e3:.3f} ms")

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

#!/bin/bash
out=gpurun_out/r02h
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_slice_merge.py -m gpu -q -p no:cacheprovider > $out/pytest_slice.log 2>&1
echo "pytest slice exit $?"; tail -6 $out/pytest_slice.log | cut -c1-300
cat > /tmp/l2exp.py <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from heavydb_amd import capi, synth
from heavydb_amd.executor import Executor
from heavydb_amd.multi_gpu import HipShard
capi.load_library()
ra, fr, info = synth.cfg4(torch, 3_200_000_000, sum_dim=True)
ex = Executor(0)
for var in ["A", "B", "C", "D", "A"]:
    os.environ["MI355Q_PROBE_L2_VARIANT"] = var
    sh = HipShard.execute(torch, ex, ra, fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sh = HipShard.execute(torch, ex, ra, fr)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    r = sh.report
    print(json.dumps({"variant": var, "ms_per_step": round(ms, 2), "scatter_ms": round(float(r.kernel_ms) / max(int(r.n_launches), 1), 2),
                      "chunks": int(r.n_launches), "slots": [int(x) for x in sh.result_set().getStorage().reshape(-1)[:2]]}), flush=True)
PY
timeout 600 python /tmp/l2exp.py > $out/l2exp.jsonl 2> $out/l2exp.err; cat $out/l2exp.jsonl; tail -3 $out/l2exp.err

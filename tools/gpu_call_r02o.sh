#!/bin/bash
out=gpurun_out/r02o
mkdir -p $out
export TMPDIR=/tmp
cat > /tmp/hexp.py <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from heavydb_amd import capi, synth
from heavydb_amd.executor import Executor
from heavydb_amd.multi_gpu import HipShard
capi.load_library()
ra, fr, info = synth.cfg3(torch, 10_000_000_000, filtered=True)
ex = Executor(0)
for off in [False, True, False, True]:
    if off: os.environ["MI355Q_NO_PAIR_RENDEZVOUS"] = "1"
    else: os.environ.pop("MI355Q_NO_PAIR_RENDEZVOUS", None)
    sh = HipShard.execute(torch, ex, ra, fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sh = HipShard.execute(torch, ex, ra, fr)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    r = sh.report
    print(json.dumps({"rendezvous": not off, "ms_per_step": round(ms, 2), "chunks": int(r.n_launches),
                      "scatter_ms": round(float(r.kernel_ms) / max(int(r.n_launches), 1), 2)}), flush=True)
os.environ.pop("MI355Q_NO_PAIR_RENDEZVOUS", None)
os.environ["MI355Q_TRACE"] = "1"
sh = HipShard.execute(torch, ex, ra, fr)
torch.cuda.synchronize()
PY
timeout 900 python /tmp/hexp.py > $out/hexp.jsonl 2> $out/hexp.err; cat $out/hexp.jsonl; grep -a Mcycles $out/hexp.err | tail -2
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $out/pytest_gpu.log | cut -c1-300

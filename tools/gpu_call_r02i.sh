#!/bin/bash
out=gpurun_out/r02i
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_join_probe.py -m gpu -q -p no:cacheprovider > $out/pytest_probe.log 2>&1
echo "pytest probe exit $?"; tail -4 $out/pytest_probe.log | cut -c1-300
sed -i 's/for var in \["A", "B", "C", "D", "A"\]/for var in ["8", "4", "8"]/' /tmp/l2exp.py 2>/dev/null
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
for var in ["8", "4", "8"]:
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
timeout 600 python /tmp/l2exp.py > $out/l2exp.jsonl 2> $out/l2exp.err; cat $out/l2exp.jsonl
unset MI355Q_PROBE_L2_VARIANT
for extra in "--sum-dim"; do
  tag=$(echo "cfg4$extra" | tr -d ' ' | tr -- '-' '_')
  timeout 900 python bench.py --config cfg4 $extra --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], d["roofline"]["whole_step_frac"], d["roofline"]["avg_launch_ms"], d["config"].get("variant"), d.get("verify"))
except Exception as e: print("$tag failed", e, open("$out/bench_$tag.err").read()[-600:])
PY
done

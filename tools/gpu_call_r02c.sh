#!/bin/bash
# Round 2, third GPU call: payload probe (joins reading the inner side / one-to-many / LEFT), gather
# microbenchmark, feature bench, default bench (3 chunks).
out=gpurun_out/r02c
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_join_probe.py -m gpu -q -x -p no:cacheprovider > $out/pytest_probe.log 2>&1
echo "pytest probe exit $?"; tail -15 $out/pytest_probe.log | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_zz_gpu_join_probe.py > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $out/pytest_gpu.log | cut -c1-300
timeout 300 ./tools/microbench/gather > $out/gather.txt 2>&1; cat $out/gather.txt
timeout 900 python tools/feature_bench.py --rows 1e9 --steps 3 > $out/feature_bench.jsonl 2> $out/feature_bench.err
python - <<PY
import json
for l in open("$out/feature_bench.jsonl"):
    try:
        d=json.loads(l); print(d.get("shape","?")[:90], d.get("ms_per_step"), d.get("kernel"))
    except Exception: pass
PY
timeout 900 python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
python - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("cfg3f", d["ms_per_step"], d["roofline"]["whole_step_frac"], d["roofline"]["avg_launch_ms"])
PY

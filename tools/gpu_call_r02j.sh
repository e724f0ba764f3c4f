#!/bin/bash
out=gpurun_out/r02j
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_join_probe.py tests/test_zz_gpu_arrow.py -m gpu -q -p no:cacheprovider > $out/pytest_new.log 2>&1
echo "pytest exit $?"; tail -12 $out/pytest_new.log | cut -c1-300
timeout 900 python bench.py --config cfg4 --sum-dim --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_cfg4b.json 2> $out/bench_cfg4b.err
python - <<PY
import json
try:
    d=json.loads(open("$out/bench_cfg4b.json").read().strip().splitlines()[-1])
    print("cfg4 sum_dim", d["ms_per_step"], d["roofline"]["whole_step_frac"], d["roofline"]["avg_launch_ms"], d["config"].get("variant"), d.get("verify"))
except Exception as e: print("failed", e, open("$out/bench_cfg4b.err").read()[-600:])
PY

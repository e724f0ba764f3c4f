#!/bin/bash
out=gpurun_out/r02q
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider -k "cfg2 or perfect or case" > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -4 $out/pytest.log | cut -c1-300
timeout 600 python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err
python - <<PY
import json
d=json.loads(open("$out/bench_cfg2.json").read().strip().splitlines()[-1])
print("cfg2", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
timeout 900 python tools/merge_time.py 1e10 > $out/merge_time.json 2> $out/merge_time.err; cat $out/merge_time.json; tail -2 $out/merge_time.err

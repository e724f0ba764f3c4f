#!/bin/bash
out=gpurun_out/r02r
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_slice_merge.py -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -15 $out/pytest.log | cut -c1-300
timeout 900 python tools/merge_time.py 1e10 > $out/merge_time.json 2> $out/merge_time.err; cat $out/merge_time.json; tail -2 $out/merge_time.err
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
python - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("cfg3f", d["ms_per_step"], d["roofline"]["whole_step_frac"])
PY
echo finished

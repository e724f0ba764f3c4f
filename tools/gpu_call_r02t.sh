#!/bin/bash
out=gpurun_out/r02t
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider -k "cfg2 or cfg1 or perfect or case" > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -4 $out/pytest.log | cut -c1-300
for c in cfg2 cfg1; do
timeout 600 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err
python - <<PY
import json
d=json.loads(open("$out/bench_$c.json").read().strip().splitlines()[-1])
print("$c", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
echo finished

#!/bin/bash
out=gpurun_out/r02x
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_join_probe.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "probe or join" > $out/pytest_join.log 2>&1
echo "pytest join exit $?"; tail -3 $out/pytest_join.log | cut -c1-300
for extra in "--sparse" "--sparse --sum-dim"; do
  tag=$(echo "cfg4$extra" | tr -d ' ' | tr -- '-' '_')
  timeout 900 python bench.py --config cfg4 $extra --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], d["roofline"]["whole_step_frac"], d["roofline"]["avg_launch_ms"], d["config"].get("kernel"), d["config"].get("variant"), d.get("verify"))
except Exception as e: print("$tag failed", e, open("$out/bench_$tag.err").read()[-600:])
PY
done
echo finished

#!/bin/bash
out=gpurun_out/r02f
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_join_probe.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "probe or join" > $out/pytest_join.log 2>&1
echo "pytest join exit $?"; tail -3 $out/pytest_join.log | cut -c1-300
for extra in "" "--sum-dim"; do
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
timeout 600 python tools/feature_bench.py --rows 1e9 --steps 3 --only "f2" > $out/feature_bench.jsonl 2> $out/feature_bench.err
python - <<PY
import json
for l in open("$out/feature_bench.jsonl"):
    try:
        d=json.loads(l); print(d.get("shape","?")[:90], round(d.get("ms_per_step"),2), d.get("kernel"))
    except Exception: pass
PY
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg4b -- python bench.py --config cfg4 --sum-dim --steps 2 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg4_sum_dim_kernel_stats.csv \;
rm -rf $out/trace
python - <<PY
import csv
for i,r in enumerate(csv.reader(open("$out/cfg4_sum_dim_kernel_stats.csv"))):
    if i<4: print(r[0][:50], r[1:4])
PY

#!/bin/bash
out=gpurun_out/r02w
mkdir -p $out
export TMPDIR=/tmp
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
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg4s -- python bench.py --config cfg4 --sparse --sum-dim --steps 2 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg4_sparse_sum_dim_kernel_stats.csv \;
rm -rf $out/trace
python - <<PY
import csv
for i,r in enumerate(csv.reader(open("$out/cfg4_sparse_sum_dim_kernel_stats.csv"))):
    if i<6: print(r[0][:60], r[1:4])
PY
echo finished

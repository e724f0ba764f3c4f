#!/bin/bash
out=gpurun_out/r02y
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg4s -- python bench.py --config cfg4 --sparse --sum-dim --steps 2 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg4_sparse_sum_dim_kernel_stats.csv \;
rm -rf $out/trace
python - <<PY
import csv
for i,r in enumerate(csv.reader(open("$out/cfg4_sparse_sum_dim_kernel_stats.csv"))):
    if i<6: print(r[0][:60], r[1:4])
PY
grep '"metric"' $out/trace.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"
echo finished

#!/bin/bash
# kernel-trace stats for the partitioned variants (1 B rows)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
for v in 2 3; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof/v$v -o cfg3f -- python bench.py --config cfg3f --rows 1e9 --variant $v --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/v$v.log 2>&1
  f=$(find gpurun_out/prof/v$v -name "*kernel_stats.csv" | head -1)
  echo "== variant $v: $f"; head -12 "$f" | cut -c1-200
done

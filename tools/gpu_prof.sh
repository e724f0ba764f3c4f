#!/bin/bash
# kernel-trace stats for one bench command: gpu_prof.sh <tag> <bench args...>
tag=$1; shift
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/$tag -o $tag -- python bench.py "$@" --no-cpu-baseline > gpurun_out/prof/$tag.log 2>&1
grep '"metric"' gpurun_out/prof/$tag.log | cut -c1-600
db=$(find gpurun_out/prof/$tag -name "*.db" | head -1)
python tools/rocpd_stats.py "$db" | tee gpurun_out/prof/$tag.stats.txt

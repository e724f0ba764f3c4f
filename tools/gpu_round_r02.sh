#!/bin/bash
# Round 2 profile set (profiles/r02_*): gpu_round_profiles.sh + the plain bench line, cfg4 Query B and sparse variants, Query B kernel table
export TMPDIR=/tmp
bash tools/gpu_round_profiles.sh r02
out=gpurun_out/round_r02
timeout 900 python bench.py > $out/bench_cfg3f.log 2>&1
grep '"metric"' $out/bench_cfg3f.log | cut -c1-300
for extra in "--sum-dim" "--sparse" "--sparse --sum-dim"; do
  tag=$(echo "cfg4$extra" | tr -d ' ' | tr -- '-' '_')
  timeout 900 python bench.py --config cfg4 $extra --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_$tag.log 2>&1
  grep '"metric"' $out/bench_$tag.log | cut -c1-200
done
rocprofv3 --kernel-trace --stats -f csv -d $out/trace4 -o cfg4b -- python bench.py --config cfg4 --sum-dim --steps 2 --warmup 1 --no-cpu-baseline > $out/trace4.log 2>&1
find $out/trace4 -name "*kernel_stats.csv" -exec cp {} $out/cfg4_sum_dim_kernel_stats.csv \;
rm -rf $out/trace4
echo finished

#!/bin/bash
# Everything profiles/ needs for one round, in one GPU session:
#   1. rocprofv3 --kernel-trace --stats of the default bench command (cfg3-filtered, 10 B rows)
#   2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE / request mix / L2 hit) of the same command
#   3. bench lines of the other configs
# usage: gpu_round_profiles.sh <round-tag> [rows]
tag=${1:-r01}; rows=${2:-}
extra=""; [ -n "$rows" ] && extra="--rows $rows"
out=gpurun_out/round_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg3f -- python bench.py $extra --steps 3 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg3f_kernel_stats.csv \;
find $out/trace -name "*kernel_trace.csv" -exec rm -f {} \;
head -8 $out/cfg3f_kernel_stats.csv | cut -c1-160
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_$i -o pmc -- python bench.py $extra --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$i.log 2>&1
  echo "== $grp" >> $out/cfg3f_pmc.txt
  python tools/rocpd_stats.py $out/pmc_$i/pmc_results.db | sed -n '/PMC/,$p' | grep -E "k_part|k_scan|k_perfect|k_baseline|k_join|k_generate|k_spill" >> $out/cfg3f_pmc.txt
  rm -rf $out/pmc_$i
done
cat $out/cfg3f_pmc.txt | cut -c1-170
for cfg in cfg1 cfg2 cfg3 cfg4; do
  timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --verify > $out/bench_$cfg.log 2>&1
  grep '"metric"' $out/bench_$cfg.log | cut -c1-1600
done
rm -rf $out/trace

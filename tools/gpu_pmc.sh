#!/bin/bash
# HBM traffic counters for one bench command, one --pmc pass per counter group (the guide:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with other trace domains)
# usage: gpu_pmc.sh <tag> <bench args...>
tag=$1; shift
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/pmc/${tag}_$i -o pmc -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/${tag}_$i.log 2>&1
  echo "== $grp"; python tools/rocpd_stats.py gpurun_out/pmc/${tag}_$i/pmc_results.db | sed -n '/PMC/,$p' | grep -E "k_part|k_scan|k_perfect|k_baseline|k_join" 
done

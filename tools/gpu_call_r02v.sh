#!/bin/bash
out=gpurun_out/r02v
mkdir -p $out
export TMPDIR=/tmp
for w in 0 2; do
  export MI355Q_PAIR_WINDOW=$w
  i=0
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_${w}_$i -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_${w}_$i.log 2>&1
    echo "== window $w: $grp" >> $out/pmc.txt
    python tools/rocpd_stats.py $out/pmc_${w}_$i/pmc_results.db | sed -n '/PMC/,$p' | grep -E "k_part" >> $out/pmc.txt
    rm -rf $out/pmc_${w}_$i
  done
done
cat $out/pmc.txt | cut -c1-200
echo finished

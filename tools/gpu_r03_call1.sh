#!/bin/bash
# Round 3, first GPU call: L2-atomic microbenchmark, the new full-size oracle parity tests + the Arrow test,
# five back-to-back default bench runs (box-internal spread), PMC passes of the join-probe kernels.
out=gpurun_out/r03a
mkdir -p $out
export TMPDIR=/tmp
timeout 120 ./tools/microbench/l2_atomics > $out/microbench_l2_atomics.txt 2>&1; echo "l2_atomics exit $?"; cat $out/microbench_l2_atomics.txt
free -g | head -2; nproc
timeout 1500 python -m pytest tests/test_zz_gpu_baseline_sizes.py tests/test_zz_gpu_arrow.py -m gpu -q -x -s -p no:cacheprovider \
   -k "full_size_vs_oracle or arrow" > $out/pytest_new.log 2>&1
echo "pytest exit $?"; grep -E "oracle cfg|passed|failed|error" $out/pytest_new.log | tail -12
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/spread_$i.json 2> $out/spread_$i.err
  python - <<PY
import json
d = json.load(open("$out/spread_$i.json"))
print("spread $i", round(d["ms_per_step"], 2), "ms  scatter", round(d["roofline"]["avg_launch_ms"], 2), d["roofline"]["measured_ceiling"])
PY
done
for cfgargs in "--sparse --sum-dim" "--sum-dim"; do
  tag=$(echo $cfgargs | tr -d ' -')
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    g=$(echo $grp | cut -d' ' -f1)
    timeout 400 rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_${tag}_$g -o pmc -- python bench.py --config cfg4 $cfgargs --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_${tag}_$g.log 2>&1
    echo "== cfg4 $cfgargs : $grp" >> $out/cfg4_join_pmc.txt
    python tools/rocpd_stats.py $out/pmc_${tag}_$g/pmc_results.db | sed -n '/PMC/,$p' | grep -E "k_part|k_join|k_probe|k_outer" >> $out/cfg4_join_pmc.txt
    rm -rf $out/pmc_${tag}_$g
  done
done
cat $out/cfg4_join_pmc.txt | cut -c1-170

#!/bin/bash
out=gpurun_out/r02g
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_slice_merge.py tests/test_zz_gpu_join_probe.py -m gpu -q -p no:cacheprovider > $out/pytest_new.log 2>&1
echo "pytest new exit $?"; tail -12 $out/pytest_new.log | cut -c1-300
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
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg4b -- python bench.py --config cfg4 --sum-dim --steps 2 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg4_sum_dim_kernel_stats.csv \;
rm -rf $out/trace
python - <<PY
import csv
for i,r in enumerate(csv.reader(open("$out/cfg4_sum_dim_kernel_stats.csv"))):
    if i<4: print(r[0][:50], r[1:4])
PY
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp -d $out/pmc -o pmc -- python bench.py --config cfg4 --sum-dim --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc.log 2>&1
  python tools/rocpd_stats.py $out/pmc/pmc_results.db | sed -n '/PMC/,$p' | grep -E "k_part_probe" | cut -c1-200
  rm -rf $out/pmc
done

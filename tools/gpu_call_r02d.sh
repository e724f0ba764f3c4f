#!/bin/bash
# Round 2, fourth GPU call: full device sort, L2-mode payload probe, cfg4 variants.
out=gpurun_out/r02d
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_sort.py tests/test_zz_gpu_join_probe.py -m gpu -q -p no:cacheprovider > $out/pytest_new.log 2>&1
echo "pytest new exit $?"; tail -25 $out/pytest_new.log | cut -c1-400
for extra in "" "--sum-dim"; do
  tag=$(echo "cfg4$extra" | tr -d ' ' | tr -- '-' '_')
  timeout 900 python bench.py --config cfg4 $extra --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], d["roofline"]["whole_step_frac"], d["roofline"]["kernel"], d["config"].get("variant"), d.get("verify"))
except Exception as e: print("$tag failed", e, open("$out/bench_$tag.err").read()[-600:])
PY
done
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg4b -- python bench.py --config cfg4 --sum-dim --steps 2 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg4_sum_dim_kernel_stats.csv \;
rm -rf $out/trace
head -8 $out/cfg4_sum_dim_kernel_stats.csv | cut -c1-60,200-330

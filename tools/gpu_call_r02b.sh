#!/bin/bash
# Round 2, second GPU call: records carry {low word, hash} (phase 2 no longer re-hashes), global
# instead of flat column loads, default scratch cap 76 GB (3 chunks at 10 B rows).
out=gpurun_out/r02b
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $out/pytest_gpu.log
timeout 900 python tools/headline_experiments.py --settings "0:0,0:32,0:48" --trace "0:0" > $out/experiments.jsonl 2> $out/experiments.err
echo "experiments exit $?"; cat $out/experiments.jsonl; grep -a "Mcycles\|gave up" $out/experiments.err | tail -4
timeout 900 python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
echo "bench exit $?"; cut -c1-2500 $out/bench_default.json
for cfg in cfg1 cfg2 cfg3; do
  timeout 600 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$cfg.json").read().strip().splitlines()[-1])
    print("$cfg", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_step_frac"], d["roofline"]["kernel"])
except Exception as e: print("$cfg failed", e)
PY
done
for extra in "" "--sum-dim" "--sparse" "--sparse --sum-dim"; do
  tag=$(echo "cfg4$extra" | tr -d ' ' | tr -- '-' '_')
  timeout 900 python bench.py --config cfg4 $extra --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], d["roofline"]["whole_step_frac"], d["roofline"]["kernel"], d.get("verify"))
except Exception as e: print("$tag failed", e, open("$out/bench_$tag.err").read()[-400:])
PY
done
rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg3f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg3f_kernel_stats.csv \;
rm -rf $out/trace
head -6 $out/cfg3f_kernel_stats.csv | cut -c1-200
bash tools/gpu_pmc2.sh r02b > $out/pmc_sq.txt 2>&1; grep "INSTS_VALU\|ACTIVE_INST_VALU\|WAVE_CYCLES\|BUSY" $out/pmc_sq.txt | cut -c1-200

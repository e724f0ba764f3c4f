#!/bin/bash
out=gpurun_out/r02p
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider -k "cfg2 or cfg1 or perfect or case" > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -4 $out/pytest.log | cut -c1-300
for cfg in cfg1 cfg2; do
  timeout 600 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  python - <<PY
import json
d=json.loads(open("$out/bench_$cfg.json").read().strip().splitlines()[-1])
print("$cfg", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
bash tools/gpu_pmc2.sh r02p_cfg2 --config cfg2 > $out/pmc_cfg2.txt 2>&1; grep -i "perfect" gpurun_out/pmc/r02p_cfg2_*.log | head -2; python tools/rocpd_stats.py gpurun_out/pmc/r02p_cfg2_1/pmc_results.db | grep perfect_lds | cut -c1-200; python tools/rocpd_stats.py gpurun_out/pmc/r02p_cfg2_2/pmc_results.db | grep perfect_lds | cut -c1-200

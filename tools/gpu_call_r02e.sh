#!/bin/bash
out=gpurun_out/r02e
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_join_probe.py -m gpu -q -p no:cacheprovider -k "l2_mode" > $out/pytest_l2.log 2>&1
echo "pytest l2 exit $?"; tail -3 $out/pytest_l2.log | cut -c1-300
for pacing in 1 0; do
  if [ $pacing = 0 ]; then export MI355Q_PROBE_NO_PACING=1; else unset MI355Q_PROBE_NO_PACING; fi
  timeout 900 python bench.py --config cfg4 --sum-dim --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_cfg4b_p$pacing.json 2> $out/bench_cfg4b_p$pacing.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_cfg4b_p$pacing.json").read().strip().splitlines()[-1])
    print("cfg4 sum_dim pacing=$pacing", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("verify"))
except Exception as e: print("failed", e, open("$out/bench_cfg4b_p$pacing.err").read()[-600:])
PY
done
unset MI355Q_PROBE_NO_PACING
timeout 600 python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err
python - <<PY
import json
d=json.loads(open("$out/bench_cfg2.json").read().strip().splitlines()[-1])
print("cfg2", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY

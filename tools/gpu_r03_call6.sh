#!/bin/bash
# Round 3, call 6 (the last ~5 GPU-minutes of the round): every -m gpu test except the full-size oracle runs (unchanged
# kernels, green in call 5), the reference benchmark table at 1 B rows with the windowed LDS group-by / FLOAT keys / early
# abort, the FETCH_SIZE / WRITE_SIZE passes of the default bench for profiles/traffic.json, the 128 M-row table.
out=gpurun_out/r03f
mkdir -p $out
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
t0=$SECONDS
left() { echo $(( ${LIMIT:-265} - (SECONDS - t0) )); }
timeout 175 python -u -m pytest tests -m gpu -q -p no:cacheprovider --timeout 100 -rf --ignore=tests/test_zz_gpu_baseline_sizes.py > $out/pytest_gpu.log 2>&1
echo "pytest exit $? at $((SECONDS - t0)) s"; tail -12 $out/pytest_gpu.log | cut -c1-200
if [ $(left) -gt 60 ]; then
  timeout $(( $(left) - 15 )) python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 400 --out $out/refbench_1b.jsonl > $out/refbench_1b.log 2>&1
  echo "refbench 1B exit $? at $((SECONDS - t0)) s"
  python - <<'PY'
import json
for l in open("gpurun_out/r03f/refbench_1b.log"):
    try: d = json.loads(l)
    except Exception: continue
    print(d.get("query"), d.get("kernel"), d.get("ms", d.get("extrapolated_ms")), d.get("groups"), d.get("whole_step_frac"), "SKIPPED" if d.get("skipped") else "", d.get("error", ""))
PY
fi
for grp in FETCH_SIZE WRITE_SIZE; do
  if [ $(left) -gt 45 ]; then
    timeout $(( $(left) - 8 )) rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_$grp -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$grp.log 2>&1
    echo "pmc $grp exit $? at $((SECONDS - t0)) s"
    python tools/rocpd_stats.py $out/pmc_$grp/pmc_results.db > $out/pmc_${grp}_stats.txt 2>&1
    grep -E "k_part_scatter|k_part_aggregate" $out/pmc_${grp}_stats.txt | grep $grp | cut -c1-200
    rm -rf $out/pmc_$grp
  fi
done
if [ $(left) -gt 40 ]; then
  timeout $(( $(left) - 8 )) python tools/refbench.py --rows 128e6 --steps 3 --budget-ms 400 --out $out/refbench_128m.jsonl > $out/refbench_128m.log 2>&1
  echo "refbench 128M exit $? at $((SECONDS - t0)) s"
fi
echo "done at $((SECONDS - t0)) s"

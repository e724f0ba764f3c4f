#!/bin/bash
# round 5: ONE script for every GPU call of the round (VERDICT r04 hygiene: no per-call scripts).  usage: tools/gpu_r05.sh <step> [out-dir]
step=${1:-help}
out=${2:-gpurun_out/r05_$step}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
case $step in
  before)   # the "before" of the expression-fusion work: the five BOOLEAN-filter shapes through the k_project interpreter pass
    timeout 500 python tools/bool_filter_bench.py --rows 1e9 --steps 3 > $out/bool_filter_1b.jsonl 2> $out/bool_filter.err; echo "bool_filter exit $?"
    cat $out/bool_filter_1b.jsonl; tail -5 $out/bool_filter.err ;;
  after)    # the five BOOLEAN-filter shapes with the filter compiled into the consuming kernel (+ parity on a sample)
    timeout 500 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --verify-rows 4e6 > $out/bool_filter_1b.jsonl 2> $out/bool_filter.err; echo "bool_filter exit $?"
    cat $out/bool_filter_1b.jsonl; tail -5 $out/bool_filter.err ;;
  nga)      # the typed scan-aggregate members: NGA01-05 at 1 B rows (parity on a sample) + the refbench parity tests
    timeout 400 python tools/refbench.py --rows 1e9 --steps 3 --only NGA --verify-rows 2e6 --out $out/refbench_nga.jsonl > $out/refbench.log 2>&1; echo "refbench exit $?"
    python - <<PY
import json
for l in open("$out/refbench_nga.jsonl"):
    d=json.loads(l); print(d.get("query"), (d.get("route") or "")[:60], d.get("ms"), d.get("whole_step_frac"), d.get("verified_rows"))
PY
    ;;
  ngasweep) timeout 400 python tools/nga_sweep.py --rows 1e9 > $out/nga_sweep.jsonl 2> $out/err.log; echo "exit $?"; cat $out/nga_sweep.jsonl; tail -3 $out/err.log ;;
  cfg4a)    # BASELINE cfg4, Query A on the dense dimension: the join is a range filter on the key (driver-style line, --verify)
    timeout 900 python bench.py --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline --verify > $out/bench_cfg4a_dense.json 2> $out/err.log; echo "exit $?"
    cat $out/bench_cfg4a_dense.json; tail -3 $out/err.log ;;
  suite)    # the whole -m gpu suite (no -x: every failure listed), then smoke()
    timeout 2500 python -u -m pytest tests -m gpu -q -p no:cacheprovider ${3:-} > $out/pytest_gpu.log 2>&1
    echo "pytest exit $?"; grep -n "FAILED\|Fatal\|fault" $out/pytest_gpu.log | head -20; tail -2 $out/pytest_gpu.log | cut -c1-200
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
  proj1)    # first contact of the Projection family: the case matrix on the device, then the 1 B-row bench lines
    timeout 900 python -u -m pytest tests/test_zz_gpu_projection.py -m gpu -x -q -p no:cacheprovider -k "case_on_the_device or larger_random" > $out/pytest.log 2>&1
    echo "pytest exit $?"; tail -5 $out/pytest.log
    timeout 600 python tools/proj_bench.py --rows 1e9 --steps 3 > $out/proj_bench_1b.jsonl 2> $out/proj_bench.err; echo "bench exit $?"
    cat $out/proj_bench_1b.jsonl; tail -3 $out/proj_bench.err ;;
  proj1b)   # the 1 B-row property + oracle tests of the Projection family
    timeout 1500 python -u -m pytest tests/test_zz_gpu_projection.py -m gpu -x -q -p no:cacheprovider -k "1b_rows" > $out/pytest.log 2>&1
    echo "pytest exit $?"; tail -8 $out/pytest.log ;;
  projpmc)  # HBM traffic of the Projection kernel (separate passes per counter, as the guide prescribes): $3 = proj_bench args
    for ctr in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
      tag=$(echo $ctr | tr ' ' '_')
      rocprofv3 --kernel-trace --pmc $ctr -d $out/pmc_$tag -o pmc -- python tools/proj_bench.py --rows 1e9 --steps 0 ${3:---sel 0.99 --cols 3} > $out/pmc_$tag.log 2>&1
      python tools/rocpd_stats.py $out/pmc_$tag/pmc_results.db | sed -n '/PMC/,$p' | grep -E "k_proj" | tee -a $out/pmc_summary.txt
    done ;;
  *) echo "unknown step $step"; exit 2 ;;
esac

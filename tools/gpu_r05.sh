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
    timeout 2500 python -u -m pytest tests -m gpu -q -p no:cacheprovider "${@:3}" > $out/pytest_gpu.log 2>&1
    echo "pytest exit $?"; grep -n "FAILED\|Fatal\|fault" $out/pytest_gpu.log | head -20; tail -2 $out/pytest_gpu.log | cut -c1-200
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
  interp)   # k_project alone: the five BOOLEAN-filter shapes with MI355Q_OPT_NO_COMPILED_FILTER (every expression through the interpreter pass)
    timeout 500 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --interpreted --verify-rows 4e6 > $out/bool_filter_1b_interpreted.jsonl 2> $out/err.log; echo "exit $?"
    cut -c1-330 $out/bool_filter_1b_interpreted.jsonl; tail -5 $out/err.log
    timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o interp -- python tools/bool_filter_bench.py --rows 1e9 --steps 3 --interpreted > /dev/null 2> $out/rocprof.err
    find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/interp_kernel_stats.csv \; ; rm -rf $out/trace; cut -c1-200 $out/interp_kernel_stats.csv | head -8 ;;
  interppmc) # k_project's instruction mix and stall split (SQ counters; one pass, kernel trace only)
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc -o pmc -- python tools/bool_filter_bench.py --rows 1e9 --steps 1 --interpreted > $out/pmc.log 2>&1
    python tools/rocpd_stats.py $out/pmc/pmc_results.db > $out/pmc_stats.txt 2>&1; rm -rf $out/pmc; grep -E "k_project|k_perfect_lds|k_groupby" $out/pmc_stats.txt | cut -c1-200; tail -3 $out/pmc.log ;;
  projvar)  # the Projection family's general member: the plain shapes forced through it, expressions in registers, a join probe per row
    for v in ${3:-general expr join}; do
      timeout 400 python tools/proj_bench.py --rows 1e9 --steps 3 --sel 0.5 --cols 3 --variant $v >> $out/proj_variants_1b.jsonl 2>> $out/err.log; echo "$v exit $?"
    done
    cut -c1-420 $out/proj_variants_1b.jsonl; tail -5 $out/err.log ;;
  countonly) # SELECT g, COUNT(*) ... WHERE <filter> GROUP BY g: the NV = 0 typed members next to the run-time-role member
    timeout 300 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --count-only --verify-rows 4e6 > $out/count_only_typed.jsonl 2> $out/err.log; echo "exit $?"
    timeout 300 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --count-only --generic-member > $out/count_only_generic.jsonl 2>> $out/err.log; echo "exit $?"
    cut -c1-260 $out/count_only_typed.jsonl $out/count_only_generic.jsonl; tail -3 $out/err.log ;;
  r1bound)  # the headline with R = 1 (fewer groups): the bound of everything a cheaper second read of the records could gain
    MI355Q_TRACE=0 timeout 500 python tools/headline_r1_bound.py 1e10 > $out/headline_r1_bound.jsonl 2> $out/err.log; echo "exit $?"
    cut -c1-400 $out/headline_r1_bound.jsonl; tail -4 $out/err.log ;;
  cfg1cost) timeout 300 python tools/cfg1_cost.py > $out/cfg1_cost.txt 2>&1; echo "exit $?"; tail -40 $out/cfg1_cost.txt ;;
  final)    # the round's kept lines on ONE build: every BASELINE config (roofline + cpu_baseline + verify), the default line's
            # rocprofv3 kernel summary and FETCH / WRITE passes (-> profiles/traffic.json), the Projection / filter / NGA shapes,
            # the Projection kernel's rocprofv3 summary and counters, the reference's 57 benchmark steps at 1 B rows
    timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default exit $? $(python -c "import json; d=json.load(open('$out/bench_default.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('whole_step_frac'))")"
    for spec in "cfg1:--config cfg1 --steps 50 --warmup 5" "cfg2:--config cfg2 --steps 20 --warmup 3" "cfg3:--config cfg3" "cfg3f:--config cfg3f" "cfg4:--config cfg4" "cfg4_sum_dim:--config cfg4 --sum-dim" "cfg4_sparse:--config cfg4 --sparse" "cfg4_sparse_sum_dim:--config cfg4 --sparse --sum-dim"; do
      tag=${spec%%:*}; args=${spec#*:}
      timeout 600 python bench.py $args --verify > $out/bench_$tag.json 2> $out/bench_$tag.err
      echo "$tag: exit $? $(python -c "import json; d=json.load(open('$out/bench_$tag.json')); print(d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('whole_step_frac'), d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)" 2>&1)"
    done
    timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg3f -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_rocprof.json 2> $out/bench_rocprof.err
    find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg3f_kernel_stats.csv \; ; rm -rf $out/trace
    for grp in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_$grp -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$grp.log 2>&1
      python tools/rocpd_stats.py $out/pmc_$grp/pmc_results.db > $out/pmc_${grp}_stats.txt 2>&1; rm -rf $out/pmc_$grp
    done
    python tools/make_traffic_json.py $out/pmc_FETCH_SIZE_stats.txt $out/pmc_WRITE_SIZE_stats.txt 1e10 r05 && cp profiles/traffic.json $out/traffic.json
    timeout 400 python bench.py > $out/bench_default_with_traffic.json 2> $out/bench_default_with_traffic.err; echo "bench (traffic) exit $?"
    timeout 500 python tools/proj_bench.py --rows 1e9 --steps 5 > $out/proj_bench_1b.jsonl 2> $out/proj_bench.err; echo "proj bench exit $?"
    timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_proj -o proj -- python tools/proj_bench.py --rows 1e9 --steps 5 --sel 0.5 --cols 3 > $out/proj_rocprof.jsonl 2> $out/proj_rocprof.err
    find $out/trace_proj -name "*kernel_stats.csv" -exec cp {} $out/proj_sel50_cols3_kernel_stats.csv \; ; rm -rf $out/trace_proj
    for grp in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_proj_$grp -o pmc -- python tools/proj_bench.py --rows 1e9 --steps 0 --sel 0.5 --cols 3 > $out/pmc_proj_$grp.log 2>&1
      python tools/rocpd_stats.py $out/pmc_proj_$grp/pmc_results.db | sed -n '/PMC/,$p' | grep -E "k_proj" > $out/pmc_proj_${grp}_stats.txt 2>&1; rm -rf $out/pmc_proj_$grp
      cat $out/pmc_proj_${grp}_stats.txt | cut -c1-170
    done
    timeout 400 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --verify-rows 4e6 > $out/bool_filter_1b.jsonl 2> $out/bool_filter.err; echo "bool filter exit $?"
    timeout 300 python tools/nga_sweep.py --rows 1e9 --bpc 0 > $out/nga_1b.jsonl 2> $out/nga.err; echo "nga exit $?"
    timeout 700 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_1b.jsonl > $out/refbench_1b.log 2>&1; echo "refbench 1B exit $?"
    head -3 $out/cfg3f_kernel_stats.csv ;;
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

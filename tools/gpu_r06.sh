#!/bin/bash
# round 6: ONE script for every GPU call of the round.  usage: tools/gpu_r06.sh <step> [out-dir] [extra args]
step=${1:-help}
out=${2:-gpurun_out/r06_$step}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
case $step in
  atoms)    # program atoms (regprog.h): the BOOLEAN-filter shapes incl. the arithmetic ones, compiled vs interpreted; device parity tests
    timeout 600 python -u -m pytest tests/test_zz_gpu_typed_filters.py -m gpu -x -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $out/pytest.log
    timeout 500 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --verify-rows 4e6 > $out/bool_filter_1b.jsonl 2> $out/bool_filter.err; echo "bool_filter exit $?"
    cut -c1-300 $out/bool_filter_1b.jsonl; tail -5 $out/bool_filter.err
    timeout 400 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --interpreted --only guarded_div,sum_gt,col_lt_col,affine > $out/bool_filter_1b_interpreted.jsonl 2>> $out/bool_filter.err; echo "interp exit $?"
    cut -c1-300 $out/bool_filter_1b_interpreted.jsonl
    timeout 400 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --prepass --verify-rows 4e6 --only guarded_div,sum_gt,col_lt_col,affine > $out/bool_filter_1b_prepass.jsonl 2>> $out/bool_filter.err; echo "prepass exit $?"
    cut -c1-300 $out/bool_filter_1b_prepass.jsonl ;;
  atomsprof) # per-kernel times of the program-atom shapes (rocprofv3 kernel trace)
    cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
    timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o atoms -- python tools/bool_filter_bench.py --rows 1e9 --steps 3 --only ${3:-guarded_div,sum_gt,col_lt_col,affine} > $out/bench.jsonl 2> $out/rocprof.err
    find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/atoms_kernel_stats.csv \; ; rm -rf $out/trace; cut -c1-160 $out/atoms_kernel_stats.csv | head -12; cut -c1-200 $out/bench.jsonl ;;
  atomspmc) # k_filter_mask's instruction mix and stall split (SQ counters; one pass, kernel trace only)
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc -o pmc -- python tools/bool_filter_bench.py --rows 1e9 --steps 1 --prepass --only ${3:-guarded_div} > $out/pmc.log 2>&1
    python tools/rocpd_stats.py $out/pmc/pmc_results.db > $out/pmc_stats.txt 2>&1; rm -rf $out/pmc; grep -E "k_filter_mask|k_groupby" $out/pmc_stats.txt | cut -c1-260; tail -3 $out/pmc.log ;;
  projexpr) # Projection with expression targets: forms in the fast member next to the plain shape and the general member's interpreter
    timeout 300 python tools/proj_bench.py --rows 1e9 --steps 3 --sel 0.5 --cols 3 --variant plain > $out/proj_plain.jsonl 2> $out/err.log; echo "plain exit $?"
    timeout 300 python tools/proj_bench.py --rows 1e9 --steps 3 --sel 0.5 --cols 3 --variant expr > $out/proj_expr.jsonl 2>> $out/err.log; echo "expr exit $?"
    timeout 300 python tools/proj_bench.py --rows 1e9 --steps 3 --sel 0.5 --cols 3 --variant expr --generic-member > $out/proj_expr_interpreted.jsonl 2>> $out/err.log; echo "expr interp exit $?"
    timeout 300 python tools/proj_bench.py --rows 1e9 --steps 3 --sel 0.5 --cols 3 --variant exprfilter > $out/proj_exprfilter.jsonl 2>> $out/err.log; echo "exprfilter exit $?"
    timeout 300 python tools/proj_bench.py --rows 1e9 --steps 3 --sel 0.5 --cols 3 --variant exprfilter --interpreted > $out/proj_exprfilter_interpreted.jsonl 2>> $out/err.log; echo "exprfilter interp exit $?"
    cut -c1-260 $out/proj_plain.jsonl $out/proj_expr.jsonl $out/proj_expr_interpreted.jsonl $out/proj_exprfilter.jsonl $out/proj_exprfilter_interpreted.jsonl; tail -3 $out/err.log ;;
  sort)     # the hand-written one-sweep radix sort: device parity tests, then the 20 M-entry timings (rocPRIM's were 2.35 / 4.26 ms)
    timeout 900 python -u -m pytest tests/test_zz_gpu_sort.py -m gpu -x -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $out/pytest.log
    timeout 600 python tools/topk_time.py > $out/sort_time.txt 2>&1; echo "time exit $?"; cat $out/sort_time.txt
    timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o sort -- python tools/topk_time.py > /dev/null 2> $out/rocprof.err
    find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/sort_kernel_stats.csv \; ; rm -rf $out/trace; cut -c1-150 $out/sort_kernel_stats.csv | head -12 ;;
  proj1n)   # one-to-many joins under a Projection: device parity (oracle + SQLite), the binding's query 7, the bench lines
    timeout 900 python -u -m pytest tests/test_zz_gpu_projection.py tests/test_integration_glue.py -m gpu -q -p no:cacheprovider -k "join or glue" > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $out/pytest.log
    for v in join join1n; do
      timeout 400 python tools/proj_bench.py --rows 1e9 --steps 3 --sel 0.5 --cols 3 --variant $v >> $out/proj_join_variants_1b.jsonl 2>> $out/err.log; echo "$v exit $?"
    done
    cut -c1-330 $out/proj_join_variants_1b.jsonl; tail -3 $out/err.log ;;
  projlow)  # the Projection family's low end: device parity of the whole projection file, then the plain shapes at 0.1 / 1 / 10 % and the
            # round's usual three, the fast member as one fused launch and as the split route (k_proj_mask + scan + pass B)
    timeout 1500 python -u -m pytest tests/test_zz_gpu_projection.py -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $out/pytest.log
    for route in fused split; do
      timeout 600 python tools/proj_bench.py --rows 1e9 --steps 5 --sel ${3:-0.001,0.01,0.1,0.5,0.99} --cols 1,3,6 --route $route > $out/proj_low_1b_$route.jsonl 2>> $out/err.log; echo "bench $route exit $?"
    done
    python -c "
import json
rows = {}
for route in ('fused', 'split'):
    for l in open('$out/proj_low_1b_%s.jsonl' % route):
        d = json.loads(l); rows.setdefault(d['shape'], {})[route] = (d['ms'], d['frac'], d['must_move_frac'], d['route'])
for k, v in rows.items(): print(k, v)"; tail -3 $out/err.log ;;
  projprof) # per-kernel times of the fast member's split route (rocprofv3 kernel trace)
    cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
    timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o proj -- python tools/proj_bench.py --rows 1e9 --steps 3 --sel ${3:-0.001} --cols ${4:-1} --route split > $out/bench.jsonl 2> $out/rocprof.err
    find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/proj_kernel_stats.csv \; ; rm -rf $out/trace; cut -c1-160 $out/proj_kernel_stats.csv | head -12; cut -c1-200 $out/bench.jsonl ;;
  cfg4)     # BASELINE cfg4 on a dimension WITH HOLES (the general perfect probe), Query A and B, driver-style lines + the 2 B-row parity tests
    timeout 900 python -u -m pytest tests/test_zz_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider -k "holes" > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $out/pytest.log
    for spec in "cfg4a_holes:--config cfg4 --holes 16" "cfg4b_holes:--config cfg4 --holes 16 --sum-dim"; do
      tag=${spec%%:*}; args=${spec#*:}
      timeout 600 python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --verify > $out/bench_$tag.json 2> $out/bench_$tag.err
      echo "$tag: exit $? $(python -c "import json; d=json.load(open('$out/bench_$tag.json')); print(d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('whole_step_frac'), d['config'].get('route'))" 2>&1)"
    done ;;
  final)    # the round's kept lines on ONE build: every BASELINE config (roofline + cpu_baseline + verify), the default line's
            # rocprofv3 kernel summary and FETCH / WRITE passes (-> profiles/traffic.json), the Projection / filter / NGA / sort shapes,
            # the reference's 57 benchmark steps at 1 B rows
    timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default exit $? $(python -c "import json; d=json.load(open('$out/bench_default.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('whole_step_frac'))")"
    for spec in "cfg1:--config cfg1 --steps 50 --warmup 5" "cfg2:--config cfg2 --steps 20 --warmup 3" "cfg3:--config cfg3" "cfg3f:--config cfg3f" "cfg4:--config cfg4" "cfg4_sum_dim:--config cfg4 --sum-dim" "cfg4_holes:--config cfg4 --holes 16" "cfg4_holes_sum_dim:--config cfg4 --holes 16 --sum-dim" "cfg4_sparse:--config cfg4 --sparse" "cfg4_sparse_sum_dim:--config cfg4 --sparse --sum-dim"; do
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
    python tools/make_traffic_json.py $out/pmc_FETCH_SIZE_stats.txt $out/pmc_WRITE_SIZE_stats.txt 1e10 r06 && cp profiles/traffic.json $out/traffic.json
    timeout 400 python bench.py > $out/bench_default_with_traffic.json 2> $out/bench_default_with_traffic.err; echo "bench (traffic) exit $?"
    timeout 500 python tools/proj_bench.py --rows 1e9 --steps 5 > $out/proj_bench_1b.jsonl 2> $out/proj_bench.err; echo "proj bench exit $?"
    timeout 400 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --verify-rows 4e6 > $out/bool_filter_1b.jsonl 2> $out/bool_filter.err; echo "bool filter exit $?"
    timeout 300 python tools/nga_sweep.py --rows 1e9 --bpc 0 > $out/nga_1b.jsonl 2> $out/nga.err; echo "nga exit $?"
    timeout 300 python tools/topk_time.py > $out/sort_time.txt 2>&1; echo "sort exit $?"
    timeout 700 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_1b.jsonl > $out/refbench_1b.log 2>&1; echo "refbench 1B exit $?"
    timeout 500 python tools/refbench.py --rows 1e9 --steps 3 --flags 1024 --only S00,PHS005,PHS006,PHS007,PHM004,PHM005,PHM006,MSPHS,MSPHM,MSBS003,MSBS004,MSBS005,BH005,BH006 --out $out/refbench_1b_idx_plain_records.jsonl > $out/refbench_1b_plain.log 2>&1; echo "refbench (plain idx records) exit $?"
    python -c "
import json, statistics
rows=[json.loads(l) for l in open('$out/refbench_1b.jsonl')]
print('refbench:', len(rows), 'steps', round(sum(r.get('ms', 0) for r in rows), 1), 'ms; skipped', [r['query'] for r in rows if 'ms' not in r], 'median', statistics.median(r['whole_step_frac'] for r in rows))"
    head -3 $out/cfg3f_kernel_stats.csv ;;
  bigkey)   # compiled filters in front of the partitioned GROUP BY (10 M INT64 keys): the mask route vs the interpreter's INT32 column
    timeout 900 python -u -m pytest tests/test_zz_gpu_typed_filters.py tests/test_zz_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider -k "large_table or cfg4_one_billion" > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $out/pytest.log
    timeout 600 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --big-key --verify-rows 4e6 --only plain,guarded_div,sum_gt,affine > $out/bool_filter_big_key.jsonl 2> $out/err.log; echo "exit $?"
    timeout 600 python tools/bool_filter_bench.py --rows 1e9 --steps 3 --big-key --interpreted --only guarded_div,sum_gt,affine > $out/bool_filter_big_key_interpreted.jsonl 2>> $out/err.log; echo "exit $?"
    cut -c1-330 $out/bool_filter_big_key.jsonl $out/bool_filter_big_key_interpreted.jsonl; tail -3 $out/err.log ;;
  idxprof)  # index-partitioned family: per-kernel times of the benchmark's lowest steps (rocprofv3 kernel trace), then the refbench lines
    cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
    timeout 500 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o idx -- python tools/refbench.py --rows 1e9 --steps 3 --only ${3:-S001,S003,PHS005,PHS007,MSPHS005,MSPHS012,PHM006} --out $out/refbench_idx.jsonl > $out/refbench.log 2> $out/rocprof.err
    find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/idx_kernel_stats.csv \; ; rm -rf $out/trace; cut -c1-170 $out/idx_kernel_stats.csv | head -24; cut -c1-250 $out/refbench_idx.jsonl; tail -3 $out/rocprof.err ;;
  idxpmc)   # k_idx_scatter's instruction mix and stall split (SQ counters; kernel trace only)
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc -o pmc -- python tools/refbench.py --rows 1e9 --steps 1 --only ${3:-S001,PHS005} > $out/pmc.log 2>&1
    python tools/rocpd_stats.py $out/pmc/pmc_results.db > $out/pmc_stats.txt 2>&1; rm -rf $out/pmc; grep -E "k_idx" $out/pmc_stats.txt | cut -c1-260; tail -3 $out/pmc.log
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $out/pmc2 -o pmc -- python tools/refbench.py --rows 1e9 --steps 1 --only ${3:-S001,PHS005} > $out/pmc2.log 2>&1
    python tools/rocpd_stats.py $out/pmc2/pmc_results.db > $out/pmc2_stats.txt 2>&1; rm -rf $out/pmc2; grep -E "k_idx" $out/pmc2_stats.txt | cut -c1-260; tail -3 $out/pmc2.log ;;
  idxtest)  # index-partitioned family: device parity (plain + packed records), then the per-kernel profile of the lowest steps
    timeout 1500 python -u -m pytest tests/test_zz_gpu_refbench.py -m gpu -x -q -p no:cacheprovider -k "idx_partitioned" > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $out/pytest.log
    bash tools/gpu_r06.sh idxprof $out ${3:-S001,S003,PHS005,PHS006,PHS007,MSPHS005,MSPHS012,PHM006,MSBS005,MSPHM005} ;;
  nga)      # typed scan-aggregate members (software-pipelined): device parity of the NGA shapes, then the blocks-per-CU sweep
    timeout 900 python -u -m pytest tests/test_zz_gpu_refbench.py tests/test_zz_gpu_typed_filters.py -m gpu -x -q -p no:cacheprovider -k "NGA or nga or scan_agg" > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $out/pytest.log
    timeout 400 python tools/nga_sweep.py --rows 1e9 --bpc ${3:-0,2,3,4} > $out/nga_1b.jsonl 2> $out/nga.err; echo "nga exit $?"; cat $out/nga_1b.jsonl; tail -3 $out/nga.err ;;
  idxtraffic) # index-partitioned family: HBM traffic by PMC (separate FETCH_SIZE / WRITE_SIZE passes), packed and plain records
    for mode in packed plain; do
      fl=0; [ $mode = plain ] && fl=1024
      for grp in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_${mode}_$grp -o pmc -- python tools/refbench.py --rows 1e9 --steps 1 --flags $fl --only ${3:-S001,PHS005,PHS007,MSPHS005} > $out/pmc_${mode}_$grp.log 2>&1
        python tools/rocpd_stats.py $out/pmc_${mode}_$grp/pmc_results.db > $out/pmc_${mode}_${grp}_stats.txt 2>&1; rm -rf $out/pmc_${mode}_$grp
        echo "== $mode $grp"; grep -E "k_idx_(scatter|aggregate)" $out/pmc_${mode}_${grp}_stats.txt | grep "$grp" | cut -c1-200
      done
    done ;;
  suite)    # the whole -m gpu suite (no -x: every failure listed), then smoke()
    timeout 2700 python -u -m pytest tests -m gpu -q -p no:cacheprovider "${@:3}" > $out/pytest_gpu.log 2>&1
    echo "pytest exit $?"; grep -n "FAILED\|Fatal\|fault" $out/pytest_gpu.log | head -20; tail -2 $out/pytest_gpu.log | cut -c1-200
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
  *) echo "unknown step $step"; exit 2 ;;
esac

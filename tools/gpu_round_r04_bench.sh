#!/bin/bash
# round 4, final session part 2: every config's bench line (roofline + cpu_baseline + verify), the default line's rocprofv3
# kernel summary and FETCH / WRITE counters (-> profiles/traffic.json, tied to this build of kernels_part.hip), the
# reference's 57 benchmark steps at 1 B and 128 M rows
out=${1:-gpurun_out/round_r04}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
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
python tools/make_traffic_json.py $out/pmc_FETCH_SIZE_stats.txt $out/pmc_WRITE_SIZE_stats.txt 1e10 r04 && cp profiles/traffic.json $out/traffic.json
timeout 400 python bench.py > $out/bench_default_with_traffic.json 2> $out/bench_default_with_traffic.err; echo "bench (traffic) exit $?"
timeout 600 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_1b.jsonl > $out/refbench_1b.log 2>&1; echo "refbench 1B exit $?"
timeout 300 python tools/refbench.py --rows 128e6 --steps 3 --budget-ms 1500 --out $out/refbench_128m.jsonl > $out/refbench_128m.log 2>&1; echo "refbench 128M exit $?"
# HBM bytes the index-partitioned family moves (<= 20 B/row): PHS005 (8-byte records), MSPHS008 (16-byte records)
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_idx_$grp -o pmc -- python tools/refbench.py --rows 1e9 --steps 1 --only PHS005,MSPHS008 --out $out/refbench_idx_pmc_$grp.jsonl > $out/pmc_idx_$grp.log 2>&1
  python tools/rocpd_stats.py $out/pmc_idx_$grp/pmc_results.db > $out/pmc_idx_${grp}_stats.txt 2>&1; rm -rf $out/pmc_idx_$grp
  grep -E "k_idx_" $out/pmc_idx_${grp}_stats.txt | cut -c1-170
done
head -3 $out/cfg3f_kernel_stats.csv
